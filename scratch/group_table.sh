# times of the grouped weight-gradient launches under a list of environment settings (one eager launch table each)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-gt}; mkdir -p $O; cd $R
while read -r label envs; do
  [ -z "$label" ] && continue
  env $envs python scratch/launch_table.py > $O/lt_$label.txt 2>&1
  echo "== $label [$envs]  $(grep '^total' $O/lt_$label.txt)" | tee -a $O/groups.txt
  grep "^group" $O/lt_$label.txt | tee -a $O/groups.txt
done
