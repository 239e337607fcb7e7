# A/B of step-level switches, printing resident / fed / sustained: each line = "<label> <env assignments>"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; shift
cd $R
while read -r label envs; do
  [ -z "$label" ] && continue
  for rep in 1 2; do
    env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 3 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label rep$rep resident %.3f fed %.3f sustained %.3f ms   [$envs]' % (d['ms_per_step'], d['ms_per_step_with_feed'], d['sustained']['ms_per_step']))" | tee -a $O/ab.txt
  done
done
