# A/B of step-level switches on one box: each line = "<label> <env assignments>"; 40 graph steps each, no kernel profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; shift
cd $R
while read -r label envs; do
  [ -z "$label" ] && continue
  for rep in 1 2; do
    ms=$(env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>>$O/err.txt | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$label rep$rep $ms ms   [$envs]" | tee -a $O/ab.txt
  done
done
