# A/B of step-level switches on one box: stdin lines "<label> <env assignments>"; 40 graph steps each, two repetitions, interleaved
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-ab}; mkdir -p $O; shift
cd $R
cat > $O/.cases
for rep in 1 2 3; do
  while read -r label envs; do
    [ -z "$label" ] && continue
    ms=$(env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile --no-feed --sustain 0 2>>$O/err.txt | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$label rep$rep $ms ms   [$envs]" | tee -a $O/ab.txt
  done < $O/.cases
done
