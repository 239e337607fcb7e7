R=$GRAFT_REPO_ROOT; cd $R
while read -r label envs; do
  [ -z "$label" ] && continue
  echo "== $label [$envs]"
  env $envs python scratch/other_configs_bench.py FPN:bf16 PSPNet:bf16 2>&1 | grep workload | cut -c1-120
done
