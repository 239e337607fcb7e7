cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
for v in "STP_FUSE_POOL_BN=0" "STP_FUSE_POOL_BN=1" "STP_FUSE_POOL_BN=0" "STP_FUSE_POOL_BN=1"; do
  for c in 4 3; do
    env $v timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config $c', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
  done
done
