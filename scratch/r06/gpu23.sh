cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "resize or pyramid" 2>&1 | tail -8 | tee $O/op_test.txt
for v in 0 1; do STP_RESIZE_BWD_TILE=$v timeout 300 python scratch/r06/resize_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/resize_bwd_bench.txt; done
for v in "STP_RESIZE_BWD_TILE=0" "STP_RESIZE_BWD_TILE=1" "STP_RESIZE_BWD_TILE=0" "STP_RESIZE_BWD_TILE=1"; do
  for c in 3; do
    env $v timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config $c', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'], 'loss', d['config'].get('loss_after_run'))" | tee -a $O/step_ab.txt
  done
done
