cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pyramid_pooling" 2>&1 | tail -8 | tee $O/op_test.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "psp or PSP or other_graphs" 2>&1 | tail -8 | tee $O/model_test.txt
for v in "STP_POOL_PYRAMID=0" "STP_POOL_PYRAMID=1" "STP_POOL_PYRAMID=0" "STP_POOL_PYRAMID=1"; do
  for c in 4; do
    env $v timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config $c', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'], 'loss', d['config'].get('loss_after_run'))" | tee -a $O/step_ab.txt
  done
done
