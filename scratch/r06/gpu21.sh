cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "upsampled_logits or softmax_categorical" 2>&1 | tail -5 | tee $O/op_test.txt
timeout 600 python scratch/r06/up_loss_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up_loss_bench.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "fpn or FPN or psp or PSP or other_graphs or linknet" 2>&1 | tail -8 | tee $O/model_test.txt
for v in "STP_UP_LOSS=0" "STP_UP_LOSS=1" "STP_UP_LOSS=0" "STP_UP_LOSS=1"; do
  for c in 4 3; do
    env $v timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config $c', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'], 'loss', d['config'].get('loss_after_run'))" | tee -a $O/step_ab.txt
  done
done
