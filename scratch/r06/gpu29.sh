cd $GRAFT_REPO_ROOT
O=gpurun_out/r07b; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fit_gpu.py -x -q -k "filter or augment or listed_order or crops or background" 2>&1 | tail -6 | tee $O/op_test.txt
for v in "STP_FILTER_TILE=0" "STP_FILTER_TILE=1" "STP_FILTER_TILE=0" "STP_FILTER_TILE=1"; do
  env $v timeout 600 python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v config 4', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'], 'loss', d['config'].get('loss_after_run'))" | tee -a $O/step_ab.txt
done
