cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "pointwise or taps_cancel or sum_of_resized or weight_gradient" > $O/ops.txt 2>&1; tail -4 $O/ops.txt
for v in "2 0" "3 256" "3 0" "2 0" "3 256"; do set -- $v
  for c in 4 3; do
    STP_WGRAD_DMA_STAGES=$1 STP_WGRAD_BLOCKS=$2 timeout 600 python bench.py --config $c --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad dma stages $1 blocks $2 config $c', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
  done
done
