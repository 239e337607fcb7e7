cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "sum_of_resized" > $O/ops_us.txt 2>&1; tail -4 $O/ops_us.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fit_gpu.py tests/test_fullsize_gpu.py -q -x -s -k "psp or PSP or other_graphs or non_default_decoder or vgg16_under" > $O/model_psp.txt 2>&1; grep -E "passed|failed|storage-quantised|NOISE|device vs" $O/model_psp.txt | tail -20
for sp in 1 0 1 0; do
  STP_PSP_SPLIT=$sp timeout 600 python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('STP_PSP_SPLIT=$sp config 4', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'], 'mfma frac', d['step_mfma_frac'])" | tee -a $O/step_ab.txt
done
python scratch/r06/gemm_floor_table.py 4 > $O/floor_config4.txt 2>&1; grep -A30 "non-GEMM launches" $O/floor_config4.txt | head -34
