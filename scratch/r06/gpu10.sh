cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "sum_of_resized or softmax or loss" > $O/ops.txt 2>&1; tail -4 $O/ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fit_gpu.py -q -x -k "psp or PSP or other_graphs or softmax or fpn or FPN or non_default_decoder or vgg16_under or multiclass" > $O/model.txt 2>&1; tail -4 $O/model.txt
for v in "1" "0" "1" "0"; do
  STP_UPSUM_BWD=$v timeout 600 python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused pyramid backward $v config 4', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
done
timeout 600 python bench.py --config 3 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 3', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
python scratch/r06/gemm_floor_table.py 4 > $O/floor_config4.txt 2>&1; grep -A16 "non-GEMM launches" $O/floor_config4.txt
