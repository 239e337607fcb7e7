cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -q -k "sum_of_resized" > $O/ops.txt 2>&1; tail -4 $O/ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x -k "psp or PSP or other_graphs" > $O/model.txt 2>&1; tail -3 $O/model.txt
for i in 1 2; do
timeout 600 python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 0 --no-feed --no-calibration --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 4', d['ms_per_step'], 'without augmentation', d['ms_per_step_without_augmentation'])" | tee -a $O/step_ab.txt
done
python scratch/r06/gemm_floor_table.py 4 > $O/floor_config4.txt 2>&1; grep -A18 "non-GEMM launches" $O/floor_config4.txt
