python -m pytest tests/test_ops_gpu.py -q -m gpu -k "grouped_weight" 2>&1 | tail -4 > gpurun_out/r04j_ops.txt
STP_WGRAD_TAPS9_M32=0 STP_WGRAD_TAPS9=2 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "grouped_weight" 2>&1 | tail -3 >> gpurun_out/r04j_ops.txt
python -m pytest tests/test_model_gpu.py -q -m gpu -k "bf16_fullsize or two_destination" 2>&1 | grep -v amdgpu | tail -40 > gpurun_out/r04j_tests.txt
bash scratch/r04/ab_bench_fields.sh r04j_m32 < scratch/r04/ab_m32.txt
bash scratch/r04/ab_bench_fields.sh r04j_sched < scratch/r04/ab_sched.txt
python -m pytest tests/test_fit_gpu.py tests/test_dp_gpu.py tests/test_convergence_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | tail -5 > gpurun_out/r04j_fit.txt
python scratch/launch_table.py bf16 > gpurun_out/r04j_launch_table.txt 2>&1
