python -m pytest tests/test_ops_gpu.py -q -m gpu -k "two_destinations_with_summed or halo_kernel" 2>&1 | tail -12 > gpurun_out/r04s_ops.txt
bash scratch/r03_ab.sh r04s_foldup < scratch/r04/ab_foldup.txt
python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_convergence_gpu.py tests/test_dp_gpu.py -q -m gpu 2>&1 | tail -8 > gpurun_out/r04s_model.txt
python scratch/launch_table.py bf16 > gpurun_out/r04s_launch_table.txt 2>&1
