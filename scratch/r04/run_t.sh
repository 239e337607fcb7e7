python -m pytest tests/test_ops_gpu.py -q -m gpu -k "two_destinations_with_summed" 2>&1 | tail -12 > gpurun_out/r04t_ops.txt
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/r04t_gputest.txt
python scratch/other_configs_bench.py > gpurun_out/r04t_other_configs.txt 2>&1
