R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_s; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "halo_kernel_forward_statistics_residual or halo_kernel_batchnorm_backward_sums" 2>&1 | tail -15 > $O/ops.txt
cat $O/ops.txt
timeout 600 python scratch/r05/p64_bench.py > $O/p64_bench.txt 2>&1
cat $O/p64_bench.txt
