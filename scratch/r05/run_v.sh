R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_v; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "halo_kernel_forward_statistics_residual or halo_kernel_batchnorm_backward_sums" 2>&1 | tail -15 > $O/ops.txt
cat $O/ops.txt
VARS=2,4,5 timeout 600 python scratch/r05/p64_bench.py 2>&1 | grep -v amdgpu.ids > $O/p64_bench.txt
cat $O/p64_bench.txt
LIB=scratch/_exp/libstp_halo_timing.so timeout 600 python scratch/r05/p64_phase.py 2>&1 | grep -v amdgpu.ids > $O/p64_phase.txt
cat $O/p64_phase.txt
