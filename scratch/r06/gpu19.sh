cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s; mkdir -p $O
timeout 600 python scratch/r06/up_loss_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up_loss_bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_up -o up -- bash -c "cd $GRAFT_REPO_ROOT && python scratch/r06/up_loss_bench.py" > /dev/null 2>&1
f=$(find /tmp/prof_up -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220 | tee $GRAFT_REPO_ROOT/$O/up_kernel_stats.txt
