cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v; mkdir -p $O
timeout 600 python scratch/r06/up_loss_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up_loss_bench.txt
