cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "upsampled_logits or softmax_categorical" 2>&1 | tail -15 | tee $O/op_test.txt
timeout 600 python scratch/r06/up_loss_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up_loss_bench.txt
