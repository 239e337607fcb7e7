cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
timeout 900 python scratch/r06/psp384_debug.py 2>&1 | grep -v amdgpu.ids | tee $O/psp384_debug.txt
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "pyramid_pooling" 2>&1 | tail -4 | tee $O/op_test.txt
