cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
timeout 300 python scratch/r06/pw_debug.py > $O/pw_debug.txt 2>&1; grep -v amdgpu $O/pw_debug.txt | head -80
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "batched_reduce" > $O/ops_br.txt 2>&1; tail -5 $O/ops_br.txt
