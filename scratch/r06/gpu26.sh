cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "psp or PSP or other_graphs" 2>&1 | tail -8 | tee $O/model_test.txt
