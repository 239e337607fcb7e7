cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
timeout 900 python scratch/r06/wgrad1x1_split_sweep.py 2>&1 | tee $O/wgrad1x1_split_sweep.txt
