cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x > $O/gputest.txt 2>&1; tail -8 $O/gputest.txt
