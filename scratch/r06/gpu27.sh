cd $GRAFT_REPO_ROOT
O=gpurun_out/r07a; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/gputest_tail.txt
