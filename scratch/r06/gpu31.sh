cd $GRAFT_REPO_ROOT
O=gpurun_out/r07d; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/gputest_full.txt 2>&1
grep -E "passed|failed|error" $O/gputest_full.txt | tail -5 | tee $O/gputest_tail.txt
