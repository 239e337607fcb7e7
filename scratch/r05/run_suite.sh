R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/suite; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > $O/gputest_tail.txt
cat $O/gputest_tail.txt
