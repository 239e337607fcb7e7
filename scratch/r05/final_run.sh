#!/bin/bash
# final evidence of the round: smoke, full GPU suite, profile set (scratch/r05/prof_r05.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05final}; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/gputest_tail.txt; grep -E "passed|failed" $O/gputest_tail.txt
bash scratch/r05/prof_r05.sh $(basename $O) > $O/prof.log 2>&1
tail -2 $O/prof.log
