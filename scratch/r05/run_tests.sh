# full GPU suite (no -x: every failure is listed) + smoke
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05t}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -m gpu -q -s > $O/gputest.txt 2>&1; tail -15 $O/gputest.txt
