# round-5 first GPU call: full GPU suite on the host-side changes, bench line with the new fields, eager launch table (baseline of the round)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05a}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gputest.txt 2>&1; tail -5 $O/gputest.txt
python bench.py --cpu-baseline short > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench.err
python scratch/launch_table.py > $O/launch_table.txt 2>&1
