cd $GRAFT_REPO_ROOT
O=gpurun_out/r06q; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/gputest_tail.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $O/smoke.txt
timeout 600 python bench.py 2>/dev/null | tee $O/bench_default.json
