cd $GRAFT_REPO_ROOT
O=gpurun_out/r07c; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/gputest_tail.txt
python bench.py > $O/bench_config1.json 2> $O/bench_config1.err
python bench.py --config 3 --cpu-baseline short > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --config 4 --cpu-baseline short > $O/bench_config4.json 2> $O/bench_config4.err
