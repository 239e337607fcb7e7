cd $GRAFT_REPO_ROOT
O=gpurun_out/r06fin_lines; mkdir -p $O
python bench.py > $O/bench_config1.json 2> $O/bench_config1.err
python bench.py --config 3 --cpu-baseline short > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --config 4 --cpu-baseline short > $O/bench_config4.json 2> $O/bench_config4.err
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_100steps.json 2> $O/bench_100steps.err
tail -c 300 $O/bench_config3.err
