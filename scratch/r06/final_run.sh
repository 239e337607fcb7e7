# final run of round 6: smoke, counters + bench lines + kernel statistics for the three BASELINE configs on ONE box (tag = $1)
cd $GRAFT_REPO_ROOT
T=${1:-r06fin}
O=gpurun_out/$T; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash scratch/r06/prof_r06.sh $T 1 full
bash scratch/r06/prof_r06.sh $T 3 short
bash scratch/r06/prof_r06.sh $T 4 short
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_100steps.json 2> $O/bench_100steps.err
ls $O | head -60
