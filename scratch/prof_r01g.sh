cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01g
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/kt.err
STP_SIDE_STREAM_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kte -o kte -- python $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_eager_profiled.json 2> $O/kte.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $O/pmc_write.json 2> $O/pmc_write.err
cd $R
python scratch/pmc_aggregate.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/kte -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eager.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/kt $O/kte
ls -la $O; cut -c1-400 $O/bench.json
