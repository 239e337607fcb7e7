# round-3 profile set: bench line, rocprofv3 kernel stats (graph + eager), PMC traffic (FETCH / WRITE, separate passes), SQ counters.
# usage (on the GPU box): bash scratch/prof_r03.sh <tag>      -> gpurun_out/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03z}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
python bench.py --steps 100 --warmup 10 > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2>> $O/bench.err
python bench.py --dtype fp32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile > $O/bench_fp32.json 2>> $O/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_profiled.json 2> $O/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kte -o kte -- python $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_eager_profiled.json 2> $O/kte.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $O/pmc_fetch.json 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $O/pmc_write.json 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_sq -o q -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $O/pmc_sq.json 2> $O/pmc_sq.err
cd $R
python scratch/pmc_aggregate.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json
python scratch/pmc_aggregate_sq.py $(find $O/pmc_sq -name "*counter_collection.csv" | head -1) $(find $O/pmc_sq -name "*kernel_trace.csv" | head -1) $O/pmc_sq.json
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/kte -name "*kernel_stats.csv" | head -1) $O/kernel_stats_eager.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/kt $O/kte
ls -la $O; cut -c1-300 $O/bench.json
python scratch/launch_table.py > $O/launch_table.txt 2>&1
python scratch/other_configs_bench.py > $O/other_configs.txt 2>&1
