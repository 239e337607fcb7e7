cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01c
STP_SIDE_STREAM_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/kte -o kte -- python $R/bench.py --eager --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01c/bench_eager_profiled.json 2> $R/gpurun_out/r01c/kte.err
head -4 $R/gpurun_out/r01c/kte/kte_kernel_stats.csv | cut -c1-160
cut -c1-300 $R/gpurun_out/r01c/bench_eager_profiled.json
