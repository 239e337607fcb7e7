cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01c
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r01c/bench.json 2> gpurun_out/r01c/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01c/kt -o kt -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r01c/bench_profiled.json 2> $R/gpurun_out/r01c/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_fetch -o f -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/r01c/pmc_fetch.json 2> $R/gpurun_out/r01c/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r01c/pmc_write -o w -- python $R/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/r01c/pmc_write.json 2> $R/gpurun_out/r01c/pmc_write.err
ls -la $R/gpurun_out/r01c $R/gpurun_out/r01c/*/ | head -40
