cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pg; mkdir -p $R/gpurun_out/pg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pg -o g -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile > $R/gpurun_out/pg/bench.json 2> $R/gpurun_out/pg/err.txt
cut -c1-200 $R/gpurun_out/pg/bench.json
