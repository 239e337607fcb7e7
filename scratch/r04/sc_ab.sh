#!/bin/bash
# on the GPU box: op tests of the small-channel kernels with the in-tree build, then scratch/sc_bench.py for every what-if library named
# ("prod" = the in-tree library)
cd "$(dirname "$0")/../.."
OUT=gpurun_out/${1:-sc_ab}; shift; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -15 > $OUT/tests.txt
for t in "$@"; do
  if [ $t = prod ]; then timeout 300 python scratch/sc_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_$t.txt
  else EXP=$t timeout 300 python scratch/sc_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_$t.txt; fi
done
for t in "$@"; do echo "== $t"; cat $OUT/bench_$t.txt; done > $OUT/all.txt
tail -15 $OUT/tests.txt
