R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_hh; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/test_ops_gpu.py -q -x 2>&1 | tail -4 > $O/ops.txt
cat $O/ops.txt
for sw in 0 1 0 1; do
  STP_EPILOGUE_SPECIAL=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "workload" | grep -v fp16 | cut -c1-120 | sed "s/^/special=$sw /" >> $O/other.txt
done
cat $O/other.txt
