R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_bb; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "scatter2x or low_resolution_gemm" 2>&1 | tail -4 > $O/ops.txt
cat $O/ops.txt
timeout 2400 python -m pytest tests/ -q -x -m gpu -k "resnet50 or resnet101 or fpn or psp or FPN or PSP" 2>&1 | tail -4 > $O/model.txt
cat $O/model.txt
for sw in 0 1; do
  STP_SCATTER_1X1S2=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep workload | cut -c1-170 | sed "s/^/scatter=$sw /" >> $O/other.txt
done
cat $O/other.txt
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | grep -n "unit1_sc\|scatter\|total us" > $O/lt_fpn.txt; cat $O/lt_fpn.txt
