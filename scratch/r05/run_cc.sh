R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_cc; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "scatter2x or low_resolution_gemm" 2>&1 | tail -3 > $O/ops.txt
cat $O/ops.txt
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | grep -n "scatter\|total us\|unit1_sc .*dgrad" > $O/lt_fpn.txt; cat $O/lt_fpn.txt
for sw in 0 1 0 1; do
  STP_SCATTER_1X1S2=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "FPN/resnet50 1024x1024 3-class bs4 bf16\|PSPNet" | cut -c1-120 | sed "s/^/scatter=$sw /" >> $O/other.txt
done
cat $O/other.txt
