R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_ee; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "tap_channels" 2>&1 | tail -8 > $O/ops.txt
cat $O/ops.txt
timeout 2400 python -m pytest tests/ -q -x -m gpu -k "fpn or psp or FPN or PSP" 2>&1 | tail -12 > $O/model.txt
cat $O/model.txt
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | grep -n "final_conv\|tapsum\|total us" > $O/lt_fpn.txt; cat $O/lt_fpn.txt
python scratch/launch_table.py bf16 PSPNet resnet101 768 8 20 2>&1 | grep -n "final_conv\|tapsum\|total us" > $O/lt_psp.txt; cat $O/lt_psp.txt
for sw in 0 1 0 1; do
  STP_TAPSUM=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "FPN/resnet50 1024x1024 3-class bs4 bf16\|PSPNet" | cut -c1-120 | sed "s/^/tapsum=$sw /" >> $O/other.txt
done
cat $O/other.txt
