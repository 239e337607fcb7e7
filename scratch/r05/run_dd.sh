R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_dd; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests/ -q -x -m gpu -k "resnet50 or resnet101 or fpn or psp or FPN or PSP or scatter2x" 2>&1 | tail -4 > $O/model.txt
cat $O/model.txt
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | grep -n "scatter\|total us\|unit1_sc .*dgrad\|unit1_conv1 .*dgrad" > $O/lt_fpn.txt; cat $O/lt_fpn.txt
for sw in 0 1 0 1; do
  STP_SCATTER_1X1S2=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "FPN/resnet50 1024x1024 3-class bs4 bf16\|PSPNet" | cut -c1-120 | sed "s/^/scatter=$sw /" >> $O/other.txt
done
cat $O/other.txt
