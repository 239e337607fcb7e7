R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_ii; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "resized_on_the_fly or softmax" 2>&1 | tail -8 > $O/ops.txt
cat $O/ops.txt
timeout 2400 python -m pytest tests/ -q -x -m gpu -k "fpn or psp or FPN or PSP or deeplab" 2>&1 | tail -8 > $O/model.txt
cat $O/model.txt
for sw in 0 1 0 1; do
  STP_LOSS_UP=$sw timeout 900 python scratch/other_configs_bench.py 2>&1 | grep "FPN/resnet50 1024x1024 3-class bs4 bf16\|PSPNet" | cut -c1-120 | sed "s/^/lossup=$sw /" >> $O/other.txt
done
cat $O/other.txt
python scratch/launch_table.py bf16 PSPNet resnet101 768 8 20 2>&1 | grep -n "softmax\|resize_bilinear  \|stp_resize_bilinear \|total us" | head > $O/lt_psp.txt; cat $O/lt_psp.txt
