R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_jj; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "softmax or loss or cce" 2>&1 | tail -4 > $O/ops.txt
cat $O/ops.txt
timeout 2400 python -m pytest tests/ -q -x -m gpu -k "fpn or psp or FPN or PSP or deeplab or multiclass or softmax" 2>&1 | tail -4 > $O/model.txt
cat $O/model.txt
python scratch/launch_table.py bf16 PSPNet resnet101 768 8 20 2>&1 | grep -n "softmax\|total us" | head -4 > $O/lt_psp.txt; cat $O/lt_psp.txt
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | grep -n "softmax\|total us" | head -4 > $O/lt_fpn.txt; cat $O/lt_fpn.txt
