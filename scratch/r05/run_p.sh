cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05p}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "bilinear or softmax or loss" > $O/optest.txt 2>&1; tail -4 $O/optest.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -k "pspnet or fpn or three_class or deeplab or softmax" > $O/modeltest.txt 2>&1; tail -4 $O/modeltest.txt
python scratch/launch_table.py bf16 PSPNet resnet101 768 8 20 2>&1 | sed -n "/---- resize/,\$p" | grep -E "resize_bilinear_bwd|softmax"
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 2>&1 | sed -n "/---- resize/,\$p" | grep -E "resize_bilinear_bwd|softmax"
python scratch/other_configs_bench.py FPN:bf16 PSPNet:bf16 > $O/other_configs.txt 2>&1; cat $O/other_configs.txt
