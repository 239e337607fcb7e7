cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05g}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "space_to_depth" > $O/optest.txt 2>&1; tail -8 $O/optest.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > $O/modeltest.txt 2>&1; tail -4 $O/modeltest.txt
printf 's2d_off STP_S2D=0\ns2d_on STP_S2D=1\npoolbn STP_FUSE_POOL_BN=1\n' | bash scratch/r05/ab.sh $T
python scratch/launch_table.py > $O/launch_table_on.txt 2>&1
python scratch/launch_table.py bf16 FPN resnet50 1024 4 3 > $O/launch_table_fpn.txt 2>&1
python scratch/launch_table.py bf16 PSPNet resnet101 768 8 20 > $O/launch_table_psp.txt 2>&1
