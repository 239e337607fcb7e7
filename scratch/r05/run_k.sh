cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05k}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fused_with_the_max_pooling or maxpool" > $O/optest.txt 2>&1; tail -4 $O/optest.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > $O/modeltest.txt 2>&1; tail -4 $O/modeltest.txt
printf 'poolfuse_off STP_FUSE_BN_POOL_FWD=0\npoolfuse_on STP_FUSE_BN_POOL_FWD=1\n' | bash scratch/r05/ab.sh $T
python scratch/other_configs_bench.py > $O/other_configs.txt 2>&1; cat $O/other_configs.txt
