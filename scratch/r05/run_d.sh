cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05d}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo or pre_reduced or conv2d" > $O/optest.txt 2>&1; tail -5 $O/optest.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -q -x > $O/modeltest.txt 2>&1; tail -3 $O/modeltest.txt
printf 'group_off STP_STATS_GROUP=0\ngroup_on STP_STATS_GROUP=1\n' | bash scratch/r05/ab.sh $T
python scratch/launch_table.py > $O/launch_table.txt 2>&1
