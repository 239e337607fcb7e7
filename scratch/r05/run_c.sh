cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05c}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo" > $O/optest.txt 2>&1; tail -5 $O/optest.txt
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "fullsize or hipgraph or fp32_step_matches" > $O/modeltest.txt 2>&1; tail -3 $O/modeltest.txt
printf 'halo2_off STP_HALO2=0\nhalo2_on STP_HALO2=1\nhalo2_on_64 STP_HALO2=1 STP_HALO2_64=1\n' | bash scratch/r05/ab.sh $T
STP_HALO2=0 python scratch/launch_table.py > $O/launch_table_off.txt 2>&1
STP_HALO2_64=1 python scratch/launch_table.py > $O/launch_table_on64.txt 2>&1
