cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05j}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
STP_HALO_RING=6 timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "halo" > $O/optest_ring6.txt 2>&1; tail -4 $O/optest_ring6.txt
STP_HALO_RING=6 timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "fullsize or hipgraph or fp32_step_matches or storage_quantised" > $O/modeltest_ring6.txt 2>&1; tail -4 $O/modeltest_ring6.txt
printf 'ring4 STP_HALO_RING=4\nring6 STP_HALO_RING=6\n' | bash scratch/r05/ab.sh $T
STP_HALO_RING=4 python scratch/launch_table.py > $O/launch_table_ring4.txt 2>&1
STP_HALO_RING=6 python scratch/launch_table.py > $O/launch_table_ring6.txt 2>&1
