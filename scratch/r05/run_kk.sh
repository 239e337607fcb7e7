R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_kk; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "one_launch_for_wide_tables" 2>&1 | tail -8 > $O/ops.txt
cat $O/ops.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -4 > $O/model.txt
cat $O/model.txt
printf 'separate STP_BN_FA_TILES=0\nonelaunch STP_BN_FA_TILES=1\n' | bash scratch/r05/ab.sh run_kk
