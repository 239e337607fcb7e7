R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_ll; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "one_launch_for_wide_tables" 2>&1 | tail -4 > $O/ops.txt
cat $O/ops.txt
printf 'separate STP_BN_FA_TILES=0\nonelaunch STP_BN_FA_TILES=1\n' | bash scratch/r05/ab.sh run_ll
STP_BN_FA_TILES=1 timeout 600 python scratch/launch_table.py 2>&1 | grep -n "finalize_apply_tiles\|stp_bn_finalize \|stp_bn_apply  \|total us" | head -8
