R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_y; mkdir -p $O; cd $R
STP_HALO_P64=1 timeout 600 python scratch/launch_table.py > $O/launch_p64.txt 2>&1
grep -n "stage1_unit\|bn_finalize\|stp_bn_apply  .*rows   262144\|tile=1029" $O/launch_p64.txt | head -60
