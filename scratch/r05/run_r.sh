R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_r; mkdir -p $O; cd $R
LIB=scratch/_exp/libstp_halo_timing.so timeout 600 python scratch/r05/halo_phase.py > $O/phase.txt 2>&1
cat $O/phase.txt
