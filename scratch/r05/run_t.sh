R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_t; mkdir -p $O; cd $R
LIB=scratch/_exp/libstp_halo_timing.so timeout 600 python scratch/r05/p64_phase.py > $O/p64_phase.txt 2>&1
cat $O/p64_phase.txt
