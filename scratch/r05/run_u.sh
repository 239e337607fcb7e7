R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_u; mkdir -p $O; cd $R
printf 'base STP_HALO_64V4=0\nv4 STP_HALO_64V4=1\np64v1 STP_HALO_P64=1\n' | bash scratch/r05/ab.sh run_u
