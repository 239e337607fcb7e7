R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_w; mkdir -p $O; cd $R
for tag in timing_old timing; do
  echo "== $tag" >> $O/halo_timing.txt
  LIB=scratch/_exp/libstp_halo_$tag.so timeout 600 python scratch/halo_timing.py 2>&1 | grep -v amdgpu.ids >> $O/halo_timing.txt
done
cat $O/halo_timing.txt
