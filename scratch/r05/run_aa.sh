R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_aa; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -k "halo" 2>&1 | tail -3 > $O/ops.txt
cat $O/ops.txt
MODES=stats,bnb LIB=scratch/_exp/libstp_halo_timing.so timeout 600 python scratch/halo_timing.py 2>&1 | grep -v amdgpu.ids | grep "var 1" > $O/halo_timing.txt
cat $O/halo_timing.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -3 > $O/model.txt
cat $O/model.txt
printf 'generic STP_EPILOGUE_SPECIAL=0\nspecial STP_EPILOGUE_SPECIAL=1\n' | bash scratch/r05/ab.sh run_aa
