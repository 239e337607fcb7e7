R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_z; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_ops_gpu.py -q -x -k "halo" 2>&1 | tail -3 > $O/ops.txt
cat $O/ops.txt
printf 'generic STP_EPILOGUE_SPECIAL=0\nspecial STP_EPILOGUE_SPECIAL=1\n' | bash scratch/r05/ab.sh run_z
