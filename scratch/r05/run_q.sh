# split-K halo variant: op tests, model tests, A/B, launch table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/run_q; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "split_k" 2>&1 | tail -15 > $O/ops.txt
cat $O/ops.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x 2>&1 | tail -5 > $O/model.txt
cat $O/model.txt
printf 'base STP_HALO_SPLITK=0\nsplitk STP_HALO_SPLITK=1\n' | bash scratch/r05/ab.sh run_q
STP_HALO_SPLITK=1 timeout 600 python scratch/launch_table.py > $O/launch_splitk.txt 2>&1
grep -n "stage4_unit" $O/launch_splitk.txt
