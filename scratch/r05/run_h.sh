cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-r05h}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python scratch/r05/s2d_bench.py > $O/s2d_bench.txt 2>&1; cat $O/s2d_bench.txt
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "folded_shortcut or space_to_depth" > $O/optest.txt 2>&1; tail -5 $O/optest.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > $O/modeltest.txt 2>&1; tail -4 $O/modeltest.txt
printf 'fold1_off STP_FOLD1=0\nfold1_on STP_FOLD1=1\n' | bash scratch/r05/ab.sh $T
python scratch/launch_table.py > $O/launch_table_on.txt 2>&1
