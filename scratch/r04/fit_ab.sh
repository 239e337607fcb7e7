R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_fit}; mkdir -p $O; cd $R
for rep in 1 2; do
  for cfg in "lookahead_default" "no_lookahead STP_FEED_LOOKAHEAD=0"; do
    set -- $cfg; label=$1; shift
    env "$@" python scratch/fit_throughput.py 2>>$O/err.txt | grep "fit loop" | sed "s/^/$label rep$rep: /" | tee -a $O/fit_ab.txt
  done
done
python bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench.py on this box: resident %.3f fed %.3f ms' % (d['ms_per_step'], d['ms_per_step_with_feed']))" | tee -a $O/fit_ab.txt
python -m pytest tests/test_fit_gpu.py -q -m gpu 2>&1 | tail -3 | tee -a $O/fit_ab.txt
