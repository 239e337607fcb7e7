R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_feedexp2}; mkdir -p $O; cd $R
ab() { label=$1; shift; env "$@" python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label resident %.3f fed %.3f ms  loss %s' % (d['ms_per_step'], d['ms_per_step_with_feed'], d['config']['loss_after_run']))" | tee -a $O/feed_exp2.txt; }
ab three_buffers_host_throttle
ab again
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('20 steps: resident %.3f fed %.3f ms' % (d['ms_per_step'], d['ms_per_step_with_feed']))" | tee -a $O/feed_exp2.txt
