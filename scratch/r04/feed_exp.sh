R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r04_feedexp}; mkdir -p $O; cd $R
ab() { label=$1; shift; env "$@" python bench.py --steps 100 --warmup 8 --no-cpu-baseline --no-kernel-profile --sustain 0 2>>$O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$label resident %.3f fed %.3f ms' % (d['ms_per_step'], d['ms_per_step_with_feed']))" | tee -a $O/feed_exp.txt; }
ab default
ab noevents STP_FEED_EXP=noevents
ab nocopy STP_FEED_EXP=nocopy
ab stage_after STP_FEED_EXP=stage_after
ab default2
