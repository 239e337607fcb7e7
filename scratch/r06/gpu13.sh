cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
python bench.py --cpu-baseline short > $O/bench_default_short.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_default_short.json')); print(d['ms_per_step'], d['sustained'], d.get('step_hbm_frac'), d['roofline']['counters_stale'])"
ls /sys/class/drm/ 2>/dev/null | head; ls /sys/class/drm/card*/device/hwmon/ 2>/dev/null | head
python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 3 > $O/bench_c4.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print(d['ms_per_step'], d['sustained'])"
timeout 900 python -m pytest tests/test_dp_gpu.py -q > $O/dp.txt 2>&1; tail -3 $O/dp.txt
