cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
python - <<'PY'
import torch, os, glob
pr=torch.cuda.get_device_properties(0)
print([a for a in dir(pr) if 'pci' in a.lower()], getattr(pr,'pci_bus_id',None), getattr(pr,'pci_device_id',None), getattr(pr,'pci_domain_id',None))
for c in sorted(glob.glob("/sys/class/drm/card*/device"))[:10]: print(c, os.path.basename(os.path.realpath(c)))
import sys; sys.path.insert(0,'.')
import bench
c=bench.ClockSampler(0); print(c.card, c.files)
PY
python bench.py --config 4 --no-cpu-baseline --no-kernel-profile --sustain 3 > $O/bench_c4.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print(d['ms_per_step'], d['sustained'])"
python bench.py --no-cpu-baseline --no-kernel-profile --sustain 3 > $O/bench_c1.json 2>> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench_c1.json')); print(d['ms_per_step'], d['sustained'])"
timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_distributed_cpu.py -q > $O/dp.txt 2>&1; tail -3 $O/dp.txt
