R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-bn}; mkdir -p $O; cd $R
python - > $O/bn_fused.txt 2>&1 <<PY
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
m = HipSegModel("Unet", "resnet34", (512, 512, 3), 1, "sigmoid", batch=16, dtype="bf16", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
p = m.plan
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (16, 512, 512, 3)).astype(np.uint8), (rng.rand(16, 512, 512, 1) < 0.2).astype(np.uint8))
st = torch.cuda.current_stream()
launches = [l for l in p.prep + p.fwd + p.bwd + p.opt if l[0] is not None]
tot = {}
reps = 3
for rep in range(reps + 1):
    evs = []
    for fn, args, name, meta in launches:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); rc = fn(*args, st.cuda_stream); e1.record(st)
        assert rc == 0, name
        evs.append((e0, e1))
    torch.cuda.synchronize()
    if rep:
        for i, (e0, e1) in enumerate(evs):
            tot[i] = tot.get(i, 0.0) + e0.elapsed_time(e1) * 1e3 / reps
agg = {}
for i, (fn, args, name, meta) in enumerate(launches):
    if name.startswith("stp_bn"):
        if name == "stp_bn_finalize_apply": rows, C, tiles = args[5], args[6], args[1]
        elif name == "stp_bn_finalize": rows, C, tiles = args[2], args[3], args[1]
        elif name == "stp_bn_apply": rows, C, tiles = args[4], args[5], 0
        elif name in ("stp_bn_backward_fused",): rows, C, tiles = args[4], args[5], args[10]
        elif name in ("stp_bn_backward_fused_add",): rows, C, tiles = args[5], args[6], args[11]
        else: rows, C, tiles = 0, 0, 0
        k = (name, rows, C, tiles)
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += tot[i]
print("total", sum(tot.values()))
for k, (n, t) in sorted(agg.items()):
    print("%-28s rows %8d C %4d tiles %5d  x%2d  avg %6.1f us" % (k[0], k[1], k[2], k[3], n, t / n))
PY
cat $O/bn_fused.txt | grep -v amdgpu
