"""Per-launch timing table of the full-size step (eager, HIP events)."""
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
arch, bb, size, batch, classes = (sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else ("Unet", "resnet34", 512, 16, 1)
m = HipSegModel(arch, bb, (size, size, 3), classes, "sigmoid" if classes == 1 else "softmax", batch=batch, dtype=dt,
                loss="binary_crossentropy+1.0*dice_loss" if classes == 1 else "categorical_crossentropy+1.0*dice_loss", use_graph=False)
p = m.plan
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (batch, size, size, 3)).astype(np.uint8), (rng.rand(batch, size, size, 1) < 0.2).astype(np.uint8))
st = torch.cuda.current_stream()
launches = [l for l in p.prep + p.fwd + p.bwd + p.opt if l[0] is not None]
reps = 3
tot = {}
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
rows = []
for i, (fn, args, name, meta) in enumerate(launches):
    rows.append((tot[i], name, meta))
print("total us", sum(tot.values()))
agg = {}
for us, name, meta in rows:
    agg[name] = agg.get(name, 0) + us
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%-28s %9.1f us" % (k, v))
print("---- slowest launches")
for us, name, meta in sorted(rows, key=lambda r: -r[0])[:25]:
    print("%-28s %-28s %8.1f us" % (name, (meta or {}).get("layer", ""), us))
print("---- BatchNormalization launches (bytes = passes x rows x C x 2)")
for us, name, meta in rows:
    pass
for i, (fn, args, name, meta) in enumerate(launches):
    if name in ("stp_bn_backward_fused", "stp_bn_backward", "stp_bn_apply"):
        if name == "stp_bn_apply":
            r, c, passes = args[4], args[5], 2
        elif name == "stp_bn_backward_fused":
            r, c, passes = args[4], args[5], 3
        else:
            r, c, passes = args[4], args[5], 5
        byt = passes * r * c * 2
        print("%-24s rows %8d C %4d  %7.1f us  %6.2f TB/s" % (name, r, c, tot[i], byt / tot[i] / 1e6))
print("---- GEMM launches")
for us, name, meta in rows:
    if meta and 'flops' in meta:
        print("%-28s %-6s tile=%-3s %8.1f us %7.1f TF" % (meta["layer"], meta["pass"], meta.get("tile", meta.get("cout")), us, meta["flops"] / us / 1e6))
print("---- resize / pooling / loss launches (N, H, W, C, factor, ldo, coff)")
for i, (fn, args, name, meta) in enumerate(launches):
    if name in ("stp_resize_bilinear", "stp_resize_bilinear_bwd"):
        print("%-26s N %d H %d W %d C %d f %d ldo %d coff %d   %8.1f us" % ((name,) + tuple(args[2:9]) + (tot[i],)))
    elif name in ("stp_avgpool", "stp_avgpool_bwd", "stp_softmax_cce_dice", "stp_upsample2x_add", "stp_channel_sum", "stp_upsample2x_bwd"):
        print("%-26s %s   %8.1f us" % (name, tuple(a for a in args[2:8] if isinstance(a, int) and a < 10**7), tot[i]))
