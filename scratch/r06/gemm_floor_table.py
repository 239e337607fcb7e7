"""Every GEMM launch of a training step against BOTH of its floors (review item 5b: decide the 1x1 kernel on in-step numbers):
eager step, HIP events per launch (3 repetitions after a warm one); per launch the algorithmic FLOP (plan meta), the ALGORITHMIC bytes
(every operand tensor of the launch once: sources, destination(s), residual, accumulate reads, the BatchNormalization input of a fused
backward epilogue; weights once), the time both would take at 2.5 PFLOP/s / 6.3 TB/s (what a device copy reaches), and the ratio of
the measured time to the larger floor.
usage: python scratch/r06/gemm_floor_table.py <config 1|3|4> [dtype]"""
import ctypes as C, sys, json, numpy as np, torch
sys.path.insert(0, ".")
import bench
from segmentation_training_pipeline_amd.backend import HipSegModel
ci = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = bench.CONFIGS[ci]
dt = sys.argv[2] if len(sys.argv) > 2 else cfg["dtype"]
arch, bb, size, batch, classes = cfg["architecture"], cfg["backbone"], cfg["size"], cfg["batch"], cfg["classes"]
m = HipSegModel(arch, bb, (size, size, 3), classes, "sigmoid" if classes == 1 else "softmax", batch=batch, dtype=dt,
                loss=bench.LOSS if classes == 1 else bench.LOSS_SOFTMAX, use_graph=False)
p = m.plan
_, img, msk = bench.synthetic_data(0, batch, size, classes)
m.load_batch(img, msk[..., None])
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
es = 4 if dt == "fp32" else 2
print("config %d: %s/%s %dx%d bs%d %d-class %s   eager step, sum of launches %.1f us, %d launches" % (ci, arch, bb, size, size, batch, classes, dt, sum(tot.values()), len(launches)))
rows = []
for i, (fn, args, name, meta) in enumerate(launches):
    if not (meta and "flops" in meta):
        continue
    us = tot[i]
    if name == "stp_conv2d":
        q = args[0]._obj
        src = q.N * q.Hs0 * q.Ws0 * q.C0 + (q.N * q.Hv * q.Wv * q.C1 if q.src1 else 0)
        if q.s2d_dgrad:
            dst = q.N * (2 * q.Ho) * (2 * q.Wo) * (q.Cout // 4)
        elif q.dst_sum2x2:
            dst = q.N * (q.Ho // 2) * (q.Wo // 2) * q.Cd0 + q.N * q.Ho * q.Wo * (q.Cout - q.Cd0)
        else:
            dst = q.N * q.Ho * q.Wo * q.Cout
        extra = dst * (int(bool(q.residual)) + int(bool(q.accumulate0)) + int(bool(q.bnb_x)))
        if q.fold_src:
            extra += q.N * q.Hs0 * q.Ws0 * q.fold_C
        wts = q.Cout * q.KH * q.KW * (q.C0 + q.C1)
        byt = (src + dst + extra + wts) * es
        shape = "%dx%dx%d %d+%d->%d k%d s%d%s%s%s" % (q.N, q.Ho, q.Wo, q.C0, q.C1, q.Cout, q.KH, q.stride, " +res" if q.residual else "",
                                                   " +acc" if q.accumulate0 else "", " +bnb" if q.bnb_x else "")
    elif name == "stp_conv2d_wgrad":
        q = args[0]._obj
        src = q.N * q.Hs0 * q.Ws0 * q.C0 + (q.N * q.Hv * q.Wv * q.C1 if q.src1 else 0)
        dy = q.N * q.Ho * q.Wo * q.Cout
        byt = (src + dy) * es + q.Cout * q.KH * q.KW * (q.C0 + q.C1) * 4
        shape = "%dx%dx%d %d+%d->%d k%d s%d (+ reduce %.1f us)" % (q.N, q.Ho, q.Wo, q.C0, q.C1, q.Cout, q.KH, q.stride, tot.get(i + 1, 0.0))
    else:
        byt, shape = 0, ""
    t_m, t_h = meta["flops"] / 2.5e15 * 1e6, byt / 6.3e12 * 1e6
    rows.append((us, meta["layer"], meta["pass"], meta.get("tile", meta.get("cout")), shape, meta["flops"], byt, t_m, t_h))
print("%-34s %-6s %-5s %-46s %8s %8s %9s %8s %8s %6s" % ("layer", "pass", "tile", "shape", "us", "TFLOP/s", "MB", "mfma us", "hbm us", "x floor"))
for us, layer, pas, tile, shape, fl, byt, t_m, t_h in rows:
    print("%-34s %-6s %-5s %-46s %8.1f %8.1f %9.1f %8.1f %8.1f %6.2f" % (layer[:34], pas, tile, shape, us, fl / us / 1e6, byt / 1e6, t_m, t_h, us / max(t_m, t_h, 1e-9)))
print("---- by (pass, kernel size, tile): launches, us, floor us (sum of per-launch max floors), ratio")
agg = {}
for us, layer, pas, tile, shape, fl, byt, t_m, t_h in rows:
    k = (pas, "1x1" if " k1 " in shape + " " else "3x3+" , str(tile))
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += us; a[2] += max(t_m, t_h)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-8s %-5s tile %-6s %4d launches %9.1f us   floors %9.1f us   x %.2f" % (k + (a[0], a[1], a[2], a[1] / max(a[2], 1e-9))))
other = sum(tot[i] for i, l in enumerate(launches) if not (l[3] and "flops" in l[3]))
print("non-GEMM launches: %.1f us" % other)
agg = {}
for i, (fn, args, name, meta) in enumerate(launches):
    if not (meta and "flops" in meta):
        a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += tot[i]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("   %-32s %4d launches %9.1f us" % (k, a[0], a[1]))
