"""Layer bench of the 1x1 convolutions of configs[3] / [4] (HIP events, 20 launches after 3 warm ones, operands cycled through 6 buffer
sets = > 256 MB so that nothing is served from the Infinity Cache): run once with STP_PW=1 (the streaming kernel) and once with STP_PW=0
(the per-tap kernel).  Columns: us, TB/s of algorithmic bytes, ratio to the 6.3 TB/s floor."""
import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops, _lib
lib = _lib.load()
DEV = "cuda"
# (pixels n,h,w ; cin ; cout ; mode)
CASES = [((8, 192, 192), 64, 256, "stats_res"), ((8, 192, 192), 64, 256, "plain"), ((8, 192, 192), 256, 64, "stats"), ((8, 192, 192), 64, 64, "stats"),
         ((8, 192, 192), 64, 256, "bnb"), ((8, 192, 192), 256, 64, "bnb"), ((8, 192, 192), 256, 64, "plain"), ((8, 192, 192), 64, 64, "bnb_acc"),
         ((8, 192, 192), 256, 128, "stats"), ((8, 192, 192), 128, 256, "bnb_acc"),
         ((8, 96, 96), 128, 512, "stats_res"), ((8, 96, 96), 512, 128, "stats"), ((8, 96, 96), 128, 512, "bnb"), ((8, 96, 96), 512, 128, "bnb"),
         ((8, 96, 96), 512, 256, "plain"), ((4, 256, 256), 256, 256, "bias"), ((4, 128, 128), 512, 256, "bias"), ((4, 128, 128), 256, 512, "plain")]
NSET = 6
for (n, h, w), cin, cout, mode in CASES:
    P_ = n * h * w
    sets = []
    for i in range(NSET):
        x = torch.randn(P_, cin, device=DEV).to(torch.bfloat16)
        y = torch.zeros(P_, cout, device=DEV, dtype=torch.bfloat16)
        r = torch.randn(P_, cout, device=DEV).to(torch.bfloat16) if mode in ("stats_res", "bnb", "bnb_acc") else None
        sets.append((x, y, r))
    wt = (torch.randn(cout, cin, device=DEV) / cin ** 0.5).to(torch.bfloat16)
    vec = lambda: torch.rand(cout, device=DEV) + 0.5
    g, b, m, rs, bias = vec(), vec(), vec(), vec(), vec()
    ps = []
    for x, y, r in sets:
        p = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=cin, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=cout, dtype=ops.dt(y),
                            residual=r if mode == "stats_res" else None, bias=bias if mode == "bias" else None, accumulate0=int(mode == "bnb_acc"))
        st = None
        if mode.startswith("bnb"):
            p.bnb_x, p.bnb_mean, p.bnb_rstd, p.bnb_gamma, p.bnb_beta, p.bnb_relu = ops.ptr(r), ops.ptr(m), ops.ptr(rs), ops.ptr(g), ops.ptr(b), 1
        if mode.startswith("bnb") or mode.startswith("stats"):
            p.stats_partial = 1
            st = torch.zeros(max(4, ops.conv2d_stats_floats(p)), device=DEV)
            p.stats_partial = ops.ptr(st)
        ps.append((p, st))
    tile = int(lib.stp_conv2d_tile_for(ops.C.byref(ps[0][0])))
    for i in range(3):
        ops.conv2d(ps[i % NSET][0])
    torch.cuda.synchronize()
    evs = []
    for i in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d(ps[i % NSET][0]); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    us = np.median([a.elapsed_time(b) * 1e3 for a, b in evs])
    byt = P_ * (cin + cout * (1 + (mode in ("stats_res", "bnb")) + 2 * (mode == "bnb_acc"))) * 2
    print("%-14s %4d -> %-4d %-10s tile %-5d %8.1f us  %5.2f TB/s  x%.2f of the 6.3 TB/s floor   %6.1f TFLOP/s" % (
        "%dx%dx%d" % (n, h, w), cin, cout, mode, tile, us, byt / us / 1e6, us / (byt / 6.3e6), 2.0 * P_ * cin * cout / us / 1e6))
