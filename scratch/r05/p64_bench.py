"""64 -> 64 @ 16 x 128 x 128 (ResNet34 stage 1) on the halo variants 2 / 3 / 4 and the persistent form 5: event-timed launches (100 back to back)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
for (n, h, w) in [(16, 128, 128), (16, 256, 256), (8, 128, 128)]:
    ci = co = 64
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    ref = None
    for v in [int(a) for a in os.environ.get("VARS", "2,3,4,5").split(",")]:
        for mode in ("plain", "stats", "bnb"):
            y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
            P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=1024 + v)
            keep = []
            if mode != "plain":
                st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV); keep.append(st)
                P.stats_partial = ops.ptr(st)
            if mode == "bnb":
                xb = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16); m_ = torch.zeros(co, device=DEV); r_ = torch.ones(co, device=DEV)
                keep += [xb, m_, r_]
                P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m_), ops.ptr(r_), None, None, 1
            try:
                for _ in range(5): ops.conv2d(P)
            except Exception as e:
                print("%dx%dx%d variant %d %s: %s" % (n, h, w, v, mode, e)); continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100): ops.conv2d(P)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 10
            if mode == "plain":
                if ref is None: ref = y.float().clone()
                err = (y.float() - ref).abs().max().item()
            else: err = float("nan")
            gf = 2.0 * n * h * w * co * 9 * ci
            print("%2dx%3dx%3d variant %d %-5s %6.1f us  %6.1f TFLOP/s  %5.2f TB/s in+out   max|diff vs first variant| %.3g" % (n, h, w, v, mode, us, gf / us * 1e-6, (2 + (mode == "bnb")) * n * h * w * 64 * 2 / us * 1e-6, err))
