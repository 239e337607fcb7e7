"""Timeline of the 64-channel halo launches (stage 1: 64 -> 64 @ 128 x 128, 1024 workgroups, two co-resident per CU): per-XCD shader-clock
stamps (STP_TIMING build) - when do the workgroups start / enter the loop / leave it / finish, i.e. are the co-resident pairs in lockstep?
LIB=scratch/_exp/libstp_halo_timing.so python scratch/r05/halo_phase.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_training_pipeline_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LIB", "scratch/_exp/libstp_halo_timing.so"))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
n, h, w, ci, co = 16, 128, 128, 64, 64
x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
for v in [int(a) for a in os.environ.get("VARS", "2,3,4").split(",")]:
    for mode in os.environ.get("MODES", "stats,bnb").split(","):
        y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        dbg = torch.zeros(4 * 8192, dtype=torch.int64, device=DEV)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=1024 + v, bias=dbg)
        st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV)
        P.stats_partial = ops.ptr(st)
        if mode == "bnb":
            xb = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16); m_ = torch.zeros(co, device=DEV); r_ = torch.ones(co, device=DEV)
            P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m_), ops.ptr(r_), None, None, 1
        for _ in range(3): ops.conv2d(P)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d(P); e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 4)
        nz = (d != 0).all(axis=1)
        ids = np.arange(len(d))[nz]
        d = d[nz].astype(np.float64)
        print("variant %d %s: %d workgroups, wall %.1f us" % (v, mode, len(d), e0.elapsed_time(e1) * 1e3))
        t0 = d[:, 0].min()
        g = (d - t0) / 100.0            # us (STP_TIMING_REAL build: 100 MHz counter shared by the chip)
        span = g[:, 3].max()
        pro, loop, epi = g[:, 1] - g[:, 0], g[:, 2] - g[:, 1], g[:, 3] - g[:, 2]
        print("  span %.1f us; per workgroup us: prologue %.2f loop %.2f epilogue %.2f life %.2f" % (span, pro.mean(), loop.mean(), epi.mean(), (g[:, 3] - g[:, 0]).mean()))
        bins = np.linspace(0, span, 25)
        for name, a, b in (("prologue", 0, 1), ("loop", 1, 2), ("epilogue", 2, 3)):
            occ = [int(((g[:, a] <= (lo + hi) / 2) & (g[:, b] > (lo + hi) / 2)).sum()) for lo, hi in zip(bins[:-1], bins[1:])]
            print("  in %-8s %s" % (name, " ".join("%4d" % o for o in occ)))
        for lo_id in (0, 256, 512, 768):
            sel = (ids >= lo_id) & (ids < lo_id + 256)
            if sel.any():
                print("  workgroups %4d..%4d: start %.1f..%.1f us, end %.1f..%.1f us, life %.1f us (prologue %.2f loop %.2f epilogue %.2f)" % (
                    lo_id, lo_id + 255, g[sel, 0].min(), g[sel, 0].max(), g[sel, 3].min(), g[sel, 3].max(), (g[sel, 3] - g[sel, 0]).mean(),
                    pro[sel].mean(), loop[sel].mean(), epi[sel].mean()))
