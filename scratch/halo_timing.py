"""Where a conv_halo_kernel launch spends its time: shader-clock stamps (STP_TIMING build) at entry / loop start / loop end /
stores drained, per workgroup, plus the launch's wall time.  LIB=scratch/_exp/libstp_halo_timing.so python scratch/halo_timing.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from segmentation_training_pipeline_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LIB", "scratch/_exp/libstp_halo_timing.so"))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
MODES = os.environ.get("MODES", "plain,stats,res,bnb").split(",")
LAYERS = [("stage1 64->64 @128", 16, 128, 128, 64, 64, [2, 3]), ("stage2 128->128 @64", 16, 64, 64, 128, 128, [0, 1]),
          ("stage3 256->256 @32", 16, 32, 32, 256, 256, [1]), ("stage4 512->512 @16", 16, 16, 16, 512, 512, [3])]
for name, n, h, w, ci, co, halos in LAYERS:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    for v, mode in [(v, m) for v in halos for m in MODES]:
        y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        dbg = torch.zeros(4 * 8192, dtype=torch.int64, device=DEV)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16,
                            tile=1024 + v, bias=dbg)
        keep = []
        if mode in ("stats", "bnb", "res"):
            st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV); keep.append(st)
            P.stats_partial = ops.ptr(st)
        if mode == "res":
            r = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16); keep.append(r); P.residual = ops.ptr(r)
        if mode == "bnb":
            xb = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16); m_ = torch.zeros(co, device=DEV); r_ = torch.ones(co, device=DEV)
            keep += [xb, m_, r_]
            P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m_), ops.ptr(r_), None, None, 1
        name = "%s %s" % (name.split(" @")[0][:12], mode)
        for _ in range(3): ops.conv2d(P)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d(P); e1.record(); torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 4)
        nz = (d != 0).all(axis=1)
        key = (np.arange(len(d)) % 8)[nz]      # workgroup ids round-robin over the 8 XCDs; every XCD has its own counter
        d = d[nz].astype(np.float64)
        wall = e0.elapsed_time(e1) * 1e3
        spans, firsts, lasts = [], [], []
        for kx in np.unique(key):
            g = d[key == kx]
            spans.append(g[:, 3].max() - g[:, 0].min()); firsts.append(g[:, 0].max() - g[:, 0].min()); lasts.append(g[:, 3].max() - g[:, 3].min())
        pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
        print("%-22s var %d wgs %4d wall %5.1f us | XCD span %6.0f ticks (start skew %5.0f, end skew %5.0f) | per WG ticks: prologue %5.0f  loop %6.0f (min %6.0f max %6.0f)  epilogue %5.0f (max %5.0f)"
              % (name, v, len(d), wall, np.mean(spans), np.mean(firsts), np.mean(lasts), pro.mean(), loop.mean(), loop.min(), loop.max(), epi.mean(), epi.max()))
