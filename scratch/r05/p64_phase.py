"""Where the persistent 64 -> 64 kernel spends a workgroup's life (STP_TIMING + STP_TIMING_REAL build: 100 MHz counter): prologue, K loops,
slab waits, epilogues.  LIB=scratch/_exp/libstp_halo_timing.so python scratch/r05/p64_phase.py"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_training_pipeline_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("LIB", "scratch/_exp/libstp_halo_timing.so"))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
n, h, w, ci, co = 16, 128, 128, 64, 64
x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
for mode in ("plain", "stats", "bnb"):
    y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
    dbg = torch.zeros(8 * 8192, dtype=torch.int64, device=DEV)
    P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=1029, bias=dbg)
    keep = []
    if mode != "plain":
        st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV); keep.append(st); P.stats_partial = ops.ptr(st)
    if mode == "bnb":
        xb = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16); m_ = torch.zeros(co, device=DEV); r_ = torch.ones(co, device=DEV); keep += [xb, m_, r_]
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m_), ops.ptr(r_), None, None, 1
    for _ in range(3): ops.conv2d(P)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d(P); e1.record(); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)[:512].astype(np.float64)
    t = d[:, :4] / 100.0
    t0 = d[:, 4].min()
    start, end = (d[:, 4] - t0) / 100.0, (d[:, 5] - t0) / 100.0
    hw = d[:, 6].astype(np.int64)
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (((hw >> 20) & 15) << 7)       # CU id, SE id, XCC id
    print("%-5s wall %.1f us; per workgroup (4 tiles) us: prologue %.2f  K loops %.2f  slab waits %.2f  epilogues + next request %.2f  (sum %.2f)" % (
        mode, e0.elapsed_time(e1) * 1e3, t[:, 0].mean(), t[:, 1].mean(), t[:, 2].mean(), t[:, 3].mean(), t.sum(axis=1).mean()))
    print("      span %.1f us; workgroups 0..255 start %.2f..%.2f end %.1f..%.1f | 256..511 start %.2f..%.2f end %.1f..%.1f | distinct CU ids %d, CUs holding one workgroup of each half %d" % (
        end.max(), start[:256].min(), start[:256].max(), end[:256].min(), end[:256].max(), start[256:].min(), start[256:].max(), end[256:].min(), end[256:].max(),
        len(set(cu)), len(set(cu[:256]) & set(cu[256:]))))
