"""Wide-output small-channel data gradient (decoder_stage3_conv1 dgrad, 16 x 256 x 256, 32 -> 64 summed + 64 skip): HIP-event time.
EXP=<n> loads scratch/_exp/libstp_sc_exp<n>.so (31 = no output stores, 32 = no halo loads, 33 = no LDS reads / MFMAs)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from segmentation_training_pipeline_amd import _lib
exp = os.environ.get("EXP")
if exp:
    _lib.LIB_PATH = os.path.join(ROOT, "scratch/_exp/libstp_sc_exp%s.so" % exp)
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
n, h, w = 16, 256, 256
dy = torch.randn(n, h, w, 32, device=DEV).to(torch.bfloat16)
wt = (torch.randn(128, 3, 3, 32, device=DEV) / 17).to(torch.bfloat16)
up = torch.empty(n, h // 2, w // 2, 64, device=DEV, dtype=torch.bfloat16)
sk = torch.empty(n, h, w, 64, device=DEV, dtype=torch.bfloat16)
xb = torch.randn(n, h // 2, w // 2, 64, device=DEV).to(torch.bfloat16)
f32 = lambda k: torch.rand(k, device=DEV) + 0.5
m, r, g, b = f32(64), f32(64), f32(64), f32(64)
for feat in ("plain", "bnb", "acc"):
    P = ops.conv_params(dy, wt, up, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=32, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=128, dtype=ops.BF16,
                        dst1=sk, Cd0=64, accumulate0=int(feat == "acc"), accumulate1=int(feat == "acc"))
    P.dst_sum2x2 = 1
    if feat == "bnb":
        st = torch.empty(max(4, ops.conv2d_stats_floats(P)), device=DEV)
        P.stats_partial = ops.ptr(st)
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    us = timeit(lambda: ops.conv2d(P))
    print("EXP=%s scw dgrad 32 -> 64+64 @256 %-6s %8.1f us  (tile %d)" % (exp, feat, us, _lib.load().stp_conv2d_tile_for(ops.C.byref(P))))
