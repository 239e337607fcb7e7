"""64 -> 64 channel 3x3 layer of ResNet stage 1 (16 x 128 x 128): the weights-in-registers kernel (default) vs the halo kernel (tile 1026)."""
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
n, h, w = int(os.environ.get("NB", "16")), 128, 128
x = torch.randn(n, h, w, 64, device=DEV).to(torch.bfloat16)
wt = (torch.randn(64, 3, 3, 64, device=DEV) / 24).to(torch.bfloat16)
y = torch.empty(n, h, w, 64, device=DEV, dtype=torch.bfloat16)
rs = torch.randn(n, h, w, 64, device=DEV).to(torch.bfloat16)
f32 = lambda k: torch.rand(k, device=DEV) + 0.5
m, r, g, b = f32(64), f32(64), f32(64), f32(64)
for tile in ((736,) if exp else (736, 1026)):
    for feat in ("plain", "stats", "res+stats", "bnb"):
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=64, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=64, dtype=ops.BF16,
                            residual=rs if feat == "res+stats" else None, tile=tile)
        if feat != "plain":
            if feat == "bnb":
                P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(rs), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
            st = torch.empty(max(4, ops.conv2d_stats_floats(P)), device=DEV)
            P.stats_partial = ops.ptr(st)
        us = timeit(lambda: ops.conv2d(P))
        print("EXP=%s " % exp + "conv 64 -> 64 @128 %-10s %8.1f us  (tile %d)  %.0f TFLOP/s" % (feat, us, _lib.load().stp_conv2d_tile_for(ops.C.byref(P)), 19.33e9 * n / 16 / us / 1e6))
