"""Small-channel (full-resolution decoder tail) convolution microbenchmark: HIP-event time and effective HBM rate per layer
shape of the U-Net/ResNet34 512x512 bs16 step.  EXP=<n> loads the what-if build scratch/_exp/libstp_sc_exp<n>.so
(scratch/sc_exp_build.sh: 11 = no output stores, 12 = no halo loads, 13 = no LDS reads / MFMAs)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from segmentation_training_pipeline_amd import _lib
exp = os.environ.get("EXP")
if exp:
    _lib.LIB_PATH = os.path.join(ROOT, "scratch/_exp/libstp_sc_exp%s.so" % exp)
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
# name, N, H, W, Cin, Cout, upsampled source
LAYERS = [("dec4c2 16->16 @512", 16, 512, 512, 16, 16, 0), ("dec4c1 32->16 @512 up", 16, 512, 512, 32, 16, 1), ("dec3c2 32->32 @256", 16, 256, 256, 32, 32, 0),
          ("final 16->1 @512", 16, 512, 512, 16, 1, 0), ("final dgrad 8->16 @512", 16, 512, 512, 8, 16, 0), ("dec4c1 dgrad 16->32 @512", 16, 512, 512, 16, 32, 0)]
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
a = torch.empty(134217728 // 2, device=DEV, dtype=torch.bfloat16); b = torch.empty_like(a)
us = timeit(lambda: b.copy_(a)); print("copy 134 MB -> 134 MB: %.1f us = %.2f TB/s" % (us, 2 * 134.2 / us))
ONLY = os.environ.get("ONLY")      # ONLY=<feature>: just that variant of the 16->16 @512 layer (PMC runs)
for name, n, h, w, ci, co, up in ([] if ONLY else LAYERS):
    hs, ws = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, hs, ws, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(max(co, 16), 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
    P = ops.conv_params(x, wt, y, N=n, Hs0=hs, Ws0=ws, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16,
                        mode=ops.SRC_NEAREST2X if up else ops.SRC_DIRECT)
    us = timeit(lambda: ops.conv2d(P))
    mb = (x.numel() + y.numel()) * 2 / 1e6
    print("EXP=%s %-26s %8.1f us  %6.1f MB  %5.2f TB/s" % (exp, name, us, mb, mb / us))
# feature cost on the 16->16 @512 layer: fused producer BatchNormalization (src_bn_*), fused statistics, BN-backward epilogue
import ctypes as C
n, h, w, ci, co = 16, 512, 512, 16, 16
x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
wt = (torch.randn(16, 3, 3, ci, device=DEV) / 12).to(torch.bfloat16)
y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
xb = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
f32 = lambda k: torch.rand(k, device=DEV) + 0.5
m, r, g, b = f32(16), f32(16), f32(16), f32(16)
for feat in ((ONLY,) if ONLY else ("plain", "pbn", "stats", "pbn+stats", "bnb", "pbn+bnb")):
    P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16)
    if "pbn" in feat:
        P.src_bn_mean, P.src_bn_rstd, P.src_bn_gamma, P.src_bn_beta, P.src_bn_relu = ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    if "stats" in feat or "bnb" in feat:
        st = torch.empty(max(4, ops.conv2d_stats_floats(P)), device=DEV)
        P.stats_partial = ops.ptr(st)
    if "bnb" in feat:
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    us = timeit(lambda: ops.conv2d(P))
    print("EXP=%s 16->16 @512 %-10s %8.1f us" % (exp, feat, us))
