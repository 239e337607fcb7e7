"""What-if timing of a write-bound 1x1 expand convolution (64 -> 256 @ 4x256x256): EXP = none | 1 (no MFMA) | 4 (no stores) | 5 (empty)."""
import sys, os, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import _lib
exp = os.environ.get("EXP")
if exp:
    _lib.LIB_PATH = os.path.abspath("scratch/_exp/libstp_exp%s.so" % exp)
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, n, h, w, ci, co, k in [("expand 64->256", 4, 256, 256, 64, 256, 1), ("reduce 256->64", 4, 256, 256, 256, 64, 1), ("3x3 64->64 @128 bs16", 16, 128, 128, 64, 64, 3)]:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, k, k, ci, device=DEV) / (k * k * ci) ** 0.5).to(torch.bfloat16)
    for tile in (65, 70, 71, 69):
        y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=1, pad=k // 2, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=tile)
        try:
            us = timeit(lambda: ops.conv2d(P))
        except Exception as e:
            continue
        print("EXP=%s %-22s tile %3d: %8.1f us" % (exp, name, tile, us))
