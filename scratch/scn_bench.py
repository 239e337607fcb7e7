"""Narrow-output small-channel forward (decoder_stage3_conv1, 16 x 256 x 256, 64 up + 64 skip -> 32): HIP-event time, new kernel vs the
generic per-tap kernel (STP_SCN=0 in a second process)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from segmentation_training_pipeline_amd import _lib, ops
DEV = "cuda"
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
n, h, w = 16, 256, 256
x = torch.randn(n, h // 2, w // 2, 64, device=DEV).to(torch.bfloat16)
sk = torch.randn(n, h, w, 64, device=DEV).to(torch.bfloat16)
wt = (torch.randn(32, 3, 3, 128, device=DEV) / 34).to(torch.bfloat16)
y = torch.empty(n, h, w, 32, device=DEV, dtype=torch.bfloat16)
for feat in ("plain", "stats"):
    P = ops.conv_params(x, wt, y, N=n, Hs0=h // 2, Ws0=w // 2, Hv=h, Wv=w, C0=64, C1=64, src1=sk, mode=ops.SRC_NEAREST2X, KH=3, KW=3, stride=1, pad=1,
                        Ho=h, Wo=w, Cout=32, dtype=ops.BF16)
    if feat == "stats":
        st = torch.empty(max(4, ops.conv2d_stats_floats(P)), device=DEV)
        P.stats_partial = ops.ptr(st)
    us = timeit(lambda: ops.conv2d(P))
    print("conv 64 up + 64 skip -> 32 @256 %-6s %8.1f us  (tile %d)  %.0f TFLOP/s" % (feat, us, _lib.load().stp_conv2d_tile_for(ops.C.byref(P)), 77.3e9 / us / 1e6))
