"""conv0 (7x7 / stride 2, 4 padded channels -> 64, 16 x 512 x 512) forward with fused statistics: HIP-event time of the stem kernel."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from segmentation_training_pipeline_amd import _lib, ops
DEV = "cuda"
n, h, w = 16, 512, 512
x = torch.randn(n, h, w, 4, device=DEV).to(torch.bfloat16)
wt = (torch.randn(64, 7, 8, 4, device=DEV) / 12).to(torch.bfloat16)
y = torch.empty(n, h // 2, w // 2, 64, device=DEV, dtype=torch.bfloat16)
P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=4, KH=7, KW=8, stride=2, pad=3, Ho=h // 2, Wo=w // 2, Cout=64, dtype=ops.BF16)
st = torch.empty(max(4, ops.conv2d_stats_floats(P)), device=DEV); P.stats_partial = ops.ptr(st)
def timeit(fn, k=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k
us = timeit(lambda: ops.conv2d(P))
print("stem 16x512x512 fwd+stats %.1f us  (%.0f MB -> %.2f TB/s)" % (us, (x.numel() + y.numel()) * 2 / 1e6, (x.numel() + y.numel()) * 2 / 1e6 / us))
