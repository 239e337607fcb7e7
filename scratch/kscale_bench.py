"""How much of a short-K conv launch is per-workgroup fixed cost?  Same tile, same pixels, K = 9*Cin scaled 1x/2x/4x."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (n, h, w, co, tile) in [(16, 128, 128, 64, 71), (16, 128, 128, 64, 69), (16, 64, 64, 128, 65), (16, 32, 32, 256, 70)]:
    for ci in (64, 128, 256, 512):
        x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
        wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
        y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=tile)
        us = timeit(lambda: ops.conv2d(P))
        fl = 2.0 * n * h * w * co * 9 * ci
        print("P=%6d Cout=%3d tile %3d Cin=%3d K=%4d: %7.1f us %6.1f TF" % (n * h * w, co, tile, ci, 9 * ci, us, fl / us / 1e6))
