"""Per-launch fixed cost of the DMA conv kernel: a 1x1 conv with K = 64 (one K-tile) at the layer shapes of the U-Net."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def empty_kernel():
    a = torch.empty(4, device=DEV)
    return timeit(lambda: ops.add_inplace(a, a, 4) if hasattr(ops, "add_inplace") else None)
for (n, h, w, co, tile) in [(16, 64, 64, 128, 65), (16, 128, 128, 64, 71), (16, 32, 32, 256, 70), (16, 16, 16, 512, 133)]:
    for ci, k in ((64, 1), (64, 3), (128, 3)):
        x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
        wt = (torch.randn(co, k, k, ci, device=DEV) / (k * k * ci) ** 0.5).to(torch.bfloat16)
        y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=1, pad=k // 2, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=tile)
        us = timeit(lambda: ops.conv2d(P))
        print("P=%6d Cout=%3d tile %3d K=%4d: %6.1f us" % (n * h * w, co, tile, k * k * ci, us))
