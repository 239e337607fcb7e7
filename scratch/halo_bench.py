"""conv_halo_kernel vs the per-tap DMA kernel on the U-Net/ResNet34 3x3 layer shapes (bs16): time per launch, TF/s, max |diff|."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
# name, n, h, w, ci, co, [old tile ids], [halo variants]
LAYERS = [("stage1 64->64 @128", 16, 128, 128, 64, 64, [71], [2, 3, 4]),
          ("stage2 128->128 @64", 16, 64, 64, 128, 128, [65], [0, 1, 2, 3]),
          ("stage3 256->256 @32", 16, 32, 32, 256, 256, [70], [0, 1, 2, 3]),
          ("stage4 512->512 @16", 16, 16, 16, 512, 512, [133], [1, 3]),
          ("dec0c2 256->256 @32", 16, 32, 32, 256, 256, [70], [1]),
          ("dec1c1d 128->384 @64", 16, 64, 64, 128, 384, [65], [0, 1]),
          ("dec2c1d 64->192 @128", 16, 128, 128, 64, 192, [71], [0, 1, 2, 4])]
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
sel = sys.argv[1:] or None
for name, n, h, w, ci, co, olds, halos in LAYERS:
    if sel and not any(s in name for s in sel): continue
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    fl = 2.0 * n * h * w * co * 9 * ci
    yref = None
    for tile in olds + [1024 + v for v in halos]:
        y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=tile)
        try:
            us = timeit(lambda: ops.conv2d(P))
        except Exception as e:
            print("%-22s tile %4d: %s" % (name, tile, e)); continue
        if yref is None: yref = y.float()
        d = (y.float() - yref).abs().max().item()
        print("%-22s tile %4d: %8.1f us %7.1f TF   max|diff| %.4f (ref max %.2f)" % (name, tile, us, fl / us / 1e6, d, yref.abs().max().item()))
