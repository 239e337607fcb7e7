"""1x1 data gradients of the bottleneck ResNets (PSPNet/ResNet101 768x768 bs8, FPN/ResNet50 1024x1024 bs4 shapes): tile variants x epilogue modes."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"; bf = torch.bfloat16
import os
LAYERS_HEAD = [("fpn head taps dgrad 64->512 @256 bs4", 4, 256, 256, 64, 512), ("psp head taps dgrad 192->512 @96 bs8", 8, 96, 96, 192, 512)]
LAYERS_EXPAND = [("fpn s1 conv1 dgrad 64->256 @256 bs4", 4, 256, 256, 64, 256), ("fpn s2 conv1 dgrad 128->512 @128 bs4", 4, 128, 128, 128, 512),
                 ("fpn s3 conv1 dgrad 256->1024 @64 bs4", 4, 64, 64, 256, 1024), ("fpn s1 conv3 fwd-like 64->256 @256 bs4", 4, 256, 256, 64, 256)]
LAYERS = [  # name, N,H,W, channels of dY, channels of dX
    ("psp s2 conv1 dgrad 128->512 @96 bs8", 8, 96, 96, 128, 512),
    ("psp s2 conv3 dgrad 512->128 @96 bs8", 8, 96, 96, 512, 128),
    ("psp s1 conv1 dgrad 64->256 @192 bs8", 8, 192, 192, 64, 256),
    ("psp s1 conv3 dgrad 256->64 @192 bs8", 8, 192, 192, 256, 64),
]
if os.environ.get("HEAD") == "1":
    LAYERS = LAYERS_HEAD
if os.environ.get("HEAD") == "2":
    LAYERS = LAYERS_EXPAND
TILES = [int(t) for t in os.environ.get("TILES", "0,65,69,70,71,133,134,97,101").split(",")]
MODES = os.environ.get("MODES", "plain,acc,bnb,acc+bnb").split(",")
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, n, h, w, co, ci in LAYERS:
    dy = torch.randn(n, h, w, co, device=DEV).to(bf)
    wb = (torch.randn(ci, co, device=DEV) / co ** 0.5).to(bf)
    x = torch.randn(n, h, w, ci, device=DEV).to(bf)
    mean, rstd = torch.zeros(ci, device=DEV), torch.ones(ci, device=DEV)
    for mode in MODES:
        for tile in TILES:
            dx = torch.zeros(n, h, w, ci, device=DEV, dtype=bf)
            P = ops.conv_params(dy, wb, dx, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=co, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=ci, dtype=ops.BF16, tile=tile,
                                accumulate0=int("acc" in mode))
            if "bnb" in mode:
                P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_relu = ops.ptr(x), ops.ptr(mean), ops.ptr(rstd), 1
                st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV)
                P.stats_partial = ops.ptr(st)
            try:
                us = timeit(lambda: ops.conv2d(P))
            except Exception as e:
                print("%-38s %-8s tile %3d: %s" % (name, mode, tile, str(e)[:60])); continue
            byt = (dy.numel() + dx.numel() * (1 + ("acc" in mode) + ("bnb" in mode))) * 2
            print("%-38s %-8s tile %3d: %8.1f us %6.2f TB/s" % (name, mode, tile, us, byt / us / 1e6))
