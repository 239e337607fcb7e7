"""1x1 bottleneck convolutions (ResNet50/101): tile variants, with and without residual + fused stats."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
LAYERS = [  # name, N,H,W,Cin,Cout
    ("expand 64->256 @256 bs4", 4, 256, 256, 64, 256),
    ("reduce 256->64 @256 bs4", 4, 256, 256, 256, 64),
    ("expand 128->512 @128 bs4", 4, 128, 128, 128, 512),
    ("reduce 512->128 @128 bs4", 4, 128, 128, 512, 128),
    ("expand 256->1024 @64 bs4", 4, 64, 64, 256, 1024),
]
TILES = [65, 69, 70, 71, 133, 134, 101, 97]
for name, n, h, w, ci, co in LAYERS:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 1, 1, ci, device=DEV) / ci ** 0.5).to(torch.bfloat16)
    res = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
    for mode in ("plain", "res+stats"):
        for tile in TILES:
            y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
            P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=co,
                                dtype=ops.BF16, tile=tile, residual=(res if mode != "plain" else None))
            if mode != "plain":
                st = torch.empty(max(4, ops.conv2d_stats_floats(P)), dtype=torch.float32, device=DEV)
                P.stats_partial = ops.ptr(st)
            try:
                ops.conv2d(P)
            except Exception as e:
                print("%-26s %-9s tile %3d: %s" % (name, mode, tile, e)); continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv2d(P)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            byt = (x.numel() + y.numel() * (2 if mode != "plain" else 1)) * 2
            print("%-26s %-9s tile %3d: %8.1f us %7.1f TF %6.2f TB/s" % (name, mode, tile, us, 2.0 * n * h * w * co * ci / us / 1e6, byt / us / 1e6))
