"""Weight-gradient split sweep over the U-Net/ResNet34 layer shapes: partial + reduce time vs number of splits."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
LAYERS = [("stage1 64->64 @128", 128, 64, 64), ("stage2 128->128 @64", 64, 128, 128), ("stage3 256->256 @32", 32, 256, 256),
          ("stage4 512->512 @16", 16, 512, 512), ("d0c1 768->256 @32", 32, 768, 256), ("d1c1 384->128 @64", 64, 384, 128),
          ("d2c1 192->64 @128", 128, 192, 64), ("d1c2 128->128 @64", 64, 128, 128), ("s3u1 128->256 @32", 32, 128, 256)]
for name, hw, ci, co in LAYERS:
    n, k = 16, 3
    x = torch.randn(n, hw, hw, ci, device=DEV).to(torch.bfloat16)
    dy = torch.randn(n, hw, hw, co, device=DEV).to(torch.bfloat16)
    dw = torch.empty(co, k, k, ci, device=DEV)
    bm, bn = (64, 128) if co <= 64 else (128, 128)
    tiles = -(-co // bm) * -(-(9 * ci) // bn)
    cands = sorted(set([0] + [max(1, t // tiles) for t in (256, 320, 384, 448, 512)] + [-(-t // tiles) for t in (256, 512, 768)]))
    res = []
    for splits in cands:
        W = ops.wgrad_params(x, dy, dw, N=n, Hs0=hw, Ws0=hw, Hv=hw, Wv=hw, C0=ci, KH=k, KW=k, stride=1, pad=1, Ho=hw, Wo=hw, Cout=co,
                             dtype=ops.BF16, splits=splits)
        ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
        v = 2
        ops.conv2d_wgrad_partial(W, ws, v); ops.conv2d_wgrad_reduce(W, ws, v); torch.cuda.synchronize()
        t = []
        for fn in (lambda: ops.conv2d_wgrad_partial(W, ws, v), lambda: ops.conv2d_wgrad_reduce(W, ws, v)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 1e3 / 20)
        res.append((splits, t[0], t[1]))
    print("%-22s tiles %3d: " % (name, tiles) + "  ".join("S=%d(%d blk) %.0f+%.0f" % (s, s * tiles, a, b) for s, a, b in res))
