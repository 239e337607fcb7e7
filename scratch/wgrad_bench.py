"""Weight-gradient microbenchmark: representative layers x kernel variants."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
LAYERS = [("stage1 64->64 @128", 16, 128, 128, 64, 64, 3), ("stage2 128->128 @64", 16, 64, 64, 128, 128, 3),
          ("stage3 256->256 @32", 16, 32, 32, 256, 256, 3), ("stage4 512->512 @16", 16, 16, 16, 512, 512, 3),
          ("dec1c1 384->128 @64", 16, 64, 64, 384, 128, 3), ("dec3c2 32->32 @256", 16, 256, 256, 32, 32, 3),
          ("dec4c2 16->16 @512", 16, 512, 512, 16, 16, 3)]
for name, n, h, w, ci, co, k in LAYERS:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    dy = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
    dw = torch.empty(co, k, k, ci, device=DEV)
    ref = None
    for splits in (0, 4, 8, 16, 32):
        W = ops.wgrad_params(x, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=1, pad=1, Ho=h, Wo=w, Cout=co,
                             dtype=ops.BF16, splits=splits)
        ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
        for variant in (2, 3):
            ops.conv2d_wgrad_partial(W, ws, variant); ops.conv2d_wgrad_reduce(W, ws, variant); torch.cuda.synchronize()
            if ref is None: ref = dw.clone()
            err = (dw - ref).abs().max().item()
            t = []
            for fn in (lambda: ops.conv2d_wgrad_partial(W, ws, variant), lambda: ops.conv2d_wgrad_reduce(W, ws, variant)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                t.append(e0.elapsed_time(e1) * 1e3 / 20)
            fl = 2.0 * n * h * w * co * k * k * ci
            print("%-22s splits %2d variant %d: partial %8.1f us %7.1f TF  reduce %6.1f us  ws %6.1f MB  maxdiff %.3g" % (name, splits, variant, t[0], fl / t[0] / 1e6, t[1], ws.numel() * 4 / 1e6, err))
