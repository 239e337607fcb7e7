"""Round-3 probe: what the row-of-taps weight gradient and the halo forward reach when a workgroup's K loop is LONG / the grid has
many rounds (batch scaled up) - the rate a grouped per-stage launch could approach.  Existing kernels only."""
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


SHAPES = [("stage1 64->64 @128", 128, 64, 64), ("stage2 128->128 @64", 64, 128, 128), ("stage3 256->256 @32", 32, 256, 256),
          ("stage4 512->512 @16", 16, 512, 512)]
print("== row-of-taps weight gradient, batch scaled (splits automatic: steps per workgroup grow with the batch)")
for name, hw, ci, co in SHAPES:
    for n in (16, 64, 176):
        if hw >= 128 and n > 64:
            continue
        x = torch.randn(n, hw, hw, ci, device=DEV).to(torch.bfloat16)
        dy = torch.randn(n, hw, hw, co, device=DEV).to(torch.bfloat16)
        dw = torch.empty(co, 3, 3, ci, device=DEV)
        W = ops.wgrad_params(x, dy, dw, N=n, Hs0=hw, Ws0=hw, Hv=hw, Wv=hw, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=hw, Wo=hw, Cout=co, dtype=ops.BF16)
        ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
        fl = 2.0 * n * hw * hw * co * 9 * ci
        tp = timeit(lambda: ops.conv2d_wgrad_partial(W, ws, 4)); tr = timeit(lambda: ops.conv2d_wgrad_reduce(W, ws, 4))
        print("%-22s N=%3d: partial %8.1f us %7.1f TF   reduce %6.1f us   per 16 images %7.1f us" % (name, n, tp, fl / tp / 1e6, tr, (tp + tr) * 16 / n))
        del x, dy, dw, ws
print("== halo forward, batch scaled")
for name, hw, ci, co in SHAPES:
    for n in (16, 64):
        x = torch.randn(n, hw, hw, ci, device=DEV).to(torch.bfloat16)
        wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
        y = torch.zeros(n, hw, hw, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=hw, Ws0=hw, Hv=hw, Wv=hw, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=hw, Wo=hw, Cout=co, dtype=ops.BF16)
        fl = 2.0 * n * hw * hw * co * 9 * ci
        us = timeit(lambda: ops.conv2d(P))
        print("%-22s N=%3d: %8.1f us %7.1f TF   per 16 images %7.1f us" % (name, n, us, fl / us / 1e6, us * 16 / n))
        del x, wt, y
