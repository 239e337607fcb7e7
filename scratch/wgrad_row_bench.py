"""Weight gradient of the direct 3x3 layers: buffer-DMA pixel-reduction GEMM (variant 2) vs row-of-taps kernel (variant 4),
partial + reduce, automatic splits.  STP_WGRAD_ROW_BLOCKS=<n> overrides the row kernel's workgroup target."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
LAYERS = [("stage1 64->64 @128", 16, 128, 128, 64, 64), ("stage2 128->128 @64", 16, 64, 64, 128, 128),
          ("stage3 256->256 @32", 16, 32, 32, 256, 256), ("stage4 512->512 @16", 16, 16, 16, 512, 512)]
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, n, h, w, ci, co in LAYERS:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    dy = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
    dw = torch.empty(co, 3, 3, ci, device=DEV)
    W = ops.wgrad_params(x, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16)
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
    ref = None
    fl = 2.0 * n * h * w * co * 9 * ci
    for variant in (2, 4):
        ops.conv2d_wgrad_partial(W, ws, variant); ops.conv2d_wgrad_reduce(W, ws, variant); torch.cuda.synchronize()
        if ref is None: ref = dw.clone()
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        tp = timeit(lambda: ops.conv2d_wgrad_partial(W, ws, variant)); tr = timeit(lambda: ops.conv2d_wgrad_reduce(W, ws, variant))
        print("%-22s variant %d: partial %7.1f us %7.1f TF  reduce %6.1f us  sum %7.1f us  rel maxdiff %.2g" % (name, variant, tp, fl / tp / 1e6, tr, tp + tr, err))
