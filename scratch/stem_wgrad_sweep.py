"""conv0 weight gradient (7x7/2, 4 -> 64 channels, 16x512x512): partial + reduce time vs splits."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
n, h, w, co = 16, 512, 512, 64
x = torch.randn(n, h, w, 4, device=DEV).to(torch.bfloat16)
dy = torch.randn(n, 256, 256, co, device=DEV).to(torch.bfloat16)
dw = torch.empty(co, 7, 8, 4, device=DEV)
for splits in (0, 64, 96, 128, 192, 256, 320, 384, 512, 640, 768, 1024):
    W = ops.wgrad_params(x, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=4, KH=7, KW=8, stride=2, pad=3, Ho=256, Wo=256, Cout=co, dtype=ops.BF16, splits=splits)
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
    t = []
    for fn in (lambda: ops.conv2d_wgrad_partial(W, ws, 0), lambda: ops.conv2d_wgrad_reduce(W, ws, 0)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e3 / 20)
    print("splits %4d: partial %6.1f us  reduce %5.1f us  slabs %5.1f MB" % (splits, t[0], t[1], ws.numel() * 4 / 1e6))
