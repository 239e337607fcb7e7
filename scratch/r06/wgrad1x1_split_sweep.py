"""Lone 1x1 weight gradients of configs[3] / [4]: partial + reduce time against the number of pixel splits (p->splits), operands rotated
through > 600 MB of buffers so that no launch finds its inputs in the last-level cache."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
# name, N, H (of the output), Cin, Cout, stride
LAYERS = [("fpn s1 64->256 @256", 4, 256, 64, 256, 1), ("fpn s1 256->64 @256", 4, 256, 256, 64, 1),
          ("fpn s2 128->512 @128", 4, 128, 128, 512, 1), ("fpn s2 512->128 @128", 4, 128, 512, 128, 1),
          ("fpn s3 256->1024 @64", 4, 64, 256, 1024, 1), ("fpn s3 1024->256 @64", 4, 64, 1024, 256, 1),
          ("fpn s4 512->2048 @32", 4, 32, 512, 2048, 1), ("fpn s4 2048->512 @32", 4, 32, 2048, 512, 1),
          ("fpn s4 sc 1024->2048 s2", 4, 32, 1024, 2048, 2), ("fpn s3 sc 512->1024 s2", 4, 64, 512, 1024, 2),
          ("fpn lat 512->256 @128", 4, 128, 512, 256, 1), ("fpn lat 256->256 @256", 4, 256, 256, 256, 1),
          ("psp s2 128->512 @96", 8, 96, 128, 512, 1), ("psp s2 512->128 @96", 8, 96, 512, 128, 1),
          ("psp s1 64->256 @192", 8, 192, 64, 256, 1), ("psp s1 256->64 @192", 8, 192, 256, 64, 1)]
for name, n, ho, ci, co, stride in LAYERS:
    hs = ho * stride
    per = n * hs * hs * ci * 2 + n * ho * ho * co * 2
    nset = max(2, min(16, (600 << 20) // per + 1))
    xs = [torch.randn(n, hs, hs, ci, device=DEV).to(torch.bfloat16) for _ in range(nset)]
    dys = [torch.randn(n, ho, ho, co, device=DEV).to(torch.bfloat16) for _ in range(nset)]
    dw = torch.empty(co, 1, 1, ci, device=DEV)
    bm, bn = (64, 128) if co <= 64 else (128, 128)
    tiles = -(-co // bm) * -(-ci // bn)
    nsteps = n * ho * ho // 64
    cands = [0] + sorted(set(max(1, min(nsteps, t // tiles)) for t in (64, 128, 192, 256, 384, 512)))
    res = []
    for splits in cands:
        Ws = [ops.wgrad_params(x, dy, dw, N=n, Hs0=hs, Ws0=hs, Hv=hs, Wv=hs, C0=ci, KH=1, KW=1, stride=stride, pad=0, Ho=ho, Wo=ho, Cout=co,
                               dtype=ops.BF16, splits=splits) for x, dy in zip(xs, dys)]
        ws = torch.empty(ops.wgrad_workspace_bytes(Ws[0]) // 4 + 4, dtype=torch.float32, device=DEV)
        for W in Ws[:2]:
            ops.conv2d_wgrad_partial(W, ws, 0); ops.conv2d_wgrad_reduce(W, ws, 0)
        torch.cuda.synchronize()
        # partial + reduce back to back as in the step (the reduce finds the slabs where the partial launch left them)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 * nset
        e0.record()
        for r in range(reps):
            W = Ws[r % nset]
            ops.conv2d_wgrad_partial(W, ws, 0); ops.conv2d_wgrad_reduce(W, ws, 0)
        e1.record(); torch.cuda.synchronize()
        res.append((splits, e0.elapsed_time(e1) * 1e3 / reps))
    floor = (per + co * ci * 4) / 6.3e6
    print("%-26s tiles %3d steps %5d floor %5.1f us: " % (name, tiles, nsteps, floor) +
          "  ".join("S=%d(%d wg) %.1f" % (s, s * tiles, t) for s, t in res), flush=True)
