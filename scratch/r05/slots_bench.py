"""Cost of the fixed-point slot form of the halo kernel's statistics (atomic int64 adds into [2][C][slots]) against the float table
[2][C][tiles]: event-timed launches on the shapes whose tables have more than 128 columns."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
for (n, h, w, ci, co, v) in [(16, 128, 128, 64, 64, 2), (16, 64, 64, 128, 128, 0), (16, 64, 64, 128, 128, 1), (16, 32, 32, 256, 256, 1)]:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    ref = None
    for slots in (0, 16, 64):
        y = torch.zeros(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=1024 + v)
        if slots:
            st = torch.zeros(2 * co * slots, dtype=torch.int64, device=DEV)
            P.stats_slots = slots
        else:
            st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV)
        P.stats_partial = ops.ptr(st)
        for _ in range(5): ops.conv2d(P)
        torch.cuda.synchronize()
        if slots: st.zero_()
        ops.conv2d(P); torch.cuda.synchronize()
        if slots:
            sums = st.view(2, co, slots).sum(dim=2).double() / 16777216.0
        else:
            sums = st.view(2, co, -1).double().sum(dim=2)
        if ref is None: ref = sums
        err = ((sums - ref).abs() / (ref.abs() + 1.0)).max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): ops.conv2d(P)
        e1.record(); torch.cuda.synchronize()
        print("%2dx%3dx%3d %3d->%3d variant %d  %s  %6.1f us   max rel diff of the channel sums vs the float table %.2e" % (
            n, h, w, ci, co, v, ("slots %2d" % slots) if slots else "float table", e0.elapsed_time(e1) * 10, err))
