"""Where are the wrong outputs of the pointwise kernel?  (debug aid)"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops, _lib
lib = _lib.load()
DEV = "cuda"
def case(cin, cout, mode, n=2, h=96, w=176, reps=3):
    P_ = n * h * w
    torch.manual_seed(1)
    x = torch.randn(P_, cin, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, device=DEV) / cin ** 0.5).to(torch.bfloat16)
    ref = (x.float() @ wt.float().t())
    r = torch.randn(P_, cout, device=DEV).to(torch.bfloat16) if mode in ("residual_stats",) else None
    if r is not None: ref = ref + r.float()
    for rep in range(reps):
        y = torch.full((P_, cout), float("nan"), device=DEV, dtype=torch.bfloat16)
        p = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=cin, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=cout, dtype=ops.dt(y), residual=r)
        st = None
        if mode != "plain":
            p.stats_partial = 1
            st = torch.zeros(max(4, ops.conv2d_stats_floats(p)), device=DEV)
            p.stats_partial = ops.ptr(st)
        assert lib.stp_conv2d_tile_for(ops.C.byref(p)) == 800
        ops.conv2d(p)
        torch.cuda.synchronize()
        d = (y.float() - ref).abs()
        bad = (d > 0.1) | torch.isnan(y.float())
        nb = int(bad.sum())
        print("%d->%d %-14s rep %d: bad %d of %d (nan %d)" % (cin, cout, mode, rep, nb, bad.numel(), int(torch.isnan(y.float()).sum())))
        if nb:
            idx = bad.nonzero()
            px, ch = idx[:, 0].cpu().numpy(), idx[:, 1].cpu().numpy()
            cols = int(lib.stp_conv2d_pw_cols(ops.C.byref(p)))
            for tp in (32, 64, 128):
                tiles = px // tp
                print("   TP %3d: %d distinct tiles, first %s; tile %% G hist top %s; pixel-in-tile min %d max %d; channels min %d max %d distinct %d" % (
                    tp, len(set(tiles)), sorted(set(tiles))[:12], np.bincount(tiles % cols).argsort()[-5:][::-1].tolist(), (px % tp).min(), (px % tp).max(), ch.min(), ch.max(), len(set(ch))))
            t0 = sorted(set(px // 64))[0]
            sub = bad[t0 * 64:(t0 + 1) * 64].cpu().numpy()
            print("   first bad 64-px block %d: bad per pixel %s" % (t0, sub.sum(1).tolist()))
            print("   bad per channel %s" % sub.sum(0).tolist())
for cin, cout in ((64, 64), (64, 256), (256, 64)):
    for mode in ("plain", "stats", "residual_stats"):
        case(cin, cout, mode)
