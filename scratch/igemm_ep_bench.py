"""Epilogue cost of the generic uniform-tap kernel (conv_igemm_ut): the same launch plain / accumulating / with the fused
BatchNormalization-backward sums / with fused statistics, on a stride-2 data gradient and on bottleneck 1x1 shapes."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def bf(*s): return torch.randn(*s, device=DEV).to(torch.bfloat16)
def run(name, mk, co, variants):
    for v in variants:
        P, keep = mk(v)
        us = timeit(lambda: ops.conv2d(P))
        print("%-34s %-12s tile %4d  %8.1f us" % (name, v, ops._lib.load().stp_conv2d_tile_for(ops.C.byref(P)), us))
# (a) stride-2 3x3 data gradient: dY [16,64,64,128] -> dX [16,128,128,64]
def mk_s2(v, n=16, h=64, cin=128, cout=64):
    dy = bf(n, h, h, cin); w = bf(cout, 3, 3, cin); dx = bf(n, 2 * h, 2 * h, cout)
    P = ops.conv_params(dy, w, dx, N=n, Hs0=h, Ws0=h, Hv=2 * h - 1, Wv=2 * h - 1, C0=cin, mode=ops.SRC_ZEROINS2X, KH=3, KW=3, stride=1, pad=1,
                        Ho=2 * h, Wo=2 * h, Cout=cout, dtype=ops.BF16, accumulate0=int(v != "plain"))
    keep = [dy, w, dx]
    if "bnb" in v:
        x = bf(n, 2 * h, 2 * h, cout); m = torch.zeros(cout, device=DEV); r = torch.ones(cout, device=DEV)
        st = torch.zeros(max(4, ops.conv2d_stats_floats(P)) * 4, device=DEV)
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_relu, P.stats_partial = ops.ptr(x), ops.ptr(m), ops.ptr(r), 1, ops.ptr(st)
        keep += [x, m, r, st]
    if "fold" in v:
        fs = bf(n, h, h, cin); fw = bf(cout, cin)
        P.fold_src, P.fold_weight, P.fold_C = ops.ptr(fs), ops.ptr(fw), cin
        keep += [fs, fw]
    return P, keep
for h, cin, cout in ((64, 128, 64), (32, 256, 128), (16, 512, 256)):
    run("s2 dgrad %d->%d @%d" % (cin, cout, 2 * h), lambda v: mk_s2(v, 16, h, cin, cout), cout, ["plain", "acc", "acc+bnb", "acc+bnb+fold"])
# (b) 1x1 convolutions (ResNet50 bottleneck shapes at 4 x 256 x 256 / 4 x 128 x 128)
def mk_1x1(v, n, h, cin, cout):
    x = bf(n, h, h, cin); w = bf(cout, 1, 1, cin); y = bf(n, h, h, cout)
    P = ops.conv_params(x, w, y, N=n, Hs0=h, Ws0=h, Hv=h, Wv=h, C0=cin, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=h, Cout=cout, dtype=ops.BF16)
    keep = [x, w, y]
    if v in ("stats", "bnb"):
        st = torch.zeros(max(4, ops.conv2d_stats_floats(P)) * 4, device=DEV); P.stats_partial = ops.ptr(st); keep.append(st)
    if v == "bnb":
        xb = bf(n, h, h, cout); m = torch.zeros(cout, device=DEV); r = torch.ones(cout, device=DEV)
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_relu = ops.ptr(xb), ops.ptr(m), ops.ptr(r), 1; keep += [xb, m, r]
    if v == "res":
        rs = bf(n, h, h, cout); P.residual = ops.ptr(rs); keep.append(rs)
    return P, keep
for n, h, cin, cout in ((4, 256, 64, 256), (4, 256, 256, 64), (4, 128, 512, 128), (4, 128, 128, 512), (16, 128, 64, 64)):
    run("1x1 %d->%d @%dx%d^2" % (cin, cout, n, h), lambda v: mk_1x1(v, n, h, cin, cout), cout, ["plain", "stats", "bnb", "res"])
