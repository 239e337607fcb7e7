"""Winograd F(2x2, 3x3) scout for the 256-channel 3x3 layers (review item 7): numerical error of a bf16 MFMA implementation against an
exact float64 convolution, next to the direct bf16 convolution the halo kernel computes.  CPU only (torch fp32 / fp64 emulation of
the 16-bit roundings: transformed inputs and transformed weights must be rounded to the MFMA operand format; accumulation fp32).
usage: python scratch/r06/winograd_scout.py"""
import numpy as np, torch
torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def rnd(t, dt):
    return t.to(dt).to(torch.float32) if dt is not None else t


def winograd(x, w, dt, acc=torch.float32):
    """x [N,C,H,W] (already in storage precision), w [O,C,3,3] fp32 master; pad 1; H, W even."""
    n, c, h, wd = x.shape
    o = w.shape[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    # tiles: 4x4 windows at stride 2
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                    # [N,C,H/2,W/2,4,4]
    V = torch.einsum("ij,ncabjk,lk->ncabil", BT.to(acc), t.to(acc), BT.to(acc))      # B^T d B in fp32 (exact-ish: adds only)
    V = rnd(V.float(), dt)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w.double(), G)     # G g G^T from the fp32 master, once per step
    U = rnd(U.float(), dt)
    M = torch.einsum("ncabil,ocil->noabil", V.to(acc), U.to(acc))                    # 16 GEMMs over C, fp32 accumulate
    Y = torch.einsum("ij,noabjk,lk->noabil", AT.to(acc), M, AT.to(acc))              # [N,O,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, o, h, wd)


def report(name, C, O, HW, n=2):
    # input: BatchNormalization + ReLU output (unit-variance pre-activation, half zeros); weights he_uniform
    x = torch.relu(torch.randn(n, C, HW, HW))
    lim = np.sqrt(6.0 / (9 * C))
    w = (torch.rand(O, C, 3, 3) * 2 - 1) * lim
    exact = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    scale = exact.abs().max().item()
    rms = exact.pow(2).mean().sqrt().item()
    for dt, nm in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        xs = rnd(x, dt)
        ex_s = torch.nn.functional.conv2d(xs.double(), rnd(w, dt).double(), padding=1)         # what an exact machine gets from the rounded operands
        direct = torch.nn.functional.conv2d(xs, rnd(w, dt), padding=1)                          # the halo kernel: rounded operands, fp32 accumulate
        wino = winograd(xs, w, dt)
        wino_noround = winograd(xs, w, None)                                                     # transforms kept in fp32 (not MFMA-feedable): the algorithm's own fp32 error
        ulp = 2.0 ** (np.floor(np.log2(scale)) - (7 if nm == "bf16" else 10))
        def e(t, ref): return (t.double() - ref).pow(2).mean().sqrt().item()
        print("%-22s %s  out rms %.3f max %.2f  storage ulp(top) %.4g | rms error vs EXACT conv of the stored operands: direct %.3g  winograd %.3g (%.1fx)"
              "  winograd fp32 transforms %.3g | vs the fp64 conv of the UNROUNDED operands: direct %.3g  winograd %.3g | after the output rounding: direct %.3g  winograd %.3g"
              % (name, nm, rms, scale, ulp, e(direct, ex_s), e(wino, ex_s), e(wino, ex_s) / max(e(direct, ex_s), 1e-30), e(wino_noround, ex_s),
                 e(direct, exact), e(wino, exact), e(rnd(direct, dt), exact), e(rnd(wino, dt), exact)))


report("stage3 256->256 @32", 256, 256, 32)
report("stage4 512->512 @16", 512, 512, 16)
report("stage2 128->128 @64", 128, 128, 64, n=1)
