"""Helper of tests/test_ops_gpu.py::test_lean_kernels_equal_the_generic_ones: runs the small-channel / stem launches the training step
uses on seeded inputs with whatever kernel the environment selects (STP_SC_LEAN, STP_STEM_LEAN, STP_STEM_WG_LEAN are read once per
process) and prints a JSON line: sha256 of every output tensor + the reduced statistic sums."""
import hashlib, json, sys
import numpy as np
import torch
from segmentation_training_pipeline_amd import _lib, ops

DEV = "cuda"
keep = []
def t16(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).to(DEV).contiguous(); keep.append(t); return t
def f32(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV); keep.append(t); return t
def sha(t):
    torch.cuda.synchronize(); return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]
def weights(wt):                      # HWIO numpy -> forward copy [rows16][kh][kw][ci] bf16
    kh, kw, ci, co = wt.shape
    master = f32(wt.transpose(3, 0, 1, 2)); rows = (co + 15) // 16 * 16
    fwd = torch.empty(rows * kh * kw * ci, dtype=torch.bfloat16, device=DEV); bwd = torch.empty(((ci + 15) // 16 * 16) * kh * kw * ((co + 7) // 8 * 8), dtype=torch.bfloat16, device=DEV)
    ops.weight_prepare(master, fwd, bwd, co, kh, kw, ci, kw, ci, (co + 7) // 8 * 8, ops.dt(fwd)); keep.extend([fwd, bwd]); return fwd
def sums(P, st, co):
    torch.cuda.synchronize(); tiles = ops.conv2d_stats_floats(P) // (2 * co)
    return st[:2 * co * tiles].reshape(2, co, tiles).double().sum(dim=2).cpu().numpy().ravel().tolist()

rng = np.random.RandomState(5)
out = {}
n, h, w = 1, 43, 139                # 6 x 5 tiles of 8 x 32: interior and ragged border tiles
for name, ci, co, up, mode in [("fwd16", 16, 16, 0, "pbn+stats"), ("fwd32up", 32, 16, 1, "pbn+stats"), ("fwd32x32", 32, 32, 0, "stats"), ("head", 16, 1, 0, "bias"),
                               ("dgrad16", 16, 16, 0, "bnb"), ("dgrad8", 8, 16, 0, "bnb"), ("dgrad32x32", 32, 32, 0, "bnb"), ("dgrad_sum2", 16, 32, 0, "sum2")]:
    hs, ws = ((h + 1) // 2, (w + 1) // 2) if up else (h, w)
    H, W = (2 * hs, 2 * ws) if up else (h, w)
    if mode == "sum2": H, W = H - (H & 1), W - (W & 1); hs, ws = H, W
    x = t16(rng.randn(n, hs, ws, ci)); fwd = weights(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci))
    oh, ow = (H // 2, W // 2) if mode == "sum2" else (H, W)
    y = torch.full((n, oh, ow, co), float("nan"), dtype=torch.bfloat16, device=DEV)
    P = ops.conv_params(x, fwd, y, N=n, Hs0=hs, Ws0=ws, Hv=H, Wv=W, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=H, Wo=W, Cout=co, dtype=ops.BF16,
                        mode=ops.SRC_NEAREST2X if up else ops.SRC_DIRECT)
    m, r, g, b = f32(rng.randn(32) * 0.2), f32(rng.rand(32) + 0.5), f32(rng.rand(32) + 0.5), f32(rng.randn(32) * 0.3)
    if "pbn" in mode: P.src_bn_mean, P.src_bn_rstd, P.src_bn_gamma, P.src_bn_beta, P.src_bn_relu = ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    if mode == "bias": P.bias = ops.ptr(f32(np.array([0.3])))
    if mode == "sum2": P.dst_sum2x2 = 1
    st = None
    if mode in ("pbn+stats", "stats", "bnb", "sum2"):
        if mode in ("bnb", "sum2"):
            xb = t16(rng.randn(n, oh, ow, co) + 0.2)
            P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xb), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
        st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV); P.stats_partial = ops.ptr(st)
    assert _lib.load().stp_conv2d_sc_eligible(P), name
    ops.conv2d(P)
    out[name] = {"y": sha(y), "sums": sums(P, st, co) if st is not None else []}
# stem forward with statistics + weight gradient
hh, ww = 58, 268
x4 = t16(np.concatenate([rng.randn(1, hh, ww, 3), np.ones((1, hh, ww, 1))], axis=-1)); ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
master = f32((rng.randn(7, 7, 3, 64) / 12).transpose(3, 0, 1, 2)); fwd = torch.empty(64 * 7 * 8 * 4, dtype=torch.bfloat16, device=DEV); bwd = torch.empty(16 * 7 * 7 * 64, dtype=torch.bfloat16, device=DEV)
ops.weight_prepare(master, fwd, bwd, 64, 7, 7, 3, 8, 4, 64, ops.dt(fwd))
y = torch.empty((1, ho, wo, 64), dtype=torch.bfloat16, device=DEV)
P = ops.conv_params(x4, fwd, y, N=1, Hs0=hh, Ws0=ww, Hv=hh, Wv=ww, C0=4, KH=7, KW=8, stride=2, pad=3, Ho=ho, Wo=wo, Cout=64, dtype=ops.BF16)
st = torch.zeros(max(4, ops.conv2d_stats_floats(P)), device=DEV); P.stats_partial = ops.ptr(st)
ops.conv2d(P)
out["stem"] = {"y": sha(y), "sums": sums(P, st, 64)}
dy = t16(rng.randn(1, ho, wo, 64)); dwp = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=DEV)
Wp = ops.wgrad_params(x4, dy, dwp, N=1, Hs0=hh, Ws0=ww, Hv=hh, Wv=ww, C0=4, KH=7, KW=8, stride=2, pad=3, Ho=ho, Wo=wo, Cout=64, dtype=ops.BF16)
ws = torch.empty(ops.wgrad_workspace_bytes(Wp) // 4 + 4, dtype=torch.float32, device=DEV)
ops.conv2d_wgrad(Wp, ws); torch.cuda.synchronize()
out["stem_wgrad"] = {"y": "", "sums": dwp[:, :, :7, :].double().cpu().numpy().ravel().tolist()}
print("LEANJSON " + json.dumps(out))
