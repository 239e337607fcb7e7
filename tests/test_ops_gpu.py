"""GPU parity tests: every C-ABI entry point against the CPU oracle (oracle/np_ops.py etc.).

Parity is "vs the in-repo CPU oracle" - the reference's Keras path is not executable (see
oracle/__init__.py).  Tolerances: fp32 mode 1e-4 relative to the output scale (exact-fp32 MFMA,
only the summation order differs); bf16 mode compares against the oracle evaluated on the same
bf16-rounded inputs, allowing one bf16 output rounding (2^-8 relative) plus fp32 accumulation.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment as oaug  # noqa: E402
from oracle import losses as olosses  # noqa: E402
from oracle import np_ops  # noqa: E402
from oracle import optim as ooptim  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from segmentation_training_pipeline_amd import ops as o
    return o


DEV = "cuda"
TD = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}
H16 = ["bf16", "fp16"]      # the two 16-bit storage builds of the kernel set (libstp_hip.so / libstp_hip_f16.so)


@pytest.fixture(autouse=True)
def _storage_build(request):
    """Tests parametrized with dtype "fp16" call into libstp_hip_f16.so (IEEE-half storage, v_mfma_*_f16)."""
    from segmentation_training_pipeline_amd import _lib
    dt = request.node.callspec.params.get("dtype") if hasattr(request.node, "callspec") else None
    with _lib.storage("fp16" if dt == "fp16" else "bf16"):
        yield


def q(a, dtype):
    """Round a numpy array through the storage dtype (so oracle and kernel see equal inputs)."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TD[dtype]).to(torch.float32).numpy()


_KEEP = []


def dev(a, dtype):
    """Device copy in the storage dtype.  Params structs hold raw pointers only, so every
    temporary is parked in _KEEP until the test ends (see the autouse fixture)."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(TD[dtype]).to(DEV).contiguous()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _release_device_temporaries():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def keep(t):
    _KEEP.append(t)
    return t


def host(t):
    torch.cuda.synchronize()
    return t.detach().to(torch.float32).cpu().numpy()


def tol(ref, dtype, k=1.0):
    s = float(np.abs(ref).max()) + 1e-6
    # one output rounding: 2^-8 relative for bf16, 2^-11 for IEEE half (the bound is stated on the tensor's scale)
    return {"fp32": 2e-4, "bf16": 1.2e-2, "fp16": 2e-3}[dtype] * s * k


def prep_weights(ops, w_hwio, dtype, KWp=None, Cinp=None, CoutB=None):
    """HWIO numpy -> (master OHWI fp32 dev, fwd copy, bwd copy)."""
    kh, kw, ci, co = w_hwio.shape
    KWp = KWp or kw
    Cinp = Cinp or ci
    vec = 8 if dtype != "fp32" else 4
    CoutB = CoutB or ((co + vec - 1) // vec * vec)
    master = torch.from_numpy(np.ascontiguousarray(w_hwio.transpose(3, 0, 1, 2), dtype=np.float32)).to(DEV)
    rows_f = (co + 15) // 16 * 16
    rows_b = (ci + 15) // 16 * 16
    fwd = torch.empty(rows_f * kh * KWp * Cinp, dtype=TD[dtype], device=DEV)
    bwd = torch.empty(rows_b * kh * kw * CoutB, dtype=TD[dtype], device=DEV)
    ops.weight_prepare(master, fwd, bwd, co, kh, kw, ci, KWp, Cinp, CoutB, ops.dt(fwd))
    return master, fwd, bwd, CoutB


CONV_CASES = [
    # n, h, w, ci, co, k, stride, pad, tile
    (2, 16, 16, 32, 128, 3, 1, 1, 1),
    (2, 16, 16, 64, 64, 3, 1, 1, 2),
    (1, 24, 20, 16, 32, 3, 1, 1, 3),
    (1, 20, 24, 32, 16, 3, 1, 1, 4),
    (2, 9, 11, 64, 64, 3, 1, 1, 5),
    (2, 12, 12, 128, 256, 3, 2, 1, 6),
    (2, 16, 16, 64, 128, 1, 2, 0, 0),
    (1, 8, 8, 256, 512, 3, 1, 1, 0),
    (3, 7, 5, 24, 40, 3, 1, 1, 0),     # ragged everything (Cout multiple of 4 only)
    # uniform-tap buffer-DMA kernel (Cin multiple of the 128-byte K-step): tile = 32*STAGES + base tile
    (2, 16, 16, 64, 128, 3, 1, 1, 65), (2, 16, 16, 64, 128, 3, 1, 1, 97),
    (2, 16, 16, 64, 64, 3, 1, 1, 66), (2, 16, 16, 64, 64, 3, 1, 1, 98),
    (2, 9, 11, 128, 64, 3, 1, 1, 69), (2, 9, 11, 128, 64, 3, 1, 1, 101), (2, 9, 11, 128, 64, 3, 1, 1, 133),
    (2, 12, 12, 128, 256, 3, 2, 1, 70), (2, 12, 12, 128, 256, 3, 2, 1, 102), (1, 13, 9, 192, 136, 3, 1, 1, 134),
    (2, 16, 16, 64, 128, 1, 2, 0, 70),
    (1, 24, 20, 64, 32, 3, 1, 1, 67), (1, 20, 24, 128, 16, 3, 1, 1, 68),
    # per-lane-tap DMA variants for 8/16/32-channel inputs (several taps per K-tile, ragged K tail)
    (1, 24, 20, 16, 64, 3, 1, 1, 258), (1, 24, 20, 32, 32, 3, 1, 1, 259), (1, 20, 24, 16, 16, 3, 1, 1, 260),
    (2, 13, 9, 32, 16, 3, 1, 1, 0), (2, 13, 9, 8, 16, 3, 1, 1, 0), (2, 12, 12, 16, 32, 3, 2, 1, 0),
    # small-channel halo-tile kernel (tile 512) incl. ragged tiles, and the GEMM path forced for the same shapes
    (2, 19, 45, 16, 16, 3, 1, 1, 512), (1, 8, 32, 32, 32, 3, 1, 1, 512), (2, 21, 33, 8, 16, 3, 1, 1, 512),
    (2, 19, 45, 16, 32, 3, 1, 1, 512), (2, 19, 45, 16, 16, 3, 1, 1, 260), (1, 40, 70, 32, 24, 3, 1, 1, 0),
    (2, 16, 16, 32, 128, 3, 1, 1, 257), (2, 16, 16, 32, 128, 3, 1, 1, 262), (2, 9, 11, 16, 64, 3, 1, 1, 261),
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward(ops, dtype, case):
    n, h, w, ci, co, k, s, p, tile = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(k, k, ci, co) / np.sqrt(k * k * ci), dtype)
    ref = np_ops.conv2d(x, wt, s, p)
    ho, wo = ref.shape[1:3]
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    y = torch.full((n, ho, wo, co), float("nan"), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(xd, fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=s, pad=p,
                        Ho=ho, Wo=wo, Cout=co, dtype=ops.dt(y), tile=tile)
    try:
        ops.conv2d(P)
    except Exception:
        if tile < 64:
            raise
        P.tile = 0        # a forced DMA tile that this dtype's K-step does not admit: the library must say so...
        ops.conv2d(P)     # ...and the automatic choice must still work
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_conv2d_epilogue_bias_residual_relu_dualdest_accumulate(ops, dtype):
    rng = np.random.RandomState(3)
    n, h, w, ci, co = 2, 10, 10, 32, 48
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / 17.0, dtype)
    bias = rng.randn(co).astype(np.float32)
    res = q(rng.randn(n, h, w, co), dtype)
    ref = np.maximum(np_ops.conv2d(x, wt, 1, 1, bias) + res, 0)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    y = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(x, dtype), fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1,
                        Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), bias=keep(torch.from_numpy(bias).to(DEV)),
                        residual=dev(res, dtype), relu=1)
    ops.conv2d(P)
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    # dual destination with accumulate on the second one
    ref2 = np_ops.conv2d(x, wt, 1, 1)
    d0 = torch.empty((n, h, w, 16), dtype=TD[dtype], device=DEV)
    init1 = q(rng.randn(n, h, w, 32), dtype)
    d1 = dev(init1, dtype)
    P = ops.conv_params(dev(x, dtype), fwd, d0, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1,
                        Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), dst1=d1, Cd0=16, accumulate1=1)
    ops.conv2d(P)
    np.testing.assert_allclose(host(d0), ref2[..., :16], atol=tol(ref2, dtype))
    np.testing.assert_allclose(host(d1), ref2[..., 16:] + init1, atol=tol(ref2, dtype, 2))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 12, 12), (1, 29, 133)])      # the second: interior tiles of the lean small-channel kernel
def test_conv2d_head_single_class_with_bias(ops, dtype, shape):
    rng = np.random.RandomState(4)
    (n, h, w), ci, co = shape, 16, 1
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / 12.0, dtype)
    bias = np.array([0.3], np.float32)
    ref = np_ops.conv2d(x, wt, 1, 1, bias)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    y = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(x, dtype), fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1,
                        Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), bias=keep(torch.from_numpy(bias).to(DEV)))
    ops.conv2d(P)
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_conv2d_upsample_concat_gather(ops, dtype):
    """decoder conv1: conv3x3(concat(UpSampling2D(2)(x), skip)) without materialising either."""
    rng = np.random.RandomState(5)
    n, h, w, c0, c1, co = 2, 6, 7, 32, 16, 64
    x = q(rng.randn(n, h, w, c0), dtype)
    skip = q(rng.randn(n, 2 * h, 2 * w, c1), dtype)
    wt = q(rng.randn(3, 3, c0 + c1, co) / 20.0, dtype)
    v = np.concatenate([np_ops.upsample2x(x), skip], axis=-1)
    ref = np_ops.conv2d(v, wt, 1, 1)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    y = torch.empty((n, 2 * h, 2 * w, co), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(x, dtype), fwd, y, N=n, Hs0=h, Ws0=w, Hv=2 * h, Wv=2 * w, C0=c0, C1=c1, src1=dev(skip, dtype),
                        mode=ops.SRC_NEAREST2X, KH=3, KW=3, stride=1, pad=1, Ho=2 * h, Wo=2 * w, Cout=co, dtype=ops.dt(y))
    ops.conv2d(P)
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64, 32), (1, 16, 32, 128, 64, 64), (2, 8, 16, 64, 128, 128)])
def test_conv2d_upsample_concat_with_class_collapsed_weights(ops, dtype, shape):
    """stp_conv_params.weight_up: per output parity class the nine taps over the nearest-2x upsampled source read 2 x 2 low-resolution
    pixels, so the uniform-tap kernel multiplies them by the class-summed weights (stp_weight_prepare_upcollapse) - 4 x C0 + 9 x C1
    K columns instead of 9 x (C0 + C1).  Must equal the plain gather (up to the one extra rounding of the summed bf16 weights) and
    the numpy reference, including the image borders; the per-tile statistics ride along."""
    from segmentation_training_pipeline_amd import _lib
    rng = np.random.RandomState(15)
    n, h, w, c0, c1, co = shape                                  # low-resolution size of the upsampled source
    x = q(rng.randn(n, h, w, c0), dtype)
    skip = q(rng.randn(n, 2 * h, 2 * w, c1), dtype)
    wt = q(rng.randn(3, 3, c0 + c1, co) / 30.0, dtype)
    ref = np_ops.conv2d(np.concatenate([np_ops.upsample2x(x), skip], axis=-1), wt, 1, 1)
    master, fwd, _, _ = prep_weights(ops, wt, dtype)
    rows = (co + 15) // 16 * 16
    wup = torch.full((rows * 16 * c0,), float("nan"), dtype=TD[dtype], device=DEV)
    _lib.call("stp_weight_prepare_upcollapse", ops.ptr(master), ops.ptr(wup), co, c0, c1, ops.dt(wup), ops.stream())
    # the collapsed matrix against numpy: class (py, px), tap (ty, tx)
    sets = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    want = np.zeros((rows, 2, 2, 2, 2, c0), np.float32)
    for py in (0, 1):
        for px in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    want[:co, py, px, ty, tx] = sum(wt[kh, kw, :c0, :].T for kh in sets[(py, ty)] for kw in sets[(px, tx)])
    np.testing.assert_allclose(host(wup).reshape(want.shape), q(want, dtype), atol=0 if dtype == "fp32" else 2e-3)

    def run(weight_up, stats):
        y = torch.empty((n, 2 * h, 2 * w, co), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(dev(x, dtype), fwd, y, N=n, Hs0=h, Ws0=w, Hv=2 * h, Wv=2 * w, C0=c0, C1=c1, src1=dev(skip, dtype),
                            mode=ops.SRC_NEAREST2X, KH=3, KW=3, stride=1, pad=1, Ho=2 * h, Wo=2 * w, Cout=co, dtype=ops.dt(y))
        P.weight_up = ops.ptr(weight_up)
        if _lib.load().stp_conv2d_scn_eligible(ops.C.byref(P)):
            P.tile = 64 + 3              # (64 + 64 -> 32 channels: the narrow-output kernel would take it and ignore weight_up; this test is about the generic one)
        st = None
        if stats:
            st = torch.zeros(max(ops.conv2d_stats_floats(P), 4), device=DEV)
            P.stats_partial = ops.ptr(st)
        ops.conv2d(P)
        return host(y), (None if st is None else host(st).reshape(2, co, -1).sum(-1))
    plain, pst = run(None, True)
    coll, cst = run(wup, True)
    np.testing.assert_allclose(plain, ref, atol=tol(ref, dtype))
    np.testing.assert_allclose(coll, ref, atol=tol(ref, dtype))
    np.testing.assert_allclose(coll, plain, atol=(1e-5 if dtype == "fp32" else 0.05) * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(cst, pst, rtol=2e-2, atol=0.5 if dtype != "fp32" else 1e-2)
    assert np.abs(coll - plain).max() > 0 or dtype == "fp32"        # (the collapsed path really ran: bf16 rounds the sums once more)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("geom", [(3, 1, 1, 10, 12), (3, 2, 1, 12, 10), (1, 2, 0, 8, 8), (3, 2, 1, 9, 11)])
def test_conv2d_data_gradient(ops, dtype, geom):
    """dgrad = stp_conv2d over dY with the flipped/transposed weight copy (zero-insertion for stride 2)."""
    k, s, p, h, w = geom
    rng = np.random.RandomState(6)
    n, ci, co = 2, 32, 64
    wt = q(rng.randn(k, k, ci, co) / np.sqrt(k * k * co), dtype)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = q(rng.randn(n, ho, wo, co), dtype)
    ref = np_ops.conv2d_dgrad(dy, wt, (h, w), s, p)
    _, _, bwd, coB = prep_weights(ops, wt, dtype)
    dx = torch.empty((n, h, w, ci), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(dy, dtype), bwd, dx, N=n, Hs0=ho, Ws0=wo, Hv=(2 * ho - 1 if s == 2 else ho),
                        Wv=(2 * wo - 1 if s == 2 else wo), C0=coB, mode=(ops.SRC_ZEROINS2X if s == 2 else ops.SRC_DIRECT),
                        KH=k, KW=k, stride=1, pad=k - 1 - p, Ho=h, Wo=w, Cout=ci, dtype=ops.dt(dx))
    ops.conv2d(P)
    np.testing.assert_allclose(host(dx), ref, atol=tol(ref, dtype))


@pytest.mark.parametrize("geom", [(3, 1, 32, 32, 128, 64), (1, 0, 16, 64, 128, 128), (3, 1, 16, 32, 64, 256), (3, 1, 64, 16, 256, 64)])
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bn_backward"])
@pytest.mark.parametrize("dtype", H16)
def test_stride2_data_gradient_in_parity_class_order(ops, geom, mode, dtype):
    """Data gradient of a stride-2 convolution through the uniform-tap kernel (bf16, dY channels % 64 == 0, even maps whose quarter
    is a multiple of the pixel tile): pixels are ordered by parity class so that the K loop visits only the taps that meet real
    samples of the zero-inserted dY (ConvArgs::zperm).  Same results as the numpy data gradient; the epilogue (accumulate, fused
    BatchNormalization-backward sums) must address the REAL pixel of every logical one."""
    k, p, h, w, co, ci = geom
    rng = np.random.RandomState(16)
    n = 2
    wt = q(rng.randn(k, k, ci, co) / np.sqrt(k * k * co), dtype)
    ho, wo = (h + 2 * p - k) // 2 + 1, (w + 2 * p - k) // 2 + 1
    dy = q(rng.randn(n, ho, wo, co), dtype)
    ref = np_ops.conv2d_dgrad(dy, wt, (h, w), 2, p)
    _, _, bwd, coB = prep_weights(ops, wt, dtype)
    prev = q(rng.randn(n, h, w, ci), dtype)
    dx = dev(prev, dtype) if mode != "plain" else torch.full((n, h, w, ci), float("nan"), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(dy, dtype), bwd, dx, N=n, Hs0=ho, Ws0=wo, Hv=2 * ho - 1, Wv=2 * wo - 1, C0=coB, mode=ops.SRC_ZEROINS2X,
                        KH=k, KW=k, stride=1, pad=k - 1 - p, Ho=h, Wo=w, Cout=ci, dtype=ops.dt(dx), accumulate0=int(mode != "plain"))
    assert 64 <= ops._lib.load().stp_conv2d_tile_for(ops.C.byref(P)) < 256          # the uniform-tap buffer-DMA kernel
    want = ref + (prev if mode != "plain" else 0.0)
    if mode != "bn_backward":
        ops.conv2d(P)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype))
        return
    # this data gradient completes the gradient of a BatchNormalization(+ReLU) output: masked store + the two backward sums
    rows = n * h * w
    x = q(rng.randn(n, h, w, ci) * 1.5 + 0.3, dtype)
    gamma, beta = (rng.rand(ci) + 0.5).astype(np.float32), (rng.randn(ci) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(ci, device=DEV), torch.empty(ci, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(ci) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, ci, 1e-3, 0.99, m, r, None, None, ws)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    tiles = ops.conv2d_stats_floats(P) // (2 * ci)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    gm = want * (pre > 0)                                                       # dY under the ReLU mask
    np.testing.assert_allclose(host(dx)[safe], gm[safe], atol=tol(want, dtype))
    part = host(st).reshape(2, ci, tiles).sum(axis=2)
    gs = host(dx).astype(np.float64)                                            # the sums are those of the STORED gradient
    xhat = (host(xd).astype(np.float64) - host(m)) * host(r)
    np.testing.assert_allclose(part[0], gs.reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs).sum(axis=(0, 1, 2)).max() + 1e-3)
    np.testing.assert_allclose(part[1], (gs * xhat).reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs * xhat).sum(axis=(0, 1, 2)).max() + 1e-3)


@pytest.mark.parametrize("geom", [(2, 32, 32, 128, 64, 0), (1, 16, 64, 128, 128, 1), (2, 32, 64, 64, 32, 1), (1, 64, 32, 256, 64, 0), (3, 16, 32, 64, 256, 1)])
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bn_backward"])
@pytest.mark.parametrize("shortcut", [False, True])
@pytest.mark.parametrize("dtype", H16)
def test_stride2_data_gradient_space_to_depth(ops, geom, mode, shortcut, dtype):
    """stp_conv_params.s2d_dgrad (round 5): the data gradient of a 3x3 / stride-2 / pad-1 convolution as ONE dense 2 x 2-tap convolution
    of dY into the four parity classes on the halo kernel (conv_halo_s2d_kernel), stored depth-to-space, with the sibling 1x1 / stride-2
    shortcut's dY as a second source.  Against the numpy data gradients; accumulate; the fused BatchNormalization-backward epilogue
    (masked store at the REAL pixel + the sums as four column blocks per channel); the weights are the layers' ordinary data-gradient copies."""
    from segmentation_training_pipeline_amd import _lib
    n, ho, wo, co, ci, var = geom               # co = channels of dY, ci = channels of the gradient
    h, w = 2 * ho, 2 * wo
    rng = np.random.RandomState(hash(geom) % 2**31)
    w3 = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * co), dtype)
    wsc = q(rng.randn(ci, co) / np.sqrt(co), dtype) if shortcut else None
    dy = q(rng.randn(n, ho, wo, co), dtype)
    dysc = q(rng.randn(n, ho, wo, co), dtype) if shortcut else None
    ref = np_ops.conv2d_dgrad(dy, w3, (h, w), 2, 1)
    if shortcut:
        ref = ref + np_ops.conv2d_dgrad(dysc, wsc.reshape(1, 1, ci, co), (h, w), 2, 0)
    _, _, wd, _ = prep_weights(ops, w3, dtype)                                   # the ordinary data-gradient copies [ci][3][3][co] ...
    wsd = prep_weights(ops, wsc.reshape(1, 1, ci, co), dtype)[2] if shortcut else None      # ... and [ci][co] of the shortcut
    prev = q(rng.randn(n, h, w, ci), dtype)
    dx = dev(prev, dtype) if mode != "plain" else torch.full((n, h, w, ci), float("nan"), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(dy, dtype), wd, dx, N=n, Hs0=ho, Ws0=wo, Hv=ho, Wv=wo, C0=co, C1=(co if shortcut else 0),
                        src1=(dev(dysc, dtype) if shortcut else None), mode=ops.SRC_DIRECT, KH=2, KW=2, stride=1, pad=0, Ho=ho, Wo=wo,
                        Cout=4 * ci, dtype=ops.dt(dx), accumulate0=int(mode != "plain"), tile=1024 + var)
    P.s2d_dgrad = 1
    if shortcut:
        P.fold_weight = ops.ptr(wsd)
    assert _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 1024 + var
    want = ref + (prev if mode != "plain" else 0.0)
    if mode != "bn_backward":
        ops.conv2d(P)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype))
        return
    rows = n * h * w
    x = q(rng.randn(n, h, w, ci) * 1.5 + 0.3, dtype)
    gamma, beta = (rng.rand(ci) + 0.5).astype(np.float32), (rng.randn(ci) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(ci, device=DEV), torch.empty(ci, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(ci) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, ci, 1e-3, 0.99, m, r, None, None, ws)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    cols = ops.conv2d_stats_floats(P) // (2 * ci)
    assert cols == P.stats_tiles == 4 * n * (ho // (16 if var == 0 else 8)) * (wo // 16)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    gm = want * (pre > 0)
    np.testing.assert_allclose(host(dx)[safe], gm[safe], atol=tol(want, dtype))
    part = host(st).reshape(2, ci, cols).sum(axis=2)
    gs = host(dx).astype(np.float64)
    xhat = (host(xd).astype(np.float64) - host(m)) * host(r)
    np.testing.assert_allclose(part[0], gs.reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs).sum(axis=(0, 1, 2)).max() + 1e-3)
    np.testing.assert_allclose(part[1], (gs * xhat).reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs * xhat).sum(axis=(0, 1, 2)).max() + 1e-3)


@pytest.mark.parametrize("geom", [(2, 32, 32, 64, 64, 3), (1, 16, 32, 128, 128, 0), (2, 16, 16, 64, 128, 1), (1, 32, 16, 128, 64, 2)])
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bn_backward"])
@pytest.mark.parametrize("dtype", H16)
def test_stride1_data_gradient_with_folded_shortcut(ops, geom, mode, dtype):
    """fold_src / fold_weight / fold_C on a 3x3 / stride-1 data gradient (conv_halo_fold1_kernel, round 5): the dY of the sibling 1x1 /
    stride-1 shortcut is a second source whose centre tap carries the shortcut's data-gradient weights - one launch produces
    dgrad3x3(dY) + dgrad1x1(dY_sc), with accumulate and the fused BatchNormalization-backward epilogue; against numpy."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, co, ci, var = geom               # co = channels of both dYs, ci = channels of the gradient
    rng = np.random.RandomState(hash(geom) % 2**31)
    w3 = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * co), dtype)
    wsc = q(rng.randn(1, 1, ci, co) / np.sqrt(co), dtype)
    dy, dysc = q(rng.randn(n, h, w, co), dtype), q(rng.randn(n, h, w, co), dtype)
    ref = np_ops.conv2d_dgrad(dy, w3, (h, w), 1, 1) + np_ops.conv2d_dgrad(dysc, wsc, (h, w), 1, 0)
    wd, wsd = prep_weights(ops, w3, dtype)[2], prep_weights(ops, wsc, dtype)[2]
    prev = q(rng.randn(n, h, w, ci), dtype)
    dx = dev(prev, dtype) if mode != "plain" else torch.full((n, h, w, ci), float("nan"), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(dy, dtype), wd, dx, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=co, mode=ops.SRC_DIRECT, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                        Cout=ci, dtype=ops.dt(dx), accumulate0=int(mode != "plain"), tile=1024 + var)
    P.fold_src, P.fold_weight, P.fold_C = ops.ptr(dev(dysc, dtype)), ops.ptr(wsd), co
    assert _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 1024 + var
    want = ref + (prev if mode != "plain" else 0.0)
    if mode != "bn_backward":
        ops.conv2d(P)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype))
        return
    rows = n * h * w
    x = q(rng.randn(n, h, w, ci) * 1.5 + 0.3, dtype)
    gamma, beta = (rng.rand(ci) + 0.5).astype(np.float32), (rng.randn(ci) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(ci, device=DEV), torch.empty(ci, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(ci) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, ci, 1e-3, 0.99, m, r, None, None, ws)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    cols = ops.conv2d_stats_floats(P) // (2 * ci)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    gm = want * (pre > 0)
    np.testing.assert_allclose(host(dx)[safe], gm[safe], atol=tol(want, dtype))
    part = host(st).reshape(2, ci, cols).sum(axis=2)
    gs = host(dx).astype(np.float64)
    xhat = (host(xd).astype(np.float64) - host(m)) * host(r)
    np.testing.assert_allclose(part[0], gs.reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs).sum(axis=(0, 1, 2)).max() + 1e-3)
    np.testing.assert_allclose(part[1], (gs * xhat).reshape(-1, ci).sum(0), atol=2e-4 * np.abs(gs * xhat).sum(axis=(0, 1, 2)).max() + 1e-3)


@pytest.mark.parametrize("size", [(32, 36), (64, 64), (37, 70), (70, 330), (59, 262)])      # the last two: interior tiles of the persistent form
@pytest.mark.parametrize("dtype", H16)
def test_stem_halo_kernel_with_fused_statistics(ops, size, dtype):
    """conv_stem_kernel (bf16, tile id 768): the ResNet conv0 through the halo-tile kernel at even, tile-aligned and ragged / odd
    sizes, against the naive oracle and the generic implicit GEMM (tile 2) on the same buffers; its fused BatchNormalization
    sums through stp_bn_finalize against stp_bn_stats of the stored output."""
    from segmentation_training_pipeline_amd import _lib
    h, w = size
    n, co = 2, 64
    rng = np.random.RandomState(9)
    x3 = q(rng.randn(n, h, w, 3), dtype)
    wt = q(rng.randn(7, 7, 3, co) / 12.0, dtype)
    ref = np_ops.conv2d(x3, wt, 2, 3)
    ho, wo = ref.shape[1:3]
    x4 = np.concatenate([x3, np.ones((n, h, w, 1), np.float32)], axis=-1)
    _, fwd, _, _ = prep_weights(ops, wt, dtype, KWp=8, Cinp=4)
    xd = dev(x4, dtype)
    mk = lambda dst, tile: ops.conv_params(xd, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=4, KH=7, KW=8, stride=2, pad=3, Ho=ho, Wo=wo,
                                           Cout=co, dtype=ops.dt(dst), tile=tile)
    y, y2 = torch.empty((n, ho, wo, co), dtype=TD[dtype], device=DEV), torch.empty((n, ho, wo, co), dtype=TD[dtype], device=DEV)
    P = mk(y, 0)
    assert _lib.load().stp_conv2d_stem_eligible(P) and _lib.load().stp_conv2d_tile_for(P) == 768
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    ops.conv2d(mk(y2, 2))
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    np.testing.assert_allclose(host(y), host(y2), atol=tol(ref, dtype))          # two summation orders, one output rounding each
    tiles = ops.conv2d_stats_floats(P) // (2 * co)
    assert tiles == P.stats_tiles and tiles <= n * -(-ho // 8) * -(-wo // 32)      # one column per tile, or per workgroup of the persistent form
    rows = n * ho * wo
    m1, r1, m0, r0 = (torch.empty(co, device=DEV) for _ in range(4))
    _lib.call("stp_bn_finalize", ops.ptr(st), tiles, rows, co, 2e-5, 0.99, ops.ptr(m1), ops.ptr(r1), None, None, ops.stream())
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(y, rows, co, 2e-5, 0.99, m0, r0, None, None, ws)
    np.testing.assert_allclose(host(m1), host(m0), atol=2e-6 * max(1.0, np.abs(host(m0)).max()))
    np.testing.assert_allclose(host(r1), host(r0), rtol=2e-5)


HALO_CASES = [
    # n, h, w, ci, co, variant (conv_halo.hip: 0 = 16x16 px x 128 ch, 1 = 8x16 x 128, 2 = 16x16 x 64, 3 = 8x16 x 64, 4 = 32x16 x 64)
    (2, 16, 32, 128, 128, 0), (1, 32, 16, 64, 192, 0), (1, 16, 16, 256, 128, 0),
    (2, 8, 16, 128, 256, 1), (1, 24, 32, 64, 80, 1),
    (1, 32, 16, 64, 64, 2), (2, 16, 16, 192, 64, 2),
    (2, 8, 32, 128, 64, 3), (1, 24, 16, 512, 64, 3),
    (1, 32, 32, 64, 64, 4), (2, 64, 16, 64, 80, 4),           # 32x16 pixels x 64 channels: 64-channel inputs only
    # 5 = the persistent 64 -> 64 form (conv_halo_p64_kernel: two 4-wave workgroups per CU walk 8 x 16-pixel tiles): 4 tiles; 36; 640 = a second, ragged round
    (1, 32, 16, 64, 64, 5), (3, 48, 32, 64, 64, 5), (5, 128, 128, 64, 64, 5),
]


@pytest.mark.parametrize("case", HALO_CASES)
@pytest.mark.parametrize("dtype", H16)
def test_conv_halo_kernel_forward_statistics_residual(ops, case, dtype):
    """conv_halo_kernel (bf16, tile id 1024 + variant): 3x3 / stride 1 through halo-resident activation slabs against the naive
    oracle and the per-tap DMA kernel on the same buffers; fused BatchNormalization sums against stp_bn_stats of the stored output;
    residual + ReLU epilogue; image borders (zero padding through out-of-range buffer offsets) on every side of every tile."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, var = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    ref = np_ops.conv2d(x, wt, 1, 1)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    mk = lambda dst, tile, **kw: ops.conv_params(xd, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                                 Cout=co, dtype=ops.dt(dst), tile=tile, **kw)
    y = torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
    y2 = torch.empty_like(y)
    P = mk(y, 1024 + var)
    assert _lib.load().stp_conv2d_tile_for(P) == 1024 + var
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    ops.conv2d(mk(y2, 69))                                                   # per-tap DMA kernel, 64x64 tile
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    np.testing.assert_allclose(host(y), host(y2), atol=tol(ref, dtype))
    th = {0: 16, 1: 8, 2: 16, 3: 8, 4: 32, 5: 8}[var]
    tiles = ops.conv2d_stats_floats(P) // (2 * co)
    assert tiles == P.stats_tiles == n * (h // th) * (w // 16)
    rows = n * h * w
    m1, r1, m0, r0 = (torch.empty(co, device=DEV) for _ in range(4))
    _lib.call("stp_bn_finalize", ops.ptr(st), tiles, rows, co, 1e-3, 0.99, ops.ptr(m1), ops.ptr(r1), None, None, ops.stream())
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(y, rows, co, 1e-3, 0.99, m0, r0, None, None, ws)
    np.testing.assert_allclose(host(m1), host(m0), atol=2e-6 * max(1.0, np.abs(host(m0)).max()))
    np.testing.assert_allclose(host(r1), host(r0), rtol=2e-5)
    # residual + ReLU (prefetched epilogue operand)
    res = q(rng.randn(n, h, w, co), dtype)
    y3 = torch.empty_like(y)
    ops.conv2d(mk(y3, 1024 + var, residual=dev(res, dtype), relu=1))
    ref3 = np.maximum(ref + res, 0)
    np.testing.assert_allclose(host(y3), ref3, atol=tol(ref3, dtype))


HALO2_CASES = [
    # n, h, w, c0, c1, co, variant, upsampled first source
    (2, 16, 32, 128, 64, 128, 0, True), (1, 32, 16, 64, 64, 192, 0, True), (2, 8, 16, 256, 128, 256, 1, True),
    (1, 32, 16, 128, 64, 64, 2, True), (2, 8, 32, 64, 128, 64, 3, True),
    (1, 16, 16, 64, 128, 128, 0, False), (2, 24, 16, 128, 0, 64, 3, True),       # plain concatenation; one upsampled source (Linknet)
]


@pytest.mark.parametrize("case", HALO2_CASES)
@pytest.mark.parametrize("dtype", H16)
def test_conv_halo_kernel_two_sources_with_upsampled_first(ops, case, dtype):
    """conv_halo2_kernel (round 5): Conv2D(3x3)(Concatenate([UpSampling2D(2)(x), skip])) on the halo-resident kernel - a 64-channel slab
    lies in one source, the slabs of the first are staged from the LOW-resolution pixels (y >> 1, x >> 1).  Against numpy on the
    materialised concatenation and against the per-tap kernel on the same buffers; fused statistics against the stored output;
    bias + ReLU + accumulate; every image border of every tile; replay bit-identical."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c0, c1, co, var, up = case
    rng = np.random.RandomState(hash(case) % 2**31)
    hs, ws_ = (h // 2, w // 2) if up else (h, w)
    xlo = q(rng.randn(n, hs, ws_, c0), dtype)
    skip = q(rng.randn(n, h, w, c1), dtype) if c1 else None
    wt = q(rng.randn(3, 3, c0 + c1, co) / np.sqrt(9 * (c0 + c1)), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd, sd = dev(xlo, dtype), (dev(skip, dtype) if c1 else None)
    xhi = xlo.repeat(2, axis=1).repeat(2, axis=2) if up else xlo
    want = np_ops.conv2d(np.concatenate([xhi, skip], axis=-1) if c1 else xhi, wt, 1, 1)

    def run(tile, stats=False, **kw):
        y = kw.pop("y", None)
        if y is None:
            y = torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(xd, fwd, y, N=n, Hs0=hs, Ws0=ws_, Hv=h, Wv=w, C0=c0, C1=c1, src1=sd, mode=ops.SRC_NEAREST2X if up else ops.SRC_DIRECT,
                            KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), tile=tile, **kw)
        st = None
        if stats:
            st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
            P.stats_partial = ops.ptr(st)
        ops.conv2d(P)
        return y, st, P
    y, st, P = run(1024 + var, stats=True)
    assert _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 1024 + var
    np.testing.assert_allclose(host(y), want, atol=tol(want, dtype))
    y_tap, _, _ = run(69)                                                      # the per-tap kernel on the same buffers
    np.testing.assert_allclose(host(y), host(y_tap), atol=tol(want, dtype))
    tiles = ops.conv2d_stats_floats(P) // (2 * co)
    assert tiles == P.stats_tiles == n * (h // {0: 16, 1: 8, 2: 16, 3: 8}[var]) * (w // 16)
    part = host(st)[:2 * co * tiles].reshape(2, co, tiles).astype(np.float64).sum(-1)
    yv = host(y).reshape(-1, co).astype(np.float64)
    np.testing.assert_allclose(part[0], yv.sum(0), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(part[1], (yv * yv).sum(0), rtol=1e-4, atol=1e-2)
    y2, st2, _ = run(1024 + var, stats=True)
    assert torch.equal(y2, y) and torch.equal(st2[:2 * co * tiles], st[:2 * co * tiles])
    base = q(rng.randn(n, h, w, co), dtype)
    bias = keep(torch.from_numpy(rng.randn(co).astype(np.float32)).to(DEV))
    y3, _, _ = run(1024 + var, y=dev(base, dtype), bias=bias, relu=1, accumulate0=1)
    ref3 = np.maximum(want + host(bias) + base, 0.0)
    np.testing.assert_allclose(host(y3), ref3, atol=tol(ref3, dtype))


GROUP_CASES = [
    # n, h, w, ci, co, tile, expected G: more than 128 statistic columns -> the last workgroup of every G tiles pre-reduces them
    (3, 56, 112, 64, 64, 1027, 2),       # 147 tiles of 8 x 16: 74 groups, the last one short
    (2, 128, 128, 64, 256, 1025, 2),     # 256 tiles x 2 channel tiles
    (5, 64, 128, 64, 64, 1027, 4),       # 320 tiles -> G = 4
    (2, 64, 128, 64, 64, 1026, 0),       # 64 tiles: nothing to do
    (3, 80, 72, 64, 128, 69, 4),         # the buffer-DMA kernel's row-major epilogue: 17280 pixels / 64 = 270 tiles
]


@pytest.mark.parametrize("case", GROUP_CASES)
@pytest.mark.parametrize("bnb", [False, True])
@pytest.mark.parametrize("dtype", H16)
def test_statistic_columns_pre_reduced_by_the_last_workgroup_of_a_group(ops, case, bnb, dtype, monkeypatch):
    """stp_conv_params.stats_group (round 5; opt-in, STP_STATS_GROUP=1 - measured slower than the finalize launches it removes): the group table equals the tile-order fp32 sum of the group's columns of stats_partial
    EXACTLY (fixed membership, fixed order - whoever arrives last), the arrival counters are back at zero, a second launch is
    bit-identical, and stp_bn_finalize over the group table equals stp_bn_finalize over the full table (forward statistics and the
    BatchNormalization-backward sums)."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, tile, G_want = case
    if os.environ.get("STP_STATS_GROUP") != "1":
        pytest.skip("the switch is read once per process: run with STP_STATS_GROUP=1 (tests/test_model_gpu.py runs this file that way)")
    rng = np.random.RandomState(hash(case) % 2**31)
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    f32 = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    bx = dev(q(rng.randn(n, h, w, co), dtype), dtype) if bnb else None
    mean, rstd, gamma, beta = (f32(rng.randn(co) * 0.1), f32(rng.rand(co) + 0.5), f32(rng.rand(co) + 0.5), f32(rng.randn(co) * 0.1)) if bnb else (None,) * 4

    def run(group):
        y = torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(xd, fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), tile=tile)
        if bnb:
            P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(bx), ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), 1
        st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
        P.stats_partial = ops.ptr(st)
        G = int(_lib.load().stp_conv2d_stats_group_for(ops.C.byref(P)))
        gt = cnt = None
        if group and G:
            cols = ops.conv2d_stats_floats(P) // (2 * co)
            ng = -(-cols // G)
            gt = torch.full((2 * co * ng,), float("nan"), dtype=torch.float32, device=DEV)
            cnt = torch.zeros(int(_lib.load().stp_conv2d_stats_group_counters(ops.C.byref(P), G)), dtype=torch.int32, device=DEV)
            P.stats_group_out, P.stats_group_counters, P.stats_group = ops.ptr(gt), ops.ptr(cnt), G
        ops.conv2d(P)
        return y, st, gt, cnt, G, P
    y, st, gt, cnt, G, P = run(True)
    assert G == G_want
    y0, st0, _, _, _, _ = run(False)
    assert torch.equal(y, y0)                                      # the grouped launch changes nothing else
    cols = ops.conv2d_stats_floats(P) // (2 * co)
    full = host(st)[:2 * co * cols].reshape(2 * co, cols)
    np.testing.assert_array_equal(full, host(st0)[:2 * co * cols].reshape(2 * co, cols))
    if not G:
        return
    ng = -(-cols // G)
    want = np.zeros((2 * co, ng), np.float32)
    for g in range(ng):
        acc = full[:, g * G].copy()
        for j in range(1, min(G, cols - g * G)):
            acc = (acc + full[:, g * G + j]).astype(np.float32)
        want[:, g] = acc
    np.testing.assert_array_equal(host(gt).reshape(2 * co, ng), want)
    assert int(cnt.abs().sum().item()) == 0                        # tickets returned: the next launch / graph replay starts from zero
    y2, st2, gt2, cnt2, _, _ = run(True)
    assert torch.equal(gt2, gt) and torch.equal(y2, y)
    if not bnb:
        rows = n * h * w
        m1, r1, m0, r0 = (torch.empty(co, device=DEV) for _ in range(4))
        _lib.call("stp_bn_finalize", ops.ptr(gt), ng, rows, co, 1e-3, 0.99, ops.ptr(m1), ops.ptr(r1), None, None, ops.stream())
        _lib.call("stp_bn_finalize", ops.ptr(st), cols, rows, co, 1e-3, 0.99, ops.ptr(m0), ops.ptr(r0), None, None, ops.stream())
        np.testing.assert_allclose(host(m1), host(m0), atol=2e-6 * max(1.0, np.abs(host(m0)).max()))
        np.testing.assert_allclose(host(r1), host(r0), rtol=2e-5)
    # a group size the kernel would not choose is refused
    P.stats_group = 16 if G != 16 else 8
    with pytest.raises(_lib.StpError):
        ops.conv2d(P)


@pytest.mark.parametrize("case", [(2, 16, 32, 128, 128, 0, 1), (1, 32, 16, 256, 64, 0, 0), (2, 8, 16, 192, 128, 1, 1), (1, 16, 16, 64, 64, 2, 1),
                                  (2, 8, 32, 512, 64, 3, 2)])
@pytest.mark.parametrize("dtype", H16)
def test_conv_halo_kernel_fused_producer_batchnorm(ops, case, dtype):
    """stp_conv_params.src_bn_* on the halo kernel: the pre-BatchNormalization tensor is normalised (+ activation) IN LDS, once
    per slab, by the thread that staged it.  Bit-identical to stp_bn_apply followed by the same kernel (same fma, activation and
    bf16 rounding; padding pixels stay zero), incl. the fused statistics of the output."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, var, relu = case
    rng = np.random.RandomState(79)
    rows = n * h * w
    ypre = q(rng.randn(n, h, w, ci) * 2 + 0.5, dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    mean, rstd = f(ypre.reshape(-1, ci).mean(0)), f(1.0 / np.sqrt(ypre.reshape(-1, ci).var(0) + 1e-3))
    gamma, beta = f(rng.rand(ci) + 0.5), f(rng.randn(ci) * 0.3 + 0.2)      # beta > 0 on average: act(shift) != 0 where padding must be 0
    yd = dev(ypre, dtype)
    act = torch.empty_like(yd)
    ops.bn_apply(yd, act, rows, ci, ci, mean, rstd, gamma, beta, relu=relu)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    mk = lambda src, dst: ops.conv_params(src, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                          Cout=co, dtype=ops.dt(dst), tile=1024 + var)
    ref, got = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV), torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
    P0, P1 = mk(act, ref), mk(yd, got)
    P1.src_bn_mean, P1.src_bn_rstd, P1.src_bn_gamma, P1.src_bn_beta, P1.src_bn_relu = ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), relu
    st0 = torch.zeros((max(4, ops.conv2d_stats_floats(P0)),), dtype=torch.float32, device=DEV)
    st1 = torch.full_like(st0, float("nan"))
    P0.stats_partial, P1.stats_partial = ops.ptr(st0), ops.ptr(st1)
    ops.conv2d(P0)
    ops.conv2d(P1)
    np.testing.assert_array_equal(host(got), host(ref))
    np.testing.assert_array_equal(host(st1), host(st0))
    np.testing.assert_allclose(host(ref), np_ops.conv2d(host(act), wt, 1, 1), atol=tol(host(ref), dtype))
    P1.tile = 69                                                           # the per-tap kernel must refuse the fields
    assert _lib.load().stp_conv2d(P1, ops.stream()) == -1


@pytest.mark.parametrize("case", [(2, 16, 32, 128, 128, 0), (2, 8, 16, 64, 256, 1), (1, 32, 16, 128, 64, 2), (2, 8, 32, 64, 64, 3),
                                  (2, 32, 32, 64, 64, 5), (6, 128, 112, 64, 64, 5)])      # (5: persistent 64 -> 64 form; 672 tiles = two rounds)
@pytest.mark.parametrize("relu", [1, 0, 3])
@pytest.mark.parametrize("dtype", H16)
def test_conv_halo_kernel_batchnorm_backward_sums(ops, case, relu, dtype):
    """bnb_x epilogue of the halo kernel == the same epilogue of the per-tap DMA kernel (masked gradient bit for bit where the
    two accumulation orders round alike, sums to rounding), incl. accumulate0 as the LAST consumer."""
    n, h, w, ci, co, var = case
    last = relu == 3
    relu = 1 if last else relu
    rng = np.random.RandomState(78)
    rows = n * h * w
    src = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    x = q(rng.randn(n, h, w, co) * 1.5 + 0.3, dtype)
    others = q(rng.randn(n, h, w, co), dtype)
    gamma, beta = (rng.rand(co) + 0.5).astype(np.float32), (rng.randn(co) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, co, 1e-3, 0.99, m, r, None, None, ws)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    sd = dev(src, dtype)
    outs = []
    for tile in (69, 1024 + var):
        gbuf = dev(others, dtype) if last else torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(sd, fwd, gbuf, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co,
                            dtype=ops.dt(gbuf), tile=tile, accumulate0=int(last))
        P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), relu
        st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
        P.stats_partial = ops.ptr(st)
        ops.conv2d(P)
        tiles = ops.conv2d_stats_floats(P) // (2 * co)
        assert tiles == P.stats_tiles
        dx, dg, db = torch.empty_like(gbuf), torch.empty(co, device=DEV), torch.empty(co, device=DEV)
        ops.bn_backward_fused(xd, gbuf, dx, rows, co, m, r, g, st, tiles, dg, db, accumulate_dx=0, workspace=ws)
        outs.append((host(gbuf), host(dx), host(dg), host(db)))
    (g0, dx0, dg0, db0), (g1, dx1, dg1, db1) = outs
    np.testing.assert_allclose(g1, g0, atol=tol(g0, dtype))
    sc = lambda a: 1e-3 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(db1, db0, atol=sc(db0) * 5)
    np.testing.assert_allclose(dg1, dg0, atol=sc(dg0) * 5)
    np.testing.assert_allclose(dx1, dx0, atol=tol(dx0, dtype))


@pytest.mark.parametrize("case", [(2, 16, 32, 64, 128, 64, 0), (1, 16, 16, 128, 256, 128, 1), (1, 32, 16, 64, 128, 64, 4), (2, 8, 16, 64, 64, 128, 3)])
@pytest.mark.parametrize("relu", [1, 0])
@pytest.mark.parametrize("dtype", H16)
def test_conv_halo_kernel_two_destinations_with_summed_upsampling_gradient(ops, case, relu, dtype):
    """dst_sum2x2 + dst1 on the halo kernel (EP 3): the data gradient of conv3x3(concat(UpSampling2D(2)(x), skip)) - the channel tiles of
    the upsampled source are summed over 2 x 2 pixel blocks in the epilogue and receive the fused BatchNormalization backward of the
    LOW-resolution tensor (mask + [2][Cd0][tiles] sums), the skip channels go to dst1 (accumulated) at full resolution.  Against float64
    numpy (conv2d_dgrad -> upsample2x_bwd -> mask / BatchNormalization backward), and the dx of stp_bn_backward_fused on the table."""
    n, h, w, ci, cd0, cd1, var = case                  # h, w: the (virtual) high-resolution size = dY's size
    rng = np.random.RandomState(91)
    co = cd0 + cd1
    dy = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)                    # data-gradient weights in forward form: dY (ci) -> d(input) (co)
    xlow = q(rng.randn(n, h // 2, w // 2, cd0) * 1.5 + 0.3, dtype)              # BatchNormalization input of the upsampled tensor
    skip_prev = q(rng.randn(n, h, w, cd1), dtype)                               # gradient already in the skip tensor's buffer
    gamma, beta = (rng.rand(cd0) + 0.5).astype(np.float32), (rng.randn(cd0) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    rows_low = n * (h // 2) * (w // 2)
    xd, g, b = dev(xlow, dtype), f(gamma), f(beta)
    m, r = torch.empty(cd0, device=DEV), torch.empty(cd0, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(cd0) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows_low, cd0, 1e-3, 0.99, m, r, None, None, ws)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    d0 = torch.full((n, h // 2, w // 2, cd0), float("nan"), dtype=TD[dtype], device=DEV)
    d1 = dev(skip_prev, dtype)
    P = ops.conv_params(dev(dy, dtype), fwd, d0, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co,
                        dtype=ops.dt(d0), tile=1024 + var, dst1=d1, Cd0=cd0, accumulate1=1)
    P.dst_sum2x2 = 1
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), relu
    nfl = ops.conv2d_stats_floats(P)
    st = torch.full((max(4, nfl),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    tiles = nfl // (2 * cd0)
    assert tiles == P.stats_tiles and tiles > 0
    # numpy: full-resolution data gradient, then the two halves
    full = np_ops.conv2d(dy, wt, 1, 1)
    low = np_ops.upsample2x_bwd(full[..., :cd0])
    mean, rstd = host(m).astype(np.float64), host(r).astype(np.float64)
    pre = (xlow - mean) * rstd * gamma + beta
    mask = (pre > 0) if relu else np.ones_like(pre, bool)
    gref = low * mask
    got = host(d0)
    near = np.abs(pre) < 1e-2 * (np.abs(pre).max() + 1.0)                       # a pre-activation within rounding of 0 may take either side
    np.testing.assert_allclose(got[~near], gref[~near], atol=tol(low, dtype))
    np.testing.assert_allclose(host(d1), skip_prev + full[..., cd0:], atol=tol(full, dtype, 2))
    dx, dg, db = torch.empty_like(d0), torch.empty(cd0, device=DEV), torch.empty(cd0, device=DEV)
    ops.bn_backward_fused(xd, d0, dx, rows_low, cd0, m, r, g, st, tiles, dg, db, accumulate_dx=0, workspace=ws)
    gh = host(d0).astype(np.float64)                                            # the sums must be those of the STORED masked gradient
    xh = (xlow - mean) * rstd
    sc = lambda a: 2e-3 * np.abs(a).max() + 1e-3
    np.testing.assert_allclose(host(db), gh.sum(axis=(0, 1, 2)), atol=sc(gh.sum(axis=(0, 1, 2))))
    np.testing.assert_allclose(host(dg), (gh * xh).sum(axis=(0, 1, 2)), atol=sc((gh * xh).sum(axis=(0, 1, 2))))
    dxr = gamma * rstd * (gh - gh.mean(axis=(0, 1, 2)) - xh * (gh * xh).mean(axis=(0, 1, 2)))
    np.testing.assert_allclose(host(dx), dxr, atol=tol(dxr, dtype, 2))
    # a channel tile must lie in one destination: Cd0 = 96 cannot take 64- or 128-channel tiles
    P.Cd0 = cd0 - 32
    assert ops._lib.load().stp_conv2d_halo_variant(ops.C.byref(P)) == -1
    # ONE destination, every channel tile summed (the data gradient of conv3x3(UpSampling2D(2)(x)), Linknet's decoder): Cd0 == Cout
    wt1 = wt[..., :cd0].copy()
    _, fwd1, _, _ = prep_weights(ops, wt1, dtype)
    e0 = torch.full((n, h // 2, w // 2, cd0), float("nan"), dtype=TD[dtype], device=DEV)
    P1 = ops.conv_params(dev(dy, dtype), fwd1, e0, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=cd0,
                         dtype=ops.dt(e0), tile=1024 + var)
    P1.dst_sum2x2 = 1
    P1.bnb_x, P1.bnb_mean, P1.bnb_rstd, P1.bnb_gamma, P1.bnb_beta, P1.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), relu
    st1 = torch.full((max(4, ops.conv2d_stats_floats(P1)),), float("nan"), dtype=torch.float32, device=DEV)
    P1.stats_partial = ops.ptr(st1)
    ops.conv2d(P1)
    assert np.array_equal(host(e0), got) and np.array_equal(host(st1), host(st))        # same tiles, same arithmetic as the two-destination launch


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 32, 36), (1, 58, 268)])      # the second: interior tiles of the persistent stem kernels (forward and weight gradient)
def test_stem_conv_7x7_s2_padded_channels(ops, dtype, shape):
    """conv0: 7x7/2 over a 3-channel image stored as 4 channels (4th = 1), weights padded to 7x8x4."""
    rng = np.random.RandomState(7)
    (n, h, w), co = shape, 64
    x3 = q(rng.randn(n, h, w, 3), dtype)
    wt = q(rng.randn(7, 7, 3, co) / 12.0, dtype)
    ref = np_ops.conv2d(x3, wt, 2, 3)
    ho, wo = ref.shape[1:3]
    x4 = np.concatenate([x3, np.ones((n, h, w, 1), np.float32)], axis=-1)
    _, fwd, _, _ = prep_weights(ops, wt, dtype, KWp=8, Cinp=4)
    y = torch.empty((n, ho, wo, co), dtype=TD[dtype], device=DEV)
    P = ops.conv_params(dev(x4, dtype), fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=4, KH=7, KW=8, stride=2, pad=3,
                        Ho=ho, Wo=wo, Cout=co, dtype=ops.dt(y))
    ops.conv2d(P)
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    # weight gradient through the same padded geometry + the bn_data beta gradient trick
    dy = q(rng.randn(n, ho, wo, co), dtype)
    refw = np_ops.conv2d_wgrad(x3, dy, (7, 7), 2, 3)
    dwp = torch.zeros((co, 7, 8, 4), dtype=torch.float32, device=DEV)
    W = ops.wgrad_params(dev(x4, dtype), dev(dy, dtype), dwp, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=4, KH=7, KW=8, stride=2,
                         pad=3, Ho=ho, Wo=wo, Cout=co, dtype=ops.dt(y))
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(W, ws)
    g = torch.empty((co, 7, 7, 3), dtype=torch.float32, device=DEV)
    ops.weight_grad_unpad(dwp, g, co, 7, 7, 3, 8, 4)
    np.testing.assert_allclose(host(g).transpose(1, 2, 3, 0), refw, atol=tol(refw, dtype))
    # d(sum over valid taps) : dbeta of an input BN with identity scale = sum_c dX[..., c]
    refdx = np_ops.conv2d_dgrad(dy, wt, (h, w), 2, 3)
    master = torch.from_numpy(np.ascontiguousarray(wt.transpose(3, 0, 1, 2))).to(DEV)
    db = torch.empty(3, dtype=torch.float32, device=DEV)
    ops.stem_beta_grad(dwp, master, db, co, 7, 7, 3, 8, 4, 3)
    refdb = refdx.sum(axis=(0, 1, 2))
    np.testing.assert_allclose(host(db), refdb, atol=tol(refdb, dtype, 4))


WGRAD_CASES = [
    # n, h, w, ci, co, k, stride, pad, splits
    (2, 16, 16, 32, 128, 3, 1, 1, 0),
    (2, 16, 16, 64, 64, 3, 1, 1, 3),
    (1, 24, 20, 16, 32, 3, 1, 1, 0),
    (1, 20, 24, 32, 16, 3, 1, 1, 5),
    (2, 12, 12, 128, 256, 3, 2, 1, 1),
    (2, 16, 16, 64, 128, 1, 2, 0, 0),
    (3, 7, 5, 24, 40, 3, 1, 1, 2),
    # small-channel halo-tile weight gradient (automatic for 16/32-channel 3x3 s1 layers), ragged tiles
    (2, 19, 45, 16, 16, 3, 1, 1, 0), (1, 9, 33, 32, 32, 3, 1, 1, 0), (2, 21, 40, 16, 32, 3, 1, 1, 0), (2, 19, 45, 32, 16, 3, 1, 1, 0),
    (1, 16, 64, 16, 8, 3, 1, 1, 0),
    # power-of-two feature maps: the DMA kernel's uniform-row addressing (one image row / whole rows per pixel step)
    (1, 2, 128, 64, 64, 3, 1, 1, 0), (2, 16, 32, 64, 64, 3, 2, 1, 0), (1, 8, 64, 128, 136, 3, 1, 1, 2), (3, 8, 8, 64, 192, 3, 1, 1, 0),
    # row-of-taps kernel (variant 4; automatic for bf16 when eligible): whole rows / row segments per pixel step, channel tails, splits
    (2, 32, 32, 128, 128, 3, 1, 1, 0), (1, 16, 64, 256, 72, 3, 1, 1, 0), (2, 4, 16, 64, 40, 3, 1, 1, 0), (1, 4, 192, 64, 64, 3, 1, 1, 3),
    (3, 16, 16, 192, 200, 3, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv2d_weight_gradient(ops, dtype, case):
    n, h, w, ci, co, k, s, p, splits = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = q(rng.randn(n, h, w, ci), dtype)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dy = q(rng.randn(n, ho, wo, co), dtype)
    ref = np_ops.conv2d_wgrad(x, dy, (k, k), s, p)  # HWIO
    dw = torch.full((co, k, k, ci), float("nan"), dtype=torch.float32, device=DEV)
    W = ops.wgrad_params(dev(x, dtype), dev(dy, dtype), dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=s,
                         pad=p, Ho=ho, Wo=wo, Cout=co, dtype=ops.dt(dev(x, dtype)), splits=splits)
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(W, ws)
    np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))
    # accumulate
    W.accumulate = 1
    ops.conv2d_wgrad(W, ws)
    np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), 2 * ref, atol=tol(ref, dtype, 2))
    # every kernel variant (1 = register-staged, 2/3 = buffer-DMA ring) through the two-phase entry points
    W.accumulate = 0
    for variant in (1, 2, 3):
        dw.fill_(float("nan"))
        ops.conv2d_wgrad_partial(W, ws, variant)
        ops.conv2d_wgrad_reduce(W, ws, variant)
        np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype), err_msg="variant %d" % variant)
    # variant 4 = row-of-taps kernel: runs on its shapes (bf16), refuses the others
    row_ok = (dtype != "fp32" and k == 3 and s == 1 and p == 1 and ci % 64 == 0 and co % 8 == 0 and
              (w % 64 == 0 or (w >= 16 and 64 % w == 0 and (h * w) % 64 == 0)))
    dw.fill_(float("nan"))
    if row_ok:
        ops.conv2d_wgrad_partial(W, ws, 4)
        ops.conv2d_wgrad_reduce(W, ws, 4)
        np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype), err_msg="variant 4")
        assert ops._lib.load().stp_conv2d_wgrad_kernel_id(ops.C.byref(W)) == ((3 if co <= 64 else 2) if splits == 0 else 0)
    else:
        with pytest.raises(Exception):
            ops.conv2d_wgrad_partial(W, ws, 4)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_batched_reduce_of_lone_weight_gradients(ops, dtype):
    """stp_wgrad_reduce_batched (round 6): the split-K reduces of several lone layers - the 1x1 convolutions of a bottleneck unit
    (classification_models residual_bottleneck_block, reference graph through segmentation.py:109-118), a 1x1 / stride-2 shortcut and a
    3x3 / stride-2 layer - as ONE launch over a descriptor table, each layer's slabs in a workspace of its own.  Against float64 numpy,
    against the per-layer stp_conv2d_wgrad_reduce, with accumulate, and bit-identical on replay; a row-of-taps layer (fragment-major
    slabs) is refused by the descriptor query and keeps its own reduce."""
    lib = ops._lib.load()
    # n, h, w, ci, co, k, stride, pad
    layers = [(2, 24, 24, 64, 256, 1, 1, 0), (2, 24, 24, 256, 64, 1, 1, 0), (2, 24, 24, 64, 128, 1, 2, 0), (2, 24, 24, 64, 64, 3, 2, 1),
              (1, 12, 12, 512, 128, 1, 1, 0)]
    rng = np.random.RandomState(11)
    db = int(lib.stp_wgrad_reduce_desc_bytes())
    table = (ops.C.c_char * (db * len(layers)))()
    Ws, refs, dws, wss, counts = [], [], [], [], []
    for i, (n, h, w, ci, co, k, st_, pd) in enumerate(layers):
        x = q(rng.randn(n, h, w, ci), dtype)
        ho, wo = (h + 2 * pd - k) // st_ + 1, (w + 2 * pd - k) // st_ + 1
        dy = q(rng.randn(n, ho, wo, co), dtype)
        refs.append(np_ops.conv2d_wgrad(x, dy, (k, k), st_, pd))
        dw = keep(torch.full((co, k, k, ci), float("nan"), dtype=torch.float32, device=DEV))
        xd, dyd = dev(x, dtype), dev(dy, dtype)
        W = ops.wgrad_params(xd, dyd, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=k, KW=k, stride=st_, pad=pd, Ho=ho, Wo=wo, Cout=co,
                             dtype=ops.dt(xd))
        ws = keep(torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV))
        c = int(lib.stp_wgrad_reduce_desc_fill(ops.C.addressof(table), i, ops.C.byref(W), ops.ptr(ws)))
        assert c == co * k * k * ci, (i, c)
        Ws.append(W); dws.append(dw); wss.append(ws); counts.append(c)
    tdev = keep(torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV))

    def run():
        for W, ws in zip(Ws, wss):
            ops.conv2d_wgrad_partial(W, ws, 0)
        ops._lib.check(lib.stp_wgrad_reduce_batched(ops.ptr(tdev), len(layers), max(counts), ops.stream()), "stp_wgrad_reduce_batched")
    run()
    first = [d.clone() for d in dws]
    for d, ref in zip(dws, refs):
        np.testing.assert_allclose(host(d).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))
    for d in dws:
        d.fill_(float("nan"))
    run()                                                      # replay: fixed walk, fixed tree
    assert all(torch.equal(a_, b_) for a_, b_ in zip(dws, first))
    for W, ws, d, f_ in zip(Ws, wss, dws, first):             # the per-layer reduce of the same slabs: same sums up to the order
        d.fill_(float("nan"))
        ops.conv2d_wgrad_reduce(W, ws, 0)
        np.testing.assert_allclose(host(d), host(f_), rtol=1e-5, atol=1e-5 * float(f_.abs().max()))
    # accumulate: the table records the flag at fill time
    for i, (W, ws) in enumerate(zip(Ws, wss)):
        W.accumulate = 1
        assert int(lib.stp_wgrad_reduce_desc_fill(ops.C.addressof(table), i, ops.C.byref(W), ops.ptr(ws))) == counts[i]
    tdev2 = keep(torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(DEV))
    for W, ws in zip(Ws, wss):
        ops.conv2d_wgrad_partial(W, ws, 0)
    ops._lib.check(lib.stp_wgrad_reduce_batched(ops.ptr(tdev2), len(layers), max(counts), ops.stream()), "stp_wgrad_reduce_batched")
    for d, ref in zip(dws, refs):
        np.testing.assert_allclose(host(d).transpose(1, 2, 3, 0), 2 * ref, atol=tol(ref, dtype, 2))
    if dtype != "fp32":      # a row-of-taps layer writes fragment-major slabs: not for the table
        n, h, w, c = 1, 16, 64, 64
        xd, dyd = dev(q(rng.randn(n, h, w, c), dtype), dtype), dev(q(rng.randn(n, h, w, c), dtype), dtype)
        dw = keep(torch.zeros((c, 3, 3, c), dtype=torch.float32, device=DEV))
        W = ops.wgrad_params(xd, dyd, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=c, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=c, dtype=ops.dt(xd))
        if lib.stp_conv2d_wgrad_kernel_id(ops.C.byref(W)) in (2, 3):
            assert int(lib.stp_wgrad_reduce_desc_fill(ops.C.addressof(table), 0, ops.C.byref(W), ops.ptr(wss[0]))) == 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("chans", [(32, 16, 64), (64, 64, 32), (32, 16, 16), (128, 64, 64, 8, 8), (64, 64, 128, 4, 32), (128, 128, 136, 8, 16),
                                   (64, 192, 72, 16, 64)])
def test_conv2d_weight_gradient_upsample_concat(ops, dtype, chans):
    rng = np.random.RandomState(9)
    n, h, w = 2, 6, 7
    if len(chans) == 5:           # power-of-two maps: uniform-row addressing with two sources
        h, w = chans[3:]
        chans = chans[:3]
    c0, c1, co = chans            # (64, 64, 32) = decoder_stage3_conv1: the small-channel kernel, one launch per source
    x = q(rng.randn(n, h, w, c0), dtype)
    skip = q(rng.randn(n, 2 * h, 2 * w, c1), dtype)
    dy = q(rng.randn(n, 2 * h, 2 * w, co), dtype)
    v = np.concatenate([np_ops.upsample2x(x), skip], axis=-1)
    ref = np_ops.conv2d_wgrad(v, dy, (3, 3), 1, 1)
    dw = torch.empty((co, 3, 3, c0 + c1), dtype=torch.float32, device=DEV)
    W = ops.wgrad_params(dev(x, dtype), dev(dy, dtype), dw, N=n, Hs0=h, Ws0=w, Hv=2 * h, Wv=2 * w, C0=c0, C1=c1,
                         src1=dev(skip, dtype), mode=ops.SRC_NEAREST2X, KH=3, KW=3, stride=1, pad=1, Ho=2 * h, Wo=2 * w,
                         Cout=co, dtype=ops.dt(dev(x, dtype)))
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(W, ws)
    np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))
    # the 64-channel-block shapes on power-of-two maps take the row-of-taps kernel in bf16 (two sources, the first upsampled)
    wo, howo = 2 * w, 4 * h * w
    row = dtype != "fp32" and c0 % 64 == 0 and c1 % 64 == 0 and (wo % 64 == 0 or (wo >= 16 and 64 % wo == 0 and howo % 64 == 0))
    lib = ops._lib.load()
    assert lib.stp_conv2d_wgrad_kernel_id(ops.C.byref(W)) == (1 if lib.stp_wgrad_sc_eligible(ops.C.byref(W)) else (3 if co <= 64 else 2) if row else 0)
    if row:
        dw.fill_(float("nan"))
        ops.conv2d_wgrad_partial(W, ws, 2); ops.conv2d_wgrad_reduce(W, ws, 2)      # the pixel-reduction GEMM on the same shape
        np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))


# (n, h, w, c0, c1 (c1 > 0: src0 is nearest-2x upsampled and concatenated with a c1-channel skip), cout) per layer; one class per group
WGRAD_GROUPS = {
    "class128": [(2, 32, 32, 128, 0, 128), (1, 16, 64, 256, 0, 72), (3, 16, 16, 192, 0, 200), (2, 8, 8, 128, 64, 136), (1, 4, 192, 64, 0, 128)],
    "class64": [(2, 32, 32, 64, 0, 64), (2, 4, 16, 64, 0, 40), (2, 4, 32, 128, 64, 64)],
    "class32": [(2, 32, 32, 64, 0, 32), (1, 4, 32, 64, 64, 24), (1, 16, 64, 128, 0, 16)],
    "one_layer": [(4, 32, 32, 128, 0, 256)],
}


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("group", sorted(WGRAD_GROUPS))
def test_grouped_weight_gradient(ops, dtype, group):
    """stp_wgrad_group_*: one partial + one reduce launch for several row-of-taps layers (work list cut into one chunk per
    workgroup slot) against the numpy oracle, layer by layer; replay is bit-identical; accumulate; foreign layers are refused."""
    rng = np.random.RandomState(11)
    layers, refs, dws = [], [], []
    for n, h, w, c0, c1, co in WGRAD_GROUPS[group]:
        up = c1 > 0
        x = q(rng.randn(n, h, w, c0), dtype)
        H, W_ = (2 * h, 2 * w) if up else (h, w)
        skip = q(rng.randn(n, H, W_, c1), dtype) if up else None
        dy = q(rng.randn(n, H, W_, co), dtype)
        v = np.concatenate([np_ops.upsample2x(x), skip], axis=-1) if up else x
        refs.append(np_ops.conv2d_wgrad(v, dy, (3, 3), 1, 1))
        dw = torch.full((co, 3, 3, c0 + c1), float("nan"), dtype=torch.float32, device=DEV)
        dws.append(dw)
        layers.append(ops.wgrad_params(dev(x, dtype), dev(dy, dtype), dw, N=n, Hs0=h, Ws0=w, Hv=H, Wv=W_, C0=c0, C1=c1,
                                       src1=dev(skip, dtype) if up else None, mode=ops.SRC_NEAREST2X if up else ops.SRC_DIRECT, KH=3, KW=3,
                                       stride=1, pad=1, Ho=H, Wo=W_, Cout=co, dtype=ops.dt(dev(x, dtype))))
    want = {"class128": 128, "class64": 64, "class32": 32, "one_layer": 128}[group]
    assert [ops.wgrad_group_class(p) for p in layers] == [want] * len(layers)
    g = ops.WgradGroup(layers)
    magic, bm, nl, nseg, nwg, ntile = g.header[:6]
    assert (bm, nl) == (want, len(layers)) and nseg >= ntile >= nl and 0 < nwg <= nseg
    g.run()
    first = [host(dw).copy() for dw in dws]
    for dw, ref in zip(first, refs):
        np.testing.assert_allclose(dw.transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))
    g.run()                                         # same table, same partition: bit-identical
    for dw, a in zip(dws, first):
        assert np.array_equal(host(dw), a)
    for p in layers:
        p.accumulate = 1
    g2 = ops.WgradGroup(layers)
    g2.run()
    for dw, ref in zip(dws, refs):
        np.testing.assert_allclose(host(dw).transpose(1, 2, 3, 0), 2 * ref, atol=tol(ref, dtype, 2))
    # a layer of another class, or one the row-of-taps kernel does not take, cannot join
    odd = ops.wgrad_params(dev(np.zeros((1, 16, 16, 32)), dtype), dev(np.zeros((1, 16, 16, 64)), dtype), dws[0], N=1, Hs0=16, Ws0=16, Hv=16, Wv=16,
                           C0=32, KH=3, KW=3, stride=1, pad=1, Ho=16, Wo=16, Cout=64, dtype=ops.dt(dev(np.zeros(1), dtype)))
    assert ops.wgrad_group_class(odd) == 0
    with pytest.raises(Exception):
        ops.WgradGroup(layers + [odd])


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("group", ["class128", "class64"])
def test_grouped_weight_gradient_with_fused_producer_batchnorm(ops, dtype, group):
    """stp_wgrad_group_* where SOME layers carry stp_wgrad_params.src_bn_* (their src0 is the tensor before a BatchNormalization +
    activation, normalised in LDS by the grouped kernel's PBN instance): against the numpy oracle on the normalised tensors, and
    bit-identical to the same group run on tensors that stp_bn_apply normalised first (same fma, activation, rounding; the zero
    padding applies to the normalised tensor)."""
    rng = np.random.RandomState(12)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    fused_layers, plain_layers, refs, dws_f, dws_p = [], [], [], [], []
    shapes = [s for s in WGRAD_GROUPS[group] if s[4] == 0]      # a fused producer BN needs a single directly-read source
    for li, (n, h, w, c0, _, co) in enumerate(shapes):
        relu = li % 3                                           # none / ReLU / ReLU6
        pre = q(rng.randn(n, h, w, c0) * 2 + 0.5, dtype)
        mean, rstd = f(pre.reshape(-1, c0).mean(0)), f(1.0 / np.sqrt(pre.reshape(-1, c0).var(0) + 1e-3))
        gamma, beta = f(rng.rand(c0) + 0.5), f(rng.randn(c0) * 0.3)
        pd = dev(pre, dtype)
        act = keep(torch.empty_like(pd))                        # (the parameter blocks hold raw pointers)
        ops.bn_apply(pd, act, n * h * w, c0, c0, mean, rstd, gamma, beta, relu=relu)
        dy = dev(q(rng.randn(n, h, w, co), dtype), dtype)
        refs.append(np_ops.conv2d_wgrad(host(act), host(dy), (3, 3), 1, 1))
        fuse = li != 1                                          # one layer of the group stays unfused (per-layer switch in the kernel)
        for lst, dwl, src, fz in ((fused_layers, dws_f, pd if fuse else act, fuse), (plain_layers, dws_p, act, False)):
            dw = torch.full((co, 3, 3, c0), float("nan"), dtype=torch.float32, device=DEV)
            W = ops.wgrad_params(src, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=c0, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(src))
            if fz:
                W.src_bn_mean, W.src_bn_rstd, W.src_bn_gamma, W.src_bn_beta, W.src_bn_relu = ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), relu
            lst.append(W)
            dwl.append(dw)
    want = 128 if group == "class128" else 64
    assert [ops.wgrad_group_class(p) for p in fused_layers] == [want] * len(fused_layers)
    gf, gp = ops.WgradGroup(fused_layers), ops.WgradGroup(plain_layers)
    assert gf.header[14] == 1 and gp.header[14] == 0             # WgGroupHeader.pbn: the fused-producer instance is selected per group
    gf.run(); gp.run()
    same_kernel = gf.header[15] == gp.header[15]               # (the plain 128-channel group runs on the all-taps kernel: another summation order)
    for a, b, ref in zip(dws_f, dws_p, refs):
        np.testing.assert_allclose(host(b).transpose(1, 2, 3, 0), ref, atol=tol(ref, dtype))
        if same_kernel:
            assert np.array_equal(host(a), host(b))
        else:
            np.testing.assert_allclose(host(a), host(b), atol=2e-5 * float(np.abs(ref).max()))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("C", [16, 64, 768])
def test_batchnorm_train_forward_backward(ops, dtype, C):
    rng = np.random.RandomState(10)
    n, h, w = 2, 12, 10
    rows = n * h * w
    x = q(rng.randn(n, h, w, C) * 2 + 0.5, dtype)
    gamma = (rng.rand(C) + 0.5).astype(np.float32)
    beta = rng.randn(C).astype(np.float32) * 0.3
    eps, mom = 1e-3, 0.99
    yref, mean, var = np_ops.bn_train(x, gamma, beta, eps)
    yref = np.maximum(yref, 0)
    xd = dev(x, dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    m, r = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    mm, mv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(C) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, C, eps, mom, m, r, mm, mv, ws)
    np.testing.assert_allclose(host(m), mean, atol=1e-4)
    np.testing.assert_allclose(host(r), 1 / np.sqrt(var + eps), rtol=1e-4)
    np.testing.assert_allclose(host(mm), mean * 0.01, atol=1e-5)
    np.testing.assert_allclose(host(mv), 0.99 + 0.01 * var * rows / (rows - 1), rtol=1e-4)
    y = torch.empty_like(xd)
    g, b = f(gamma), f(beta)
    ops.bn_apply(xd, y, rows, C, C, m, r, g, b, relu=1)
    np.testing.assert_allclose(host(y), yref, atol=tol(yref, dtype))
    # backward through ReLU + BN
    dy = q(rng.randn(n, h, w, C), dtype)
    pre, _, _ = np_ops.bn_train(x, gamma, beta, eps)
    dxr, dgr, dbr = np_ops.bn_train_bwd(x, dy * (pre > 0), gamma, eps)
    dx = torch.empty_like(xd)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_backward(xd, dev(dy, dtype), dx, rows, C, m, r, g, b, dg, db, relu=1, accumulate_dx=0, workspace=ws)
    np.testing.assert_allclose(host(db), dbr, atol=2e-3 * np.abs(dbr).max() + 1e-3)
    np.testing.assert_allclose(host(dg), dgr, atol=2e-3 * np.abs(dgr).max() + 1e-3)
    safe = np.abs(pre) > 1e-4  # a pre-activation within rounding of 0 may take either side of the ReLU
    np.testing.assert_allclose(host(dx)[safe], dxr[safe], atol=tol(dxr, dtype, 2))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128, 0), (2, 9, 11, 128, 64, 5), (2, 19, 45, 16, 16, 512), (2, 19, 45, 16, 32, 512),
                                  (1, 24, 20, 16, 64, 258), (2, 13, 9, 64, 24, 0), (1, 20, 24, 128, 16, 68),
                                  # interior tiles of the lean small-channel kernel (3+ tiles each way, ragged right / bottom edge)
                                  (2, 43, 139, 16, 16, 512), (1, 40, 136, 32, 32, 512), (1, 35, 130, 8, 16, 512)])
@pytest.mark.parametrize("relu", [1, 0, 3])
def test_batchnorm_backward_sums_fused_in_conv_epilogue(ops, dtype, case, relu):
    """stp_conv_params.bnb_x: the convolution that produces dY of a BN(+ReLU) output masks it and reduces the
    BatchNormalization-backward sums in its epilogue; stp_bn_backward_fused must then equal conv + stp_bn_backward."""
    n, h, w, ci, co, tile = case
    if dtype == "fp32" and tile == 512 and ci > 16:
        pytest.skip("fp32 small-channel kernel: Cin <= 16")
    last = relu == 3            # relu 3 = ReLU with this convolution as the LAST of several consumers: accumulate0 on top of theirs
    relu = 1 if last else relu
    rng = np.random.RandomState(77)
    rows = n * h * w
    src = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    x = q(rng.randn(n, h, w, co) * 1.5 + 0.3, dtype)          # the BN input
    others = q(rng.randn(n, h, w, co), dtype)                  # what the other consumers already wrote into dY
    gamma = (rng.rand(co) + 0.5).astype(np.float32)
    beta = (rng.randn(co) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, co, 1e-3, 0.99, m, r, None, None, ws)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    sd = dev(src, dtype)
    mk = lambda dst: ops.conv_params(sd, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                     Cout=co, dtype=ops.dt(dst), tile=tile, accumulate0=int(last))
    # unfused reference on the device
    dy = dev(others, dtype) if last else torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
    ops.conv2d(mk(dy))
    dx0 = torch.empty_like(dy)
    dg0, db0 = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ops.bn_backward(xd, dy, dx0, rows, co, m, r, g, b, dg0, db0, relu=relu, accumulate_dx=0, workspace=ws)
    # fused
    gbuf = dev(others, dtype) if last else torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
    P = mk(gbuf)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), relu
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    tiles = ops.conv2d_stats_floats(P) // (2 * co)
    assert tiles == P.stats_tiles
    dx1 = torch.empty_like(dy)
    dg1, db1 = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ops.bn_backward_fused(xd, gbuf, dx1, rows, co, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    # the stored gradient is dY under the ReLU mask, bit for bit
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    want = host(dy) * ((pre > 0) if relu else 1.0)
    np.testing.assert_array_equal(host(gbuf)[safe], want[safe])
    scale = lambda a: 2e-4 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(host(db1), host(db0), atol=scale(host(db0)) * 5)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=scale(host(dg0)) * 5)
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 0.5))


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("case", [(4096, 128, 32), (640, 64, 5), (16384, 256, 128), (333, 192, 7), (2048, 512, 96)])
def test_batchnorm_finalize_fused_into_the_apply_pass(ops, dtype, case):
    """stp_bn_finalize_apply (one launch: every workgroup reduces the partial sums of its own 64-channel slab, then normalises its
    rows) against stp_bn_finalize + stp_bn_apply; and the one-launch form of stp_bn_backward_fused_add - with the accumulated addend
    in ANOTHER buffer, left intact - against the float64 formula."""
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    rows, C, tiles = case
    rng = np.random.RandomState(rows + C)
    x = q(rng.randn(rows, C) * 1.3 + 0.2, dtype)
    bounds = np.linspace(0, rows, tiles + 1).astype(int)
    part = np.zeros((2, C, tiles), np.float32)
    for t in range(tiles):
        blk = x[bounds[t]:bounds[t + 1]].astype(np.float64)
        part[0, :, t], part[1, :, t] = blk.sum(0), (blk * blk).sum(0)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    gamma, beta = f(rng.rand(C) + 0.5), f(rng.randn(C) * 0.3)
    xd, pd = dev(x, dtype), f(part)
    assert lib.stp_bn_finalize_apply_ok(ops.dt(xd), rows, C, tiles) == 1
    outs = []
    for fused in (0, 1):
        m, r = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        mm, mv = f(np.full(C, 0.25)), f(np.full(C, 2.0))
        y = torch.full((rows, C), float("nan"), dtype=TD[dtype], device=DEV)
        if fused:
            _lib.call("stp_bn_finalize_apply", ops.ptr(pd), tiles, ops.ptr(xd), ops.ptr(y), ops.dt(xd), rows, C, 1e-3, 0.9, ops.ptr(m), ops.ptr(r),
                      ops.ptr(mm), ops.ptr(mv), ops.ptr(gamma), ops.ptr(beta), 1, ops.stream())
        else:
            _lib.call("stp_bn_finalize", ops.ptr(pd), tiles, rows, C, 1e-3, 0.9, ops.ptr(m), ops.ptr(r), ops.ptr(mm), ops.ptr(mv), ops.stream())
            ops.bn_apply(xd, y, rows, C, C, m, r, gamma, beta, relu=1)
        outs.append([host(t) for t in (m, r, mm, mv, y)])
    for a, b in zip(outs[0][:4], outs[1][:4]):
        np.testing.assert_allclose(b, a, rtol=2e-6, atol=1e-7)
    y0, y1 = outs[0][4], outs[1][4]
    assert np.isfinite(y1).all()
    np.testing.assert_allclose(y1, y0, atol=tol(y0, dtype, 0.6))        # (a constant one float ulp apart can flip a 16-bit rounding)
    assert (y1 != y0).mean() < 2e-3
    mean, rstd = outs[0][0].astype(np.float64), outs[0][1].astype(np.float64)
    # backward: g is the masked gradient, partial its per-tile sums of g and g * xhat
    g = q(rng.randn(rows, C), dtype)
    add = q(rng.randn(rows, C), dtype)
    xh = (x.astype(np.float64) - mean) * rstd
    pb = np.zeros((2, C, tiles), np.float32)
    for t in range(tiles):
        sl = slice(bounds[t], bounds[t + 1])
        pb[0, :, t], pb[1, :, t] = g[sl].astype(np.float64).sum(0), (g[sl] * xh[sl]).sum(0)
    S, Q = pb[0].astype(np.float64).sum(1), pb[1].astype(np.float64).sum(1)
    gm = host(gamma).astype(np.float64)
    want = gm * rstd * (g - S / rows - xh * (Q / rows))
    md, rd = f(mean), f(rstd)
    ws = torch.empty(ops.bn_workspace_bytes(C) // 4, dtype=torch.float32, device=DEV)
    gd, addd, pbd = dev(g, dtype), dev(add, dtype), f(pb)
    for acc in (0, 1):
        dx = torch.full((rows, C), float("nan"), dtype=TD[dtype], device=DEV)
        dg, db = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        _lib.call("stp_bn_backward_fused_add", ops.ptr(xd), ops.ptr(gd), ops.ptr(dx), ops.ptr(addd), ops.dt(xd), rows, C, ops.ptr(md), ops.ptr(rd),
                  ops.ptr(gamma), ops.ptr(pbd), tiles, ops.ptr(dg), ops.ptr(db), acc, ops.ptr(ws), ws.numel() * 4, ops.stream())
        ref = want + (add if acc else 0.0)
        np.testing.assert_allclose(host(dx), ref, atol=tol(ref, dtype))
        np.testing.assert_allclose(host(db), S, rtol=1e-5, atol=1e-4)
        np.testing.assert_allclose(host(dg), Q, rtol=1e-5, atol=1e-4)
        assert np.array_equal(host(addd), add)                          # the addend is read, never written
    # in place (dadd == dx): the form stp_bn_backward_fused keeps
    dx = dev(add, dtype)
    ops.bn_backward_fused(xd, gd, dx, rows, C, md, rd, gamma, pbd, tiles, dg, db, accumulate_dx=1, workspace=ws)
    np.testing.assert_allclose(host(dx), want + add, atol=tol(want + add, dtype))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 16, 16, 64, 128, 0), (2, 19, 45, 16, 16, 512), (2, 13, 9, 64, 24, 0), (3, 32, 32, 64, 512, 0)])
def test_batchnorm_sums_in_fixed_point_slots(ops, dtype, case):
    """stp_conv_params.stats_slots: the epilogue adds its tile sums atomically into int64 fixed-point slots (order-independent,
    so deterministic); stp_bn_apply_slots must equal stp_bn_stats + stp_bn_apply on the stored conv output, and
    stp_bn_backward_slots must equal conv + stp_bn_backward.  Two runs give bit-identical results."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, tile = case
    rng = np.random.RandomState(91)
    rows = n * h * w
    src, wt = q(rng.randn(n, h, w, ci), dtype), q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    gamma, beta = (rng.rand(co) + 0.5).astype(np.float32), (rng.randn(co) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    g, b = f(gamma), f(beta)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    sd = dev(src, dtype)
    mk = lambda dst: ops.conv_params(sd, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                     Cout=co, dtype=ops.dt(dst), tile=tile)
    nslots = 16 if co * 16 <= 1024 else max(1, 1024 // co)
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    outs = []
    for rep in range(2):
        y = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
        slots = torch.zeros(2 * co * nslots, dtype=torch.int64, device=DEV)
        P = mk(y)
        P.stats_partial, P.stats_slots = ops.ptr(slots), nslots
        ops.conv2d(P)
        m1, r1 = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
        mm1, mv1 = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)
        a1 = torch.empty_like(y)
        _lib.call("stp_bn_apply_slots", ops.ptr(y), ops.ptr(a1), ops.dt(y), rows, co, ops.ptr(slots), nslots, 1e-3, 0.99, ops.ptr(m1), ops.ptr(r1),
                  ops.ptr(mm1), ops.ptr(mv1), ops.ptr(g), ops.ptr(b), 1, ops.stream())
        outs.append((host(a1), host(m1), host(r1), host(mm1), host(mv1)))
    for u, v in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(u, v)                                  # atomics in integer arithmetic: run-to-run identical
    m0, r0 = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    mm0, mv0 = torch.zeros(co, device=DEV), torch.ones(co, device=DEV)
    ops.bn_stats(y, rows, co, 1e-3, 0.99, m0, r0, mm0, mv0, ws)
    a0 = torch.empty_like(y)
    ops.bn_apply(y, a0, rows, co, co, m0, r0, g, b, relu=1)
    np.testing.assert_allclose(outs[0][1], host(m0), atol=2e-6 * max(1.0, np.abs(host(m0)).max()))
    np.testing.assert_allclose(outs[0][2], host(r0), rtol=2e-5)
    np.testing.assert_allclose(outs[0][3], host(mm0), atol=1e-6)
    np.testing.assert_allclose(outs[0][4], host(mv0), rtol=2e-5)
    np.testing.assert_allclose(outs[0][0], host(a0), atol=tol(host(a0), dtype, 0.5))
    # backward sums through slots
    x = dev(q(rng.randn(n, h, w, co) * 1.5 + 0.3, dtype), dtype)
    ops.bn_stats(x, rows, co, 1e-3, 0.99, m0, r0, None, None, ws)
    dy = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
    ops.conv2d(mk(dy))
    dx0, dg0, db0 = torch.empty_like(dy), torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ops.bn_backward(x, dy, dx0, rows, co, m0, r0, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    gbuf = torch.empty_like(dy)
    slots = torch.zeros(2 * co * nslots, dtype=torch.int64, device=DEV)
    P = mk(gbuf)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(x), ops.ptr(m0), ops.ptr(r0), ops.ptr(g), ops.ptr(b), 1
    P.stats_partial, P.stats_slots = ops.ptr(slots), nslots
    ops.conv2d(P)
    dx1, dg1, db1 = torch.empty_like(dy), torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    _lib.call("stp_bn_backward_slots", ops.ptr(x), ops.ptr(gbuf), ops.ptr(dx1), ops.dt(x), rows, co, ops.ptr(m0), ops.ptr(r0), ops.ptr(g),
              ops.ptr(slots), nslots, ops.ptr(dg1), ops.ptr(db1), 0, ops.stream())
    sc = lambda a: 1e-3 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(host(db1), host(db0), atol=sc(host(db0)))
    np.testing.assert_allclose(host(dg1), host(dg0), atol=sc(host(dg0)))
    pre = host(x) * (host(r0) * gamma) + (beta - host(m0) * host(r0) * gamma)
    safe = np.abs(pre) > 1e-3
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 0.5))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 19, 45, 16, 16, 0, 1), (2, 20, 34, 16, 8, 1, 1), (1, 12, 40, 8, 16, 0, 0), (2, 16, 32, 32, 16, 1, 2),
                                  (2, 43, 139, 16, 16, 0, 1), (1, 21, 67, 32, 16, 1, 1), (1, 40, 136, 32, 32, 0, 2)])      # interior tiles
def test_producer_batchnorm_fused_into_small_channel_staging(ops, dtype, case):
    """stp_conv_params.src_bn_* / stp_wgrad_params.src_bn_*: the small-channel forward and weight-gradient kernels normalise the
    pre-BatchNormalization tensor while staging it.  Bit-identical to stp_bn_apply followed by the plain kernels (same fma,
    activation and rounding), with and without nearest-2x upsampling; the generic kernels refuse the fields."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, up, relu = case
    if dtype == "fp32" and ci > 16:
        pytest.skip("fp32 small-channel kernel: Cin <= 16")
    rng = np.random.RandomState(77)
    rows = n * h * w
    ypre = q(rng.randn(n, h, w, ci) * 2 + 0.5, dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    mean, rstd = f(ypre.reshape(-1, ci).mean(0)), f(1.0 / np.sqrt(ypre.reshape(-1, ci).var(0) + 1e-3))
    gamma, beta = f(rng.rand(ci) + 0.5), f(rng.randn(ci) * 0.3)
    yd = dev(ypre, dtype)
    act = torch.empty_like(yd)
    ops.bn_apply(yd, act, rows, ci, ci, mean, rstd, gamma, beta, relu=relu)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    H, W = (2 * h, 2 * w) if up else (h, w)
    mk = lambda src, dst: ops.conv_params(src, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=H, Wv=W, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=H, Wo=W,
                                          Cout=co, dtype=ops.dt(dst), mode=(ops.SRC_NEAREST2X if up else ops.SRC_DIRECT))
    ref, got = torch.empty((n, H, W, co), dtype=TD[dtype], device=DEV), torch.empty((n, H, W, co), dtype=TD[dtype], device=DEV)
    P0 = mk(act, ref)
    assert _lib.load().stp_conv2d_sc_eligible(P0)
    ops.conv2d(P0)
    P1 = mk(yd, got)
    P1.src_bn_mean, P1.src_bn_rstd, P1.src_bn_gamma, P1.src_bn_beta, P1.src_bn_relu = ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), relu
    ops.conv2d(P1)
    np.testing.assert_array_equal(host(got), host(ref))
    P1.tile = 5                                                            # a GEMM tile: the generic kernel must refuse
    assert _lib.load().stp_conv2d(P1, ops.stream()) == -1
    # weight gradient
    dy = dev(q(rng.randn(n, H, W, co), dtype), dtype)
    def wgrad(src, fused):
        wp = _lib.WgradParams()
        wp.src0, wp.dy = ops.ptr(src), ops.ptr(dy)
        wp.N, wp.Hs0, wp.Ws0, wp.Hv, wp.Wv, wp.C0, wp.C1 = n, h, w, H, W, ci, 0
        wp.src0_mode = ops.SRC_NEAREST2X if up else ops.SRC_DIRECT
        wp.KH, wp.KW, wp.stride, wp.pad, wp.Ho, wp.Wo, wp.Cout = 3, 3, 1, 1, H, W, co
        wp.accumulate, wp.dtype, wp.splits = 0, ops.dt(src), 0
        if fused:
            wp.src_bn_mean, wp.src_bn_rstd, wp.src_bn_gamma, wp.src_bn_beta, wp.src_bn_relu = ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), relu
        dw = torch.zeros(co * 9 * ci, dtype=torch.float32, device=DEV)
        wp.dw = ops.ptr(dw)
        nb = int(_lib.load().stp_conv2d_wgrad_workspace_bytes(wp))
        ws = torch.empty(max(nb, 4) // 4 + 4, dtype=torch.float32, device=DEV)
        rc = _lib.load().stp_conv2d_wgrad(wp, ops.ptr(ws), ws.numel() * 4, ops.stream())
        return rc, host(dw), wp, ws
    if _lib.load().stp_wgrad_sc_eligible(wgrad(act, False)[2]):
        rc0, d0, _, _ = wgrad(act, False)
        rc1, d1, wp1, ws1 = wgrad(yd, True)
        assert rc0 == 0 and rc1 == 0
        np.testing.assert_array_equal(d1, d0)
        wp1.splits = 2                                                     # forces the GEMM plan, which has no fused producer BN
        assert _lib.load().stp_conv2d_wgrad_partial(wp1, ops.ptr(ws1), ws1.numel() * 4, 1, ops.stream()) == -1


@pytest.mark.parametrize("case", [(2, 16, 32, 128, 128, 1), (1, 8, 64, 64, 72, 0), (2, 32, 16, 192, 64, 2), (1, 4, 128, 64, 136, 1)])
@pytest.mark.parametrize("dtype", H16)
def test_producer_batchnorm_fused_into_row_of_taps_weight_gradient(ops, case, dtype):
    """stp_wgrad_params.src_bn_* on the row-of-taps kernel (variant 4 / automatic): the halo tile of every pixel step is normalised
    in LDS by the thread that fetched it.  Bit-identical to stp_bn_apply followed by the same kernel on the normalised tensor
    (same fma, activation, rounding; padding stays zero); the pixel-reduction GEMM variants refuse the fields."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co, relu = case
    rng = np.random.RandomState(78)
    rows = n * h * w
    ypre = q(rng.randn(n, h, w, ci) * 2 + 0.5, dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    mean, rstd = f(ypre.reshape(-1, ci).mean(0)), f(1.0 / np.sqrt(ypre.reshape(-1, ci).var(0) + 1e-3))
    gamma, beta = f(rng.rand(ci) + 0.5), f(rng.randn(ci) * 0.3)
    yd = dev(ypre, dtype)
    act = torch.empty_like(yd)
    ops.bn_apply(yd, act, rows, ci, ci, mean, rstd, gamma, beta, relu=relu)
    dy = dev(q(rng.randn(n, h, w, co), dtype), dtype)

    def wgrad(src, fused, variant):
        dw = torch.full((co, 3, 3, ci), float("nan"), dtype=torch.float32, device=DEV)
        W = ops.wgrad_params(src, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(src))
        if fused:
            W.src_bn_mean, W.src_bn_rstd, W.src_bn_gamma, W.src_bn_beta, W.src_bn_relu = ops.ptr(mean), ops.ptr(rstd), ops.ptr(gamma), ops.ptr(beta), relu
        ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
        rc = _lib.load().stp_conv2d_wgrad_partial(ops.C.byref(W), ops.ptr(ws), ws.numel() * 4, variant, ops.stream())
        if rc == 0:
            ops.conv2d_wgrad_reduce(W, ws, variant)
        return rc, host(dw), W

    rc0, d0, W0 = wgrad(act, False, 4)
    assert rc0 == 0 and _lib.load().stp_conv2d_wgrad_kernel_id(ops.C.byref(W0)) in (2, 3)
    np.testing.assert_allclose(d0.transpose(1, 2, 3, 0), np_ops.conv2d_wgrad(host(act), host(dy), (3, 3), 1, 1), atol=tol(d0, dtype))
    for variant in (4, 0):
        rc1, d1, _ = wgrad(yd, True, variant)
        assert rc1 == 0
        np.testing.assert_array_equal(d1, d0)
    assert wgrad(yd, True, 2)[0] == -1                                      # the GEMM variants have no fused producer BN


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bn_backward"])
@pytest.mark.parametrize("shape", [(2, 18, 44), (1, 42, 138)])      # ragged 8x32 tiles; the second one has interior tiles (lean kernel)
def test_upsample_gradient_folded_into_small_channel_epilogue(ops, dtype, mode, shape):
    """stp_conv_params.dst_sum2x2: the data-gradient convolution of an UpSampling2D(2) input writes the 2x2 block sums
    at low resolution; must equal convolution at high resolution followed by stp_upsample2x_bwd (and, with bnb_x, by the
    unfused BatchNormalization backward)."""
    (n, h, w), ci, co = shape, 16, 32            # virtual (upsampled) size
    rng = np.random.RandomState(21)
    src = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    sd = dev(src, dtype)
    mk = lambda dst: ops.conv_params(sd, fwd, dst, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                     Cout=co, dtype=ops.dt(dst), tile=512)
    hi = torch.empty((n, h, w, co), dtype=TD[dtype], device=DEV)
    ops.conv2d(mk(hi))
    base = q(rng.randn(n, h // 2, w // 2, co), dtype)
    want = dev(base, dtype) if mode == "accumulate" else torch.empty((n, h // 2, w // 2, co), dtype=TD[dtype], device=DEV)
    ops.upsample2x_bwd(hi, want, n, h // 2, w // 2, co, co, accumulate=int(mode == "accumulate"))
    got = dev(base, dtype) if mode == "accumulate" else torch.full((n, h // 2, w // 2, co), float("nan"), dtype=TD[dtype], device=DEV)
    P = mk(got)
    P.dst_sum2x2 = 1
    P.accumulate0 = int(mode == "accumulate")
    if mode != "bn_backward":
        ops.conv2d(P)
        # bf16: the unfused path rounds the hi-res gradient to bf16 before summing, the fused one sums in fp32
        np.testing.assert_allclose(host(got), host(want), atol=tol(host(want), dtype, 1.0 if dtype != "fp32" else 0.05))
        return
    rows = n * (h // 2) * (w // 2)
    x = dev(q(rng.randn(n, h // 2, w // 2, co) + 0.2, dtype), dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    g, b = f(rng.rand(co) + 0.5), f(rng.randn(co) * 0.3)
    m, r = torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(co) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(x, rows, co, 1e-3, 0.99, m, r, None, None, ws)
    dx0, dg0, db0 = torch.empty_like(want), torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ops.bn_backward(x, want, dx0, rows, co, m, r, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(x), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    tiles = ops.conv2d_stats_floats(P) // (2 * co)
    dx1, dg1, db1 = torch.empty_like(want), torch.empty(co, device=DEV), torch.empty(co, device=DEV)
    ops.bn_backward_fused(x, got, dx1, rows, co, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    k = 1.0 if dtype != "fp32" else 0.05
    np.testing.assert_allclose(host(db1), host(db0), atol=(2e-2 if dtype != "fp32" else 1e-3) * np.abs(host(db0)).max() + 1e-3)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=(2e-2 if dtype != "fp32" else 1e-3) * np.abs(host(dg0)).max() + 1e-3)
    pre = host(x) * (host(r) * host(g)) + (host(b) - host(m) * host(r) * host(g))
    safe = np.abs(pre) > 1e-3
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, k))


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("mode", ["plain", "accumulate", "bn_backward"])
@pytest.mark.parametrize("c_up", [64, 32, 96])
def test_wide_output_data_gradient_of_upsample_concat(ops, dtype, mode, c_up):
    """stp_conv2d_scw: the data gradient of conv3x3(concat(UpSampling2D(2)(x), skip)) -> 32 channels in ONE launch - the first
    c_up of the 128 output channels are summed 2x2 into the low-resolution gradient (optionally with the fused
    BatchNormalization-backward mask + sums), the others are the skip gradient.  Must equal the generic two-destination kernel at
    high resolution followed by stp_upsample2x_bwd (and the unfused BatchNormalization backward)."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co = 2, 20, 72, 32, 128        # ragged 8 x 32 tiles in both directions; more tiles than one workgroup pass
    c_sk = co - c_up
    rng = np.random.RandomState(33)
    dyv = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    dy = dev(dyv, dtype)
    acc = int(mode == "accumulate")
    base_up, base_sk = q(rng.randn(n, h // 2, w // 2, c_up), dtype), q(rng.randn(n, h, w, c_sk), dtype)
    fresh = lambda a: dev(a, dtype) if acc else torch.full(a.shape, float("nan"), dtype=TD[dtype], device=DEV)
    mk = lambda d0, d1, tile: ops.conv_params(dy, fwd, d0, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w,
                                              Cout=co, dtype=ops.dt(d0), dst1=d1, Cd0=c_up, accumulate1=acc, tile=tile)
    # reference: generic kernel at high resolution, then the upsampling gradient
    hi, want_sk, want_up = torch.empty((n, h, w, c_up), dtype=TD[dtype], device=DEV), fresh(base_sk), fresh(base_up)
    R = mk(hi, want_sk, 0)
    assert not _lib.load().stp_conv2d_scw_eligible(ops.C.byref(R))
    ops.conv2d(R)
    ops.upsample2x_bwd(hi, want_up, n, h // 2, w // 2, c_up, c_up, accumulate=acc)
    got_sk, got_up = fresh(base_sk), fresh(base_up)
    P = mk(got_up, got_sk, 0)
    P.dst_sum2x2, P.accumulate0 = 1, acc
    assert _lib.load().stp_conv2d_scw_eligible(ops.C.byref(P)) and _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 640
    if mode != "bn_backward":
        ops.conv2d(P)
        np.testing.assert_allclose(host(got_sk), host(want_sk), atol=tol(host(want_sk), dtype, 1.0))
        # the unfused path rounds the hi-res gradient to 16 bits before summing, the fused one sums in fp32
        np.testing.assert_allclose(host(got_up), host(want_up), atol=tol(host(want_up), dtype, 1.0))
        return
    rows = n * (h // 2) * (w // 2)
    x = dev(q(rng.randn(n, h // 2, w // 2, c_up) + 0.2, dtype), dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    g, b = f(rng.rand(c_up) + 0.5), f(rng.randn(c_up) * 0.3)
    m, r = torch.empty(c_up, device=DEV), torch.empty(c_up, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(c_up) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(x, rows, c_up, 1e-3, 0.99, m, r, None, None, ws)
    dx0, dg0, db0 = torch.empty_like(want_up), torch.empty(c_up, device=DEV), torch.empty(c_up, device=DEV)
    ops.bn_backward(x, want_up, dx0, rows, c_up, m, r, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(x), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
    st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
    P.stats_partial = ops.ptr(st)
    ops.conv2d(P)
    tiles = ops.conv2d_stats_floats(P) // (2 * c_up)
    assert tiles == P.stats_tiles and np.isfinite(host(st)[:2 * c_up * tiles]).all()
    np.testing.assert_allclose(host(got_sk), host(want_sk), atol=tol(host(want_sk), dtype, 1.0))
    dx1, dg1, db1 = torch.empty_like(want_up), torch.empty(c_up, device=DEV), torch.empty(c_up, device=DEV)
    ops.bn_backward_fused(x, got_up, dx1, rows, c_up, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    np.testing.assert_allclose(host(db1), host(db0), atol=2e-2 * np.abs(host(db0)).max() + 1e-3)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=2e-2 * np.abs(host(dg0)).max() + 1e-3)
    pre = host(x) * (host(r) * host(g)) + (host(b) - host(m) * host(r) * host(g))
    safe = np.abs(pre) > 1e-3
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 1.0))
    # replay: fixed partition, fixed order
    st2 = torch.full_like(st, float("nan"))
    P.stats_partial = ops.ptr(st2)
    again = fresh(base_up)
    P.dst0 = ops.ptr(again)
    ops.conv2d(P)
    assert torch.equal(again, got_up) and torch.equal(st2[:2 * c_up * tiles], st[:2 * c_up * tiles])


PW_SHAPES = [(64, 64), (64, 256), (256, 64), (256, 128), (128, 256), (128, 512), (512, 128), (256, 256), (512, 256), (256, 512), (64, 512)]


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("mode", ["plain", "bias", "accumulate", "residual", "stats", "residual_stats", "bn_backward", "bn_backward_accumulate",
                                  "bn_backward_relu6", "bn_backward_linear"])
@pytest.mark.parametrize("shape", PW_SHAPES, ids=lambda s_: "%dto%d" % s_)
def test_pointwise_streaming_kernel(ops, dtype, mode, shape):
    """stp_conv2d_pw (conv_pw.hip, round 6): the 1x1 / stride-1 convolutions of the bottleneck ResNets (classification_models
    residual_bottleneck_block conv1 / conv3 / shortcut, FPN laterals: reference graph through segmentation.py:109-118) and their data
    gradients as a pixel-streaming kernel, against a float64 matrix product of the same 16-bit operands: every served (Cin, Cout) pair x
    every epilogue (bias | accumulate | residual | statistics (+ residual) | BatchNormalization backward (+ accumulate; ReLU / ReLU6 /
    linear)).  33792 pixels = more tiles than workgroups for every tile size (32 / 64 / 128 pixels), so the persistent loop, the double
    buffer and the cross-tile sums are exercised; the table of fused sums has one column per workgroup; a replay is bit-identical."""
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    cin, cout = shape
    n, h, w = 2, 96, 176
    P_ = n * h * w
    rng = np.random.RandomState(7 + cin + 3 * cout)
    x = q(rng.randn(n, h, w, cin), dtype)
    wt = q(rng.randn(1, 1, cin, cout) / np.sqrt(cin), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    f = lambda a_: keep(torch.from_numpy(np.ascontiguousarray(a_, dtype=np.float32)).to(DEV))
    bnb = mode.startswith("bn_backward")
    relu = {"bn_backward_relu6": 2, "bn_backward_linear": 0}.get(mode, 1)
    acc = mode in ("accumulate", "bn_backward_accumulate")
    res_np = q(rng.randn(n, h, w, cout), dtype) if mode in ("residual", "residual_stats") else None
    res = dev(res_np, dtype) if res_np is not None else None
    bias_np = (rng.randn(cout) * 0.5).astype(np.float32) if mode == "bias" else None
    bias = f(bias_np) if bias_np is not None else None
    prev_np = q(rng.randn(n, h, w, cout), dtype) if acc else None
    bx_np = q(rng.randn(n, h, w, cout) * 1.5 + 0.2, dtype) if bnb else None
    bx = dev(bx_np, dtype) if bnb else None
    gam, bet, mean, rstd = rng.rand(cout) + 0.5, rng.randn(cout) * 0.3, rng.randn(cout) * 0.1 + 0.2, rng.rand(cout) + 0.5
    g_, b_, m_, r_ = f(gam), f(bet), f(mean), f(rstd)

    def run(tile):
        y = dev(prev_np, dtype).clone() if acc else torch.full((n, h, w, cout), float("nan"), dtype=TD[dtype], device=DEV)
        keep(y)
        P = ops.conv_params(xd, fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=cin, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=cout, dtype=ops.dt(y),
                            residual=res, bias=bias, accumulate0=int(acc), tile=tile)
        st = None
        if bnb or mode in ("stats", "residual_stats"):
            if bnb:
                P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(bx), ops.ptr(m_), ops.ptr(r_), ops.ptr(g_), ops.ptr(b_), relu
            P.stats_partial = 1          # (non-NULL: the sizing query keys on it)
            nfl = ops.conv2d_stats_floats(P)
            st = keep(torch.full((max(4, nfl),), float("nan"), dtype=torch.float32, device=DEV))
            P.stats_partial = ops.ptr(st)
        return y, st, P

    y, st, P = run(0)
    served = bool(lib.stp_conv2d_pw_eligible(ops.C.byref(P)))
    level = 0 if (cin, cout) == (256, 512) else 1 if (cin, cout) in ((128, 512), (512, 128)) else 2
    assert served == (not bnb or level >= (2 if acc else 1)), "eligibility of %s / %s" % (shape, mode)
    assert (lib.stp_conv2d_tile_for(ops.C.byref(P)) == 800) == served
    if not served:
        P.tile = 800
        assert lib.stp_conv2d(ops.C.byref(P), ops.stream()) == -1          # STP_E_BADARG: never a silent fallback under a forced tile id
        return
    ops.conv2d(P)
    got = host(y).astype(np.float64)
    ref = (x.reshape(P_, cin).astype(np.float64) @ wt.reshape(cin, cout).astype(np.float64)).reshape(n, h, w, cout)
    if bias_np is not None:
        ref = ref + bias_np
    if res_np is not None:
        ref = ref + res_np
    if acc:
        ref = ref + prev_np
    cols = ops.conv2d_stats_floats(P) // (2 * cout) if st is not None else 0
    if bnb:
        sc = (rstd * gam).astype(np.float32)
        sh = (bet.astype(np.float32) - mean.astype(np.float32) * sc).astype(np.float32)
        tt = (bx_np.astype(np.float64) * sc.astype(np.float64) + sh.astype(np.float64)).astype(np.float32)        # the forward's fma, one rounding
        on = np.ones_like(tt, bool) if relu == 0 else (tt > 0) if relu == 1 else ((tt > 0) & (tt < 6))
        ref = np.where(on, ref, 0.0)
        bad = np.abs(got - ref) > tol(ref, dtype)
        assert bad.mean() < 1e-5, bad.mean()                        # (a mask decision within rounding of the activation's kink may differ)
        assert np.array_equal(got == 0, ~on | (got == 0))           # masked positions are exact zeros
        sums = host(st)[:2 * cout * cols].reshape(2, cout, cols).astype(np.float64).sum(-1)
        xhat = (bx_np.astype(np.float64) - mean.astype(np.float32).astype(np.float64)) * rstd.astype(np.float32).astype(np.float64)
        want0, want1 = got.reshape(P_, cout).sum(0), (got * xhat).reshape(P_, cout).sum(0)      # the sums take the values as STORED
        np.testing.assert_allclose(sums[0], want0, rtol=1e-4, atol=2e-4 * np.abs(got).sum(axis=(0, 1, 2)).max())
        np.testing.assert_allclose(sums[1], want1, rtol=1e-3, atol=1e-3 * np.abs(got * xhat).sum(axis=(0, 1, 2)).max())
    else:
        np.testing.assert_allclose(got, ref, atol=tol(ref, dtype))
        if st is not None:
            sums = host(st)[:2 * cout * cols].reshape(2, cout, cols).astype(np.float64).sum(-1)
            np.testing.assert_allclose(sums[0], got.reshape(P_, cout).sum(0), rtol=1e-4, atol=2e-4 * np.abs(got).sum(axis=(0, 1, 2)).max())
            np.testing.assert_allclose(sums[1], (got ** 2).reshape(P_, cout).sum(0), rtol=1e-4)
    if st is not None:
        assert P.stats_tiles == cols == lib.stp_conv2d_pw_cols(ops.C.byref(P)) and 0 < cols <= 512
    # replay: static tile assignment, fixed order of every sum
    y2, st2, P2 = run(0)
    ops.conv2d(P2)
    assert torch.equal(y2, y) and (st is None or torch.equal(st2[:2 * cout * cols], st[:2 * cout * cols]))
    # against the per-tap kernel on the same operands (what served these launches before): same values up to the summation order
    if not bnb:
        y3, st3, P3 = run(0)
        P3.tile = 65 if cin % 64 == 0 else 1
        if lib.stp_conv2d_tile_for(ops.C.byref(P3)) == P3.tile:
            if st3 is not None:
                st3 = keep(torch.full((max(4, ops.conv2d_stats_floats(P3)),), float("nan"), dtype=torch.float32, device=DEV))
                P3.stats_partial = ops.ptr(st3)
            ops.conv2d(P3)
            np.testing.assert_allclose(host(y3), host(y), atol=tol(ref, dtype))


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("mode", ["plain", "stats", "residual_stats", "bn_backward"])
def test_64_channel_kernel_with_weights_in_registers(ops, dtype, mode):
    """stp_conv2d_s64 (64 -> 64 channels, 3x3: ResNet stage 1 forward / data gradient): against a plain PyTorch fp32 convolution of
    the same 16-bit operands, and against the generic kernel for the fused statistics / BatchNormalization-backward sums.  Ragged
    8 x 32 tiles in both directions, three tiles per workgroup."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c = 3, 100, 440, 64
    rng = np.random.RandomState(45)
    x = q(rng.randn(n, h, w, c), dtype)
    wt = q(rng.randn(3, 3, c, c) / np.sqrt(9 * c), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    res = dev(q(rng.randn(n, h, w, c), dtype), dtype) if mode == "residual_stats" else None
    bx = dev(q(rng.randn(n, h, w, c) + 0.2, dtype), dtype) if mode == "bn_backward" else None
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    g, b, m, r = f(rng.rand(c) + 0.5), f(rng.randn(c) * 0.3), f(rng.randn(c) * 0.1 + 0.2), f(rng.rand(c) + 0.5)
    def run(tile):
        y = torch.full((n, h, w, c), float("nan"), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(xd, fwd, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=c, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=c, dtype=ops.dt(y),
                            residual=res, tile=tile)
        st = None
        if mode != "plain":
            if mode == "bn_backward":
                P.bnb_x, P.bnb_mean, P.bnb_rstd, P.bnb_gamma, P.bnb_beta, P.bnb_relu = ops.ptr(bx), ops.ptr(m), ops.ptr(r), ops.ptr(g), ops.ptr(b), 1
            st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
            P.stats_partial = ops.ptr(st)
        ops.conv2d(P)
        tiles = ops.conv2d_stats_floats(P) // (2 * c)
        sums = host(st)[:2 * c * tiles].reshape(2, c, tiles).astype(np.float64).sum(-1) if st is not None else None
        return y, st, sums, P
    y, st, sums, P = run(736)                                    # (opt-in kernel: by tile id)
    assert _lib.load().stp_conv2d_s64_eligible(ops.C.byref(P)) and _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 736
    yr, _, sums_r, Pr = run(2)                                   # the general implicit-GEMM kernel
    assert _lib.load().stp_conv2d_tile_for(ops.C.byref(Pr)) == 2
    if mode != "bn_backward":
        ref = torch.nn.functional.conv2d(xd.float().permute(0, 3, 1, 2), torch.from_numpy(wt.transpose(3, 2, 0, 1).astype(np.float32)).to(DEV), padding=1)
        ref = ref.permute(0, 2, 3, 1)
        if res is not None:
            ref = ref + res.float()
        np.testing.assert_allclose(host(y), host(ref), atol=tol(host(ref), dtype))
    else:
        # masked gradients: equal to the generic kernel's except where the two accumulate in a different order across a rounding boundary
        d = np.abs(host(y) - host(yr))
        assert (d > tol(host(yr), dtype)).mean() < 1e-4
    if sums is not None:
        np.testing.assert_allclose(sums, sums_r, rtol=2e-3, atol=2e-3 * np.abs(sums_r).max())
        assert P.stats_tiles == ops.conv2d_stats_floats(P) // (2 * c)
        y2, st2, _, _ = run(736)                              # replay: fixed partition, fixed order
        assert torch.equal(y2, y) and torch.equal(st2[:2 * c * P.stats_tiles], st[:2 * c * P.stats_tiles])


@pytest.mark.parametrize("dtype", H16)
@pytest.mark.parametrize("mode", ["plain", "stats", "bias_relu_accumulate"])
def test_narrow_output_forward_of_upsample_concat(ops, dtype, mode):
    """stp_conv2d_scn: Conv2D(32, 3x3)(Concatenate([UpSampling2D(2)(x), skip])) with 64 + 64 input channels - both halos resident in
    LDS, the upsampled one as low-resolution pixels.  Against numpy on the materialised concatenation, and against the generic
    kernel for the fused statistics; ragged 8 x 32 tiles, more tiles than workgroups' first pass."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c0, c1, co = 2, 36, 88, 64, 64, 32
    rng = np.random.RandomState(44)
    xlo = q(rng.randn(n, h // 2, w // 2, c0), dtype)
    skip = q(rng.randn(n, h, w, c1), dtype)
    wt = q(rng.randn(3, 3, c0 + c1, co) / np.sqrt(9 * (c0 + c1)), dtype)
    _, fwd, _, _ = prep_weights(ops, wt, dtype)
    xd, sd = dev(xlo, dtype), dev(skip, dtype)
    cat = np.concatenate([xlo.repeat(2, axis=1).repeat(2, axis=2), skip], axis=-1)
    want = np_ops.conv2d(cat, wt, 1, 1)
    base = q(rng.randn(n, h, w, co), dtype)
    bias = keep(torch.from_numpy(rng.randn(co).astype(np.float32)).to(DEV)) if mode == "bias_relu_accumulate" else None
    def run(tile):
        y = dev(base, dtype) if mode == "bias_relu_accumulate" else torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
        P = ops.conv_params(xd, fwd, y, N=n, Hs0=h // 2, Ws0=w // 2, Hv=h, Wv=w, C0=c0, C1=c1, src1=sd, mode=ops.SRC_NEAREST2X, KH=3, KW=3,
                            stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(y), bias=bias, relu=int(mode == "bias_relu_accumulate"),
                            accumulate0=int(mode == "bias_relu_accumulate"), tile=tile)
        st = None
        if mode == "stats":
            st = torch.full((max(4, ops.conv2d_stats_floats(P)),), float("nan"), dtype=torch.float32, device=DEV)
            P.stats_partial = ops.ptr(st)
        ops.conv2d(P)
        return y, st, P
    y, st, P = run(0)
    assert _lib.load().stp_conv2d_scn_eligible(ops.C.byref(P)) and _lib.load().stp_conv2d_tile_for(ops.C.byref(P)) == 704
    ref = want
    if mode == "bias_relu_accumulate":
        ref = np.maximum(want + host(bias) + base, 0.0)
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    if mode == "stats":
        tiles = ops.conv2d_stats_floats(P) // (2 * co)
        assert tiles == P.stats_tiles
        part = host(st)[:2 * co * tiles].reshape(2, co, tiles).astype(np.float64).sum(-1)
        yv = host(y).reshape(-1, co).astype(np.float64)
        np.testing.assert_allclose(part[0], yv.sum(0), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(part[1], (yv * yv).sum(0), rtol=1e-4, atol=1e-2)
        y2, st2, _ = run(0)                                   # replay: fixed partition, fixed order
        assert torch.equal(y2, y) and torch.equal(st2[:2 * co * tiles], st[:2 * co * tiles])


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("C,Cy", [(5, 8), (4, 8), (7, 8), (1, 4)])
def test_input_batchnorm_uint8_to_padded_channels(ops, dtype, C, Cy):
    """uint8 images of C channels -> BatchNormalization -> dtype tensor padded to 4 / 8 channels, padding = the constant 1."""
    rng = np.random.RandomState(21)
    n, h, w = 2, 9, 7
    rows = n * h * w
    x = rng.randint(0, 256, (n, h, w, C)).astype(np.uint8)
    beta = rng.randn(C).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd = keep(torch.from_numpy(x).to(DEV))
    mean, rstd = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(8) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, C, 2e-5, 0.99, mean, rstd, None, None, ws)
    xf = x.reshape(-1, C).astype(np.float64)
    np.testing.assert_allclose(host(mean), xf.mean(0), rtol=1e-5)
    np.testing.assert_allclose(host(rstd), 1.0 / np.sqrt(xf.var(0) + 2e-5), rtol=1e-5)
    y = torch.full((n, h, w, Cy), float("nan"), dtype=TD[dtype], device=DEV)
    ops.bn_apply(xd, y, rows, C, Cy, mean, rstd, None, f(beta), relu=0, pad_value=1.0)
    want = np.ones((rows, Cy))
    want[:, :C] = (xf - xf.mean(0)) / np.sqrt(xf.var(0) + 2e-5) + beta
    np.testing.assert_allclose(host(y).reshape(rows, Cy), want, atol=tol(want, dtype))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_input_batchnorm_uint8_to_padded4(ops, dtype):
    rng = np.random.RandomState(11)
    n, h, w = 2, 16, 18
    x = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    beta = np.array([0.1, -0.2, 0.05], np.float32)
    yref, mean, var = np_ops.bn_train(x.astype(np.float32), None, beta, 2e-5)
    xd = torch.from_numpy(x).to(DEV)
    m, r = torch.empty(3, device=DEV), torch.empty(3, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(4) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, n * h * w, 3, 2e-5, 0.99, m, r, None, None, ws)
    np.testing.assert_allclose(host(m), mean, rtol=1e-5)
    y = torch.empty((n, h, w, 4), dtype=TD[dtype], device=DEV)
    ops.bn_apply(xd, y, n * h * w, 3, 4, m, r, None, keep(torch.from_numpy(beta).to(DEV)), relu=0, pad_value=1.0)
    out = host(y)
    np.testing.assert_allclose(out[..., :3], yref, atol=tol(yref, dtype))
    np.testing.assert_array_equal(out[..., 3], 1.0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_maxpool_and_upsample_gradients(ops, dtype):
    rng = np.random.RandomState(12)
    n, h, w, c = 2, 14, 12, 32
    x = q(np.maximum(rng.randn(n, h, w, c), 0), dtype)
    ref = np_ops.maxpool3x3s2(x)
    ho, wo = ref.shape[1:3]
    xd = dev(x, dtype)
    y = torch.empty((n, ho, wo, c), dtype=TD[dtype], device=DEV)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=DEV)
    ops.maxpool3x3s2(xd, y, idx, n, h, w, c)
    np.testing.assert_array_equal(host(y), ref)
    dy = q(rng.randn(n, ho, wo, c), dtype)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
    out = torch.nn.functional.max_pool2d(torch.nn.functional.pad(xt, (1, 1, 1, 1)), 3, 2)
    out.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    dxref = xt.grad.permute(0, 2, 3, 1).numpy()
    dx = torch.empty_like(xd)
    ops.maxpool3x3s2_bwd(idx, dev(dy, dtype), dx, n, h, w, c)
    # ties (zeros after ReLU) may route to a different tap; compare only where x > 0 (gradient
    # through ReLU is zero elsewhere, which is what the network sees)
    msk = x > 0
    np.testing.assert_allclose(host(dx)[msk], dxref[msk], atol=tol(dxref, dtype, 2))
    # upsample gradient
    g = q(rng.randn(n, 2 * h, 2 * w, c), dtype)
    dxu = torch.empty((n, h, w, c), dtype=TD[dtype], device=DEV)
    ops.upsample2x_bwd(dev(g, dtype), dxu, n, h, w, c, c)
    np.testing.assert_allclose(host(dxu), np_ops.upsample2x_bwd(g), atol=tol(g, dtype, 4))


@pytest.mark.parametrize("case", [(2, 16, 24, 64, 1, True), (1, 64, 70, 64, 1, True), (3, 6, 4, 16, 2, False), (2, 128, 256, 8, 0, True)])
@pytest.mark.parametrize("dtype", H16)
def test_batchnorm_apply_fused_with_the_max_pooling_behind_it(ops, case, dtype):
    """stp_bn_apply_maxpool3x3s2 (round 5: the stem's bn0 -> relu0 -> pooling0 as one launch): the normalised tensor, the pooled tensor
    and the argmax indexes are BIT-identical to stp_bn_apply followed by stp_maxpool3x3s2."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c, relu, with_gamma = case
    rng = np.random.RandomState(hash(case) % 2**31)
    x = q(rng.randn(n, h, w, c) * 2.0 + 0.5, dtype)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd = dev(x, dtype)
    m, r = f(rng.randn(c) * 0.3), f(rng.rand(c) + 0.5)
    g, b = (f(rng.rand(c) + 0.5) if with_gamma else None), f(rng.randn(c) * 0.3)
    ho, wo = h // 2, w // 2
    act0, act1 = torch.empty_like(xd), torch.full_like(xd, float("nan"))
    y0, y1 = (torch.full((n, ho, wo, c), float("nan"), dtype=TD[dtype], device=DEV) for _ in range(2))
    i0, i1 = (torch.full((n, ho, wo, c), 255, dtype=torch.uint8, device=DEV) for _ in range(2))
    ops.bn_apply(xd, act0, n * h * w, c, c, m, r, g, b, relu=relu)
    ops.maxpool3x3s2(act0, y0, i0, n, h, w, c)
    _lib.call("stp_bn_apply_maxpool3x3s2", ops.ptr(xd), ops.ptr(act1), ops.ptr(y1), ops.ptr(i1), n, h, w, c, ops.dt(xd), ops.ptr(m), ops.ptr(r),
              ops.ptr(g) if g is not None else None, ops.ptr(b), relu, ops.stream())
    torch.cuda.synchronize()
    assert torch.equal(act1.view(torch.int16), act0.view(torch.int16))
    assert torch.equal(y1.view(torch.int16), y0.view(torch.int16)) and torch.equal(i1, i0)
    with pytest.raises(_lib.StpError):      # odd maps: the 2 x 2 blocks would not cover the tensor
        _lib.call("stp_bn_apply_maxpool3x3s2", ops.ptr(xd), ops.ptr(act1), ops.ptr(y1), ops.ptr(i1), n, h - 1, w, c, ops.dt(xd), ops.ptr(m), ops.ptr(r),
                  None, ops.ptr(b), relu, ops.stream())


@pytest.mark.parametrize("case", [(2, 12, 24, 24, 24, 0, 8), (1, 16, 64, 8, 8, 0, 4), (2, 6, 10, 8, 24, 8, 2), (1, 9, 33, 16, 16, 0, 8),
                                  (2, 5, 7, 16, 48, 16, 1), (1, 8, 8, 128, 320, 64, 1),       # factor 1 = a channel slice copied / added back
                                  # tiny maps, large factors, a slice of a wide tensor (PSPNet's pyramid levels): the separable two-pass form
                                  (2, 1, 1, 128, 320, 64, 32), (1, 2, 2, 512, 1024, 512, 16), (2, 3, 3, 64, 192, 128, 24), (1, 6, 6, 256, 256, 0, 8),
                                  # wide slices of a concatenation (FPN's pyramid levels: round 6, resize_bilinear_bwd_tile_kernel): two channel
                                  # chunks, ragged column segments, an odd row count (a row pair with one row), factors 2 .. 16 and a factor of 3
                                  (1, 18, 20, 128, 512, 128, 8), (2, 17, 33, 64, 192, 64, 4), (1, 40, 36, 128, 512, 384, 2), (1, 17, 3, 32, 32, 0, 16),
                                  (1, 20, 20, 48, 48, 0, 3)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_bilinear_resize_gradient_of_class_logits_and_of_an_unresized_slice(ops, case, dtype):
    """Round 5: resize_bilinear_bwd_rows_kernel - the gradient of the x4 / x8 bilinear resize of class logits (channel count padded to
    the 16-byte group: PSPNet's 20 classes in rows of 24, FPN's 3 in rows of 8), output rows streamed whole through LDS - and
    slice_copy_kernel (factor 1: the concatenation level that keeps its resolution).  Against torch autograd of the oracle's TF-1.x
    resize; channel slices (ldo / coff), accumulate, edge rows and columns, replay bit-identical."""
    from oracle import nets as onets
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c, ldo, coff, f = case
    rng = np.random.RandomState(hash(case) % 2**31)
    xt = torch.zeros(n, c, h, w, requires_grad=True)
    gy = q(rng.randn(n, h * f, w * f, ldo), dtype)
    onets.resize_bilinear_tf1(xt, f).backward(torch.from_numpy(gy[..., coff:coff + c]).permute(0, 3, 1, 2))
    ref = xt.grad.permute(0, 2, 3, 1).numpy()
    gd = dev(gy, dtype)
    base = q(rng.randn(n, h, w, c), dtype)
    outs = []
    for acc in (0, 1, 1):
        dx = dev(base, dtype) if acc else torch.full((n, h, w, c), float("nan"), dtype=TD[dtype], device=DEV)
        wsb = int(_lib.load().stp_resize_bilinear_bwd_workspace_bytes(n, h, w, c, f))
        ws = keep(torch.empty(max(wsb // 4, 1), dtype=torch.float32, device=DEV))
        _lib.call("stp_resize_bilinear_bwd", ops.ptr(gd), ops.ptr(dx), n, h, w, c, f, ldo, coff, ops.dt(dx), acc, ops.ptr(ws) if wsb else None, wsb, ops.stream())
        want = ref + (base if acc else 0.0)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype, 1.0))
        outs.append(host(dx).copy())
    assert np.array_equal(outs[1], outs[2])


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("factor", [1, 2, 4, 8])
def test_tf1_bilinear_resize_into_channel_slice_and_gradient(ops, dtype, factor):
    """stp_resize_bilinear(_bwd): TF 1.x resize_bilinear(align_corners=False) by an integer factor, written into a channel
    slice of a wider tensor (FPN's Concatenate), against the oracle's index formulation and torch autograd."""
    from oracle import nets as onets
    from segmentation_training_pipeline_amd import _lib
    rng = np.random.RandomState(41)
    n, h, w, c, ldo, coff = 2, 5, 7, 8, 24, 8
    x = q(rng.randn(n, h, w, c), dtype)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = onets.resize_bilinear_tf1(xt, factor)
    gy = q(rng.randn(n, h * factor, w * factor, ldo), dtype)
    ref.backward(torch.from_numpy(gy[..., coff:coff + c]).permute(0, 3, 1, 2))
    y = torch.full((n, h * factor, w * factor, ldo), 7.0, dtype=TD[dtype], device=DEV)
    xd = dev(x, dtype)
    _lib.call("stp_resize_bilinear", ops.ptr(xd), ops.ptr(y), n, h, w, c, factor, ldo, coff, ops.dt(xd), ops.stream())
    got = host(y)
    np.testing.assert_allclose(got[..., coff:coff + c], ref.detach().permute(0, 2, 3, 1).numpy(), atol=tol(x, dtype, 0.5))
    assert np.all(got[..., :coff] == 7.0) and np.all(got[..., coff + c:] == 7.0)          # neighbours' slices untouched
    for acc in (0, 1):
        base = q(rng.randn(n, h, w, c), dtype)
        dx = dev(base, dtype)
        _lib.call("stp_resize_bilinear_bwd", ops.ptr(dev(gy, dtype)), ops.ptr(dx), n, h, w, c, factor, ldo, coff, ops.dt(dx), acc, None, 0, ops.stream())
        want = xt.grad.permute(0, 2, 3, 1).numpy() + (base if acc else 0)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype, 1.0))
    # PSPNet pyramid pooling: AveragePooling2D(k, k) and its gradient
    xp = q(rng.randn(n, 6, 6, 8), dtype)
    xpt = torch.from_numpy(xp).permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = torch.nn.functional.avg_pool2d(xpt, 3, 3)
    gyp = q(rng.randn(n, 2, 2, 8), dtype)
    pr.backward(torch.from_numpy(gyp).permute(0, 3, 1, 2))
    yp, dxp = torch.empty((n, 2, 2, 8), dtype=TD[dtype], device=DEV), torch.empty((n, 6, 6, 8), dtype=TD[dtype], device=DEV)
    xpd = dev(xp, dtype)
    _lib.call("stp_avgpool", ops.ptr(xpd), ops.ptr(yp), n, 6, 6, 8, 3, ops.dt(xpd), None, 0, ops.stream())
    _lib.call("stp_avgpool_bwd", ops.ptr(dev(gyp, dtype)), ops.ptr(dxp), n, 6, 6, 8, 3, ops.dt(xpd), 0, ops.stream())
    np.testing.assert_allclose(host(yp), pr.detach().permute(0, 2, 3, 1).numpy(), atol=tol(xp, dtype, 0.5))
    np.testing.assert_allclose(host(dxp), xpt.grad.permute(0, 2, 3, 1).numpy(), atol=tol(gyp, dtype, 0.5))
    # FPN top-down add: x += nearest-2x(m)
    xa, ma = q(rng.randn(n, 6, 8, 16), dtype), q(rng.randn(n, 3, 4, 16), dtype)
    xd2 = dev(xa, dtype)
    _lib.call("stp_upsample2x_add", ops.ptr(xd2), ops.ptr(dev(ma, dtype)), n, 6, 8, 16, ops.dt(xd2), ops.stream())
    np.testing.assert_allclose(host(xd2), xa + np_ops.upsample2x(ma), atol=tol(xa, dtype, 1.0))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 14, 12, 64, 1), (1, 9, 21, 64, 0), (2, 8, 8, 32, 1)])
def test_maxpool_gradient_with_fused_batchnorm_backward_sums(ops, dtype, case):
    """stp_maxpool3x3s2_bwd_bn (bn0 of the stem: its gradient = the pool gradient on top of a decoder skip's): must equal
    stp_maxpool3x3s2_bwd + stp_bn_backward."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c, acc = case
    rng = np.random.RandomState(6)
    rows = n * h * w
    x = q(rng.randn(n, h, w, c) * 1.5 + 0.3, dtype)                    # BN input
    gamma, beta = (rng.rand(c) + 0.5).astype(np.float32), (rng.randn(c) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(c) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, c, 1e-3, 0.99, m, r, None, None, ws)
    act = torch.empty_like(xd)
    ops.bn_apply(xd, act, rows, c, c, m, r, g, b, relu=1)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = torch.empty((n, ho, wo, c), dtype=TD[dtype], device=DEV)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=DEV)
    ops.maxpool3x3s2(act, y, idx, n, h, w, c)
    dyp = dev(q(rng.randn(n, ho, wo, c), dtype), dtype)
    prior = q(rng.randn(n, h, w, c), dtype)
    dy0 = dev(prior, dtype)
    _lib.call("stp_maxpool3x3s2_bwd", ops.ptr(idx), ops.ptr(dyp), ops.ptr(dy0), n, h, w, c, ops.dt(xd), acc, ops.stream())
    dx0, dg0, db0 = torch.empty_like(dy0), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward(xd, dy0, dx0, rows, c, m, r, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    tiles = int(_lib.load().stp_maxpool3x3s2_bwd_bn_tiles(n, h, w, c, ops.dt(xd)))
    assert tiles > 0
    st = torch.full((2 * c * tiles,), float("nan"), dtype=torch.float32, device=DEV)
    gbuf = dev(prior, dtype)
    _lib.call("stp_maxpool3x3s2_bwd_bn", ops.ptr(idx), ops.ptr(dyp), ops.ptr(gbuf), n, h, w, c, ops.dt(xd), acc, ops.ptr(xd), ops.ptr(m),
              ops.ptr(r), ops.ptr(g), ops.ptr(b), 1, ops.ptr(st), ops.stream())
    assert not np.isnan(host(st)).any()
    dx1, dg1, db1 = torch.empty_like(dy0), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward_fused(xd, gbuf, dx1, rows, c, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    np.testing.assert_array_equal(host(gbuf)[safe], (host(dy0) * (pre > 0))[safe])
    scale = lambda a: 2e-4 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(host(db1), host(db0), atol=scale(host(db0)) * 5)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=scale(host(dg0)) * 5)
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 0.5))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 12, 16, 128, 3), (1, 9, 7, 256, 20), (2, 6, 6, 64, 1)])
def test_class_head_as_tap_channels_plus_tap_sum(ops, dtype, case):
    """Conv2D(classes, 3x3, padding 1, bias) over a wide feature map as a 1x1 stp_conv2d into 9 x classes tap channels (the 3x3 kernel's own
    bytes read as [9 classes][Cin]) + stp_tapsum_fwd (round 5): against the float64 oracle and the direct 3x3 launch; stp_tapsum_bwd
    against numpy (the adjoint: <tapsum(z), dy> = <z, tapsum_bwd(dy)> on the values it produced)."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co = case
    rng = np.random.RandomState(12)
    x = q(rng.randn(n, h, w, ci), dtype)
    wt = q(rng.randn(3, 3, ci, co) / np.sqrt(9 * ci), dtype)
    bias = (rng.randn(co) * 0.2).astype(np.float32)
    ref = np_ops.conv2d(x, wt, 1, 1) + bias
    # the 3x3 master [co][3][3][ci] read as a 1x1 kernel with 9 co outputs: HWIO view (1, 1, ci, 9 co) with column o * 9 + t
    w11 = np.ascontiguousarray(wt.transpose(3, 0, 1, 2).reshape(co * 9, ci).T.reshape(1, 1, ci, co * 9))
    _, f11, _, _ = prep_weights(ops, w11, dtype)
    xd = dev(x, dtype)
    zc = co * 9
    z = torch.full((n, h, w, zc), float("nan"), dtype=TD[dtype], device=DEV)
    ops.conv2d(ops.conv_params(xd, f11, z, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=zc, dtype=ops.dt(z)))
    y = torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV)
    bd = keep(torch.from_numpy(bias).to(DEV))
    _lib.call("stp_tapsum_fwd", ops.ptr(z), ops.ptr(y), ops.ptr(bd), n, h, w, co, zc, co, ops.dt(z), ops.stream())
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype, 2.0))
    zh = host(z)
    want = np.tile(bias, (n, h, w, 1)).astype(np.float64)
    zp = np.pad(zh.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
    for o in range(co):
        for t in range(9):
            want[..., o] += zp[:, t // 3:t // 3 + h, t % 3:t % 3 + w, o * 9 + t]
    np.testing.assert_allclose(host(y), want, atol=tol(want, dtype, 0.6))
    # backward: dz of the padded gradient layout [.., rup(co, vec)] -> [.., rup(9 co, vec)]
    vec = 4 if dtype == "fp32" else 8
    cdy, cdz = -(-co // vec) * vec, -(-zc // vec) * vec
    dy = np.zeros((n, h, w, cdy), np.float32)
    dy[..., :co] = q(rng.randn(n, h, w, co), dtype)
    dyd = dev(dy, dtype)
    dz = torch.full((n, h, w, cdz), float("nan"), dtype=TD[dtype], device=DEV)
    _lib.call("stp_tapsum_bwd", ops.ptr(dyd), ops.ptr(dz), n, h, w, co, cdy, cdz, ops.dt(dz), ops.stream())
    dzh = host(dz)
    dyp = np.pad(dy, ((0, 0), (1, 1), (1, 1), (0, 0)))
    for o in range(co):
        for t in range(9):
            np.testing.assert_array_equal(dzh[..., o * 9 + t], dyp[:, 2 - t // 3:2 - t // 3 + h, 2 - t % 3:2 - t % 3 + w, o])
    assert not dzh[..., zc:].any()
    lhs = float((want - bias) .reshape(-1, co).astype(np.float64).ravel() @ dy[..., :co].astype(np.float64).ravel())
    rhs = float(zh.astype(np.float64).ravel() @ dzh[..., :zc].astype(np.float64).ravel())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))


@pytest.mark.parametrize("dtype", H16)
def test_tap_channel_class_head_when_the_taps_cancel(ops, dtype):
    """Advisor finding (round 5): the tap-channel form of the class convolution stores nine PARTIAL sums in 16 bits before they are added, the
    direct 3x3 launch accumulates all taps in fp32 - the two differ most when the taps cancel.  Input: a constant map and a kernel whose
    nine taps sum to zero per (class, channel) with taps ~60x larger than the interior result; the tap form must stay within the bound its
    rounding points give - nine roundings of partials of size |z| (<= 9 x half a storage ulp of max |z|) plus the output rounding - and the
    oracle's restatement of it (oracle.nets._class_head: the storage-quantised oracle models these roundings) must agree with the device
    to one output rounding."""
    from segmentation_training_pipeline_amd import _lib
    from oracle import nets as onets
    n, h, w, ci, co = 1, 12, 16, 128, 3
    rng = np.random.RandomState(3)
    x = q(1.0 + 0.01 * rng.randn(n, h, w, ci), dtype)                  # nearly constant: the interior response is the SUM of the taps
    base = rng.randn(3, 3, ci, co)
    wt = q((base - base.mean(axis=(0, 1), keepdims=True)) * 0.5, dtype)         # taps cancel per (channel, class)
    bias = np.zeros(co, np.float32)
    exact = np_ops.conv2d(x, wt, 1, 1)
    w11 = np.ascontiguousarray(wt.transpose(3, 0, 1, 2).reshape(co * 9, ci).T.reshape(1, 1, ci, co * 9))
    _, f11, _, _ = prep_weights(ops, w11, dtype)
    _, f33, _, _ = prep_weights(ops, wt, dtype)
    xd = dev(x, dtype)
    zc = co * 9
    z = keep(torch.full((n, h, w, zc), float("nan"), dtype=TD[dtype], device=DEV))
    ops.conv2d(ops.conv_params(xd, f11, z, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=1, KW=1, stride=1, pad=0, Ho=h, Wo=w, Cout=zc, dtype=ops.dt(z)))
    y = keep(torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV))
    bd = keep(torch.from_numpy(bias).to(DEV))
    _lib.call("stp_tapsum_fwd", ops.ptr(z), ops.ptr(y), ops.ptr(bd), n, h, w, co, zc, co, ops.dt(z), ops.stream())
    yd = keep(torch.full((n, h, w, co), float("nan"), dtype=TD[dtype], device=DEV))
    ops.conv2d(ops.conv_params(xd, f33, yd, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.dt(yd)))
    zmax = float(np.abs(host(z)).max())
    half_ulp = 2.0 ** (np.floor(np.log2(zmax)) - (8 if dtype == "bf16" else 11))
    inner = (slice(None), slice(1, -1), slice(1, -1))
    assert np.abs(exact[inner]).max() < zmax / 20.0                     # the case is what it claims: partials >> result
    err_taps, err_direct = np.abs(host(y) - exact), np.abs(host(yd) - exact)
    print("taps cancel [%s]: max |partial| %.3g, interior |result| max %.3g; error tap form %.3g (bound %.3g), direct 3x3 %.3g"
          % (dtype, zmax, np.abs(exact[inner]).max(), err_taps.max(), 9 * half_ulp + tol(exact, dtype), err_direct.max()))
    assert err_taps.max() <= 9 * half_ulp + tol(exact, dtype)
    np.testing.assert_allclose(host(yd), exact, atol=tol(exact, dtype))
    # the oracle's tap-channel restatement (what the storage-quantised whole-step oracle uses for this layer)
    P = {"final_conv/kernel": torch.from_numpy(wt.astype(np.float32)), "final_conv/bias": torch.from_numpy(bias)}
    ctx = onets._Ctx(P, True, None, storage=TD[dtype])
    got_o = onets._class_head(ctx, torch.from_numpy(x.astype(np.float32)).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(host(y), got_o, atol=tol(exact, dtype) + 2 * half_ulp)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 8, 12, 64, 0), (1, 9, 7, 256, 1), (2, 6, 6, 32, 1), (1, 16, 16, 512, 0), (3, 5, 8, 20, 1)])
def test_scatter2x_gradient_of_a_1x1_stride_2_convolution(ops, dtype, case):
    """stp_scatter2x_bwd (round 5): t[n, a, b] lands at (2a, 2b) of dx - zeros elsewhere, or added to what dx holds; with a 1x1 / stride-1
    stp_conv2d in front it is the data gradient of a 1x1 / stride-2 convolution: checked against the zero-inserted launch it replaces
    and against numpy.  The _bn form (the launch that completes the gradient of a BatchNormalization + ReLU output): must equal
    stp_scatter2x_bwd + stp_bn_backward."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c, acc = case
    rng = np.random.RandomState(8)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    t = q(rng.randn(n, ho, wo, c), dtype)
    prior = q(rng.randn(n, h, w, c), dtype)
    td, dx = dev(t, dtype), dev(prior, dtype)
    _lib.call("stp_scatter2x_bwd", ops.ptr(td), ops.ptr(dx), n, h, w, c, ops.dt(td), acc, ops.stream())
    ref = prior.copy() if acc else np.zeros_like(prior)
    ref[:, ::2, ::2, :] += t
    np.testing.assert_allclose(host(dx), ref, atol=tol(ref, dtype, 0.5))
    if not acc:
        assert np.array_equal(host(dx)[:, 1::2], np.zeros_like(prior)[:, 1::2]) and np.array_equal(host(dx)[:, ::2, ::2], t)
    tiles = int(_lib.load().stp_scatter2x_bwd_bn_tiles(n, h, w, c, ops.dt(td)))
    v = 8 if (dtype != "fp32" and c % 8 == 0) else 4
    if 256 % (c // v):
        assert tiles == 0
        return
    assert tiles > 0
    rows = n * h * w
    x = q(rng.randn(n, h, w, c) * 1.5 + 0.3, dtype)                    # BN input
    gamma, beta = (rng.rand(c) + 0.5).astype(np.float32), (rng.randn(c) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(c) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, c, 1e-3, 0.99, m, r, None, None, ws)
    dx0, dg0, db0 = torch.empty_like(dx), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward(xd, dx, dx0, rows, c, m, r, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    st = torch.full((2 * c * tiles,), float("nan"), dtype=torch.float32, device=DEV)
    gbuf = dev(prior, dtype)
    _lib.call("stp_scatter2x_bwd_bn", ops.ptr(td), ops.ptr(gbuf), n, h, w, c, ops.dt(td), acc, ops.ptr(xd), ops.ptr(m), ops.ptr(r), ops.ptr(g),
              ops.ptr(b), 1, ops.ptr(st), ops.stream())
    assert not np.isnan(host(st)).any()
    dx1, dg1, db1 = torch.empty_like(dx), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward_fused(xd, gbuf, dx1, rows, c, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    np.testing.assert_array_equal(host(gbuf)[safe], (host(dx) * (pre > 0))[safe])
    scale = lambda a: 2e-4 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(host(db1), host(db0), atol=scale(host(db0)) * 5)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=scale(host(dg0)) * 5)
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 0.5))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_1x1_stride_2_data_gradient_low_resolution_gemm_plus_scatter_equals_the_zero_inserted_launch(ops, dtype):
    """The two forms of the data gradient of Conv2D(1x1, strides=2) on the same buffers: the zero-inserted stp_conv2d over the high-resolution
    grid (what graph.py issued before round 5) and 1x1 / stride-1 GEMM at low resolution + stp_scatter2x_bwd."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, ci, co = 2, 16, 24, 64, 128
    rng = np.random.RandomState(9)
    ho, wo = h // 2, w // 2
    dy = q(rng.randn(n, ho, wo, co), dtype)
    wt = q(rng.randn(1, 1, ci, co) / np.sqrt(ci), dtype)
    _, _, bwd, _ = prep_weights(ops, wt, dtype)
    dyd = dev(dy, dtype)
    d0 = torch.full((n, h, w, ci), float("nan"), dtype=TD[dtype], device=DEV)
    ops.conv2d(ops.conv_params(dyd, bwd, d0, N=n, Hs0=ho, Ws0=wo, Hv=2 * ho - 1, Wv=2 * wo - 1, C0=co, mode=ops.SRC_ZEROINS2X, KH=1, KW=1, stride=1,
                               pad=0, Ho=h, Wo=w, Cout=ci, dtype=ops.dt(d0)))
    tl = torch.empty((n, ho, wo, ci), dtype=TD[dtype], device=DEV)
    ops.conv2d(ops.conv_params(dyd, bwd, tl, N=n, Hs0=ho, Ws0=wo, Hv=ho, Wv=wo, C0=co, KH=1, KW=1, stride=1, pad=0, Ho=ho, Wo=wo, Cout=ci,
                               dtype=ops.dt(tl)))
    d1 = torch.full((n, h, w, ci), float("nan"), dtype=TD[dtype], device=DEV)
    _lib.call("stp_scatter2x_bwd", ops.ptr(tl), ops.ptr(d1), n, h, w, ci, ops.dt(tl), 0, ops.stream())
    ref = np.zeros((n, h, w, ci), np.float32)
    ref[:, ::2, ::2, :] = np.einsum("nhwo,io->nhwi", dy.astype(np.float64), wt[0, 0].astype(np.float64))
    np.testing.assert_allclose(host(d1), ref, atol=tol(ref, dtype))
    np.testing.assert_allclose(host(d1), host(d0), atol=tol(ref, dtype, 0.5))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 6, 10, 64, 0), (2, 5, 7, 512, 1), (1, 8, 16, 128, 1), (2, 4, 4, 32, 0), (1, 3, 5, 256, 1)])
def test_upsample_gradient_with_fused_batchnorm_backward_sums(ops, dtype, case):
    """stp_upsample2x_bwd_bn: the 2x2 fold that completes the gradient of a BatchNormalization(+ReLU) output (optionally on top of
    what other consumers accumulated) masks it and reduces the backward sums in the same pass; followed by
    stp_bn_backward_fused it must equal stp_upsample2x_bwd + stp_bn_backward."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, c, acc = case
    rng = np.random.RandomState(5)
    rows = n * h * w
    hi = dev(q(rng.randn(n, 2 * h, 2 * w, c), dtype), dtype)
    x = q(rng.randn(n, h, w, c) * 1.5 + 0.3, dtype)
    prior = q(rng.randn(n, h, w, c), dtype)
    gamma, beta = (rng.rand(c) + 0.5).astype(np.float32), (rng.randn(c) * 0.3).astype(np.float32)
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    xd, g, b = dev(x, dtype), f(gamma), f(beta)
    m, r = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ws = torch.empty(ops.bn_workspace_bytes(c) // 4, dtype=torch.float32, device=DEV)
    ops.bn_stats(xd, rows, c, 1e-3, 0.99, m, r, None, None, ws)
    # reference: plain fold, then the two-pass backward
    dy0 = dev(prior, dtype)
    ops.upsample2x_bwd(hi, dy0, n, h, w, c, c, accumulate=acc)
    dx0, dg0, db0 = torch.empty_like(dy0), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward(xd, dy0, dx0, rows, c, m, r, g, b, dg0, db0, relu=1, accumulate_dx=0, workspace=ws)
    # fused
    tiles = int(_lib.load().stp_upsample2x_bwd_bn_tiles(n, h, w, c, c, ops.dt(xd)))
    assert tiles > 0
    st = torch.full((2 * c * tiles,), float("nan"), dtype=torch.float32, device=DEV)
    gbuf = dev(prior, dtype)
    _lib.call("stp_upsample2x_bwd_bn", ops.ptr(hi), ops.ptr(gbuf), n, h, w, c, c, ops.dt(xd), acc, ops.ptr(xd), ops.ptr(m), ops.ptr(r),
              ops.ptr(g), ops.ptr(b), 1, ops.ptr(st), ops.stream())
    assert not np.isnan(host(st)).any()
    dx1, dg1, db1 = torch.empty_like(dy0), torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    ops.bn_backward_fused(xd, gbuf, dx1, rows, c, m, r, g, st, tiles, dg1, db1, accumulate_dx=0, workspace=ws)
    pre = host(xd) * (host(r) * gamma) + (beta - host(m) * host(r) * gamma)
    safe = np.abs(pre) > 1e-3
    np.testing.assert_array_equal(host(gbuf)[safe], (host(dy0) * (pre > 0))[safe])       # dY under the ReLU mask, bit for bit
    scale = lambda a: 2e-4 * np.abs(a).max() + 1e-4
    np.testing.assert_allclose(host(db1), host(db0), atol=scale(host(db0)) * 5)
    np.testing.assert_allclose(host(dg1), host(dg0), atol=scale(host(dg0)) * 5)
    np.testing.assert_allclose(host(dx1)[safe], host(dx0)[safe], atol=tol(host(dx0), dtype, 0.5))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 24, 64, (1, 2, 3, 6)), (1, 12, 40, (1, 2, 3, 6)), (3, 8, 16, (2, 4)), (2, 6, 8, (3,))])
def test_sum_of_resized_pyramid_terms_and_column_ranges_of_a_shared_kernel(ops, dtype, case):
    """The two pieces of the PSPNet head without its concatenation (segmentation_models PSPNet, schemas/segmentation.raml:226-249):
    stp_upsample_sum = sum of the TF-1.x bilinear resizes of up to four square maps (against oracle.nets.resize_bilinear_tf1 in float64,
    and against the per-level stp_resize_bilinear launches it replaces); stp_copy_cols_f32 = a column range of an fp32 matrix, copy and
    accumulate.  And the identity the restructuring rests on: conv1x1(concat(resized maps)) == sum of resize(conv1x1 with the map's columns)."""
    from oracle import nets as onets
    lib = ops._lib.load()
    n, ho, c, levels = case
    rng = np.random.RandomState(ho + c)
    maps = [q(rng.randn(n, l, l, c), dtype) for l in levels]
    ref = sum(onets.resize_bilinear_tf1(torch.from_numpy(m.astype(np.float64)).permute(0, 3, 1, 2), ho // l).permute(0, 2, 3, 1).numpy() for m, l in zip(maps, levels))
    md = [dev(m, dtype) for m in maps]
    y = keep(torch.full((n, ho, ho, c), float("nan"), dtype=TD[dtype], device=DEV))
    ptrs = [ops.ptr(t) for t in md] + [None] * (4 - len(md))
    hs = list(levels) + [0] * (4 - len(levels))
    ops._lib.check(lib.stp_upsample_sum(ptrs[0], ptrs[1], ptrs[2], ptrs[3], hs[0], hs[1], hs[2], hs[3], ops.ptr(y), n, ho, ho, c, ops.dt(y), ops.stream()),
                   "stp_upsample_sum")
    np.testing.assert_allclose(host(y), ref, atol=tol(ref, dtype))
    # the per-level resize launches, summed on the host: equal up to the one rounding per level they carry
    acc = np.zeros_like(ref)
    for t, l in zip(md, levels):
        z = keep(torch.empty((n, ho, ho, c), dtype=TD[dtype], device=DEV))
        ops._lib.check(lib.stp_resize_bilinear(ops.ptr(t), ops.ptr(z), n, l, l, c, ho // l, c, 0, ops.dt(z), ops.stream()), "stp_resize_bilinear")
        acc += host(z)
    np.testing.assert_allclose(host(y), acc, atol=tol(ref, dtype, 1.0 + len(levels)))
    # conv1x1(concat(resize(m_i))) == sum_i resize(conv1x1(m_i; the columns of part i)) in float64
    co = 16
    W = rng.randn(co, c * len(levels))
    cat = np.concatenate([onets.resize_bilinear_tf1(torch.from_numpy(m.astype(np.float64)).permute(0, 3, 1, 2), ho // l).permute(0, 2, 3, 1).numpy()
                          for m, l in zip(maps, levels)], axis=-1)
    lhs = cat @ W.T
    rhs = sum(onets.resize_bilinear_tf1(torch.from_numpy(m.astype(np.float64) @ W[:, i * c:(i + 1) * c].T).permute(0, 3, 1, 2), ho // l).permute(0, 2, 3, 1).numpy()
              for i, (m, l) in enumerate(zip(maps, levels)))
    np.testing.assert_allclose(lhs, rhs, atol=1e-9)
    if dtype == "fp32":
        rows, total, c0, cols = 24, 96, 32, 40
        A = rng.randn(rows, total).astype(np.float32)
        Ad = keep(torch.from_numpy(A).to(DEV))
        D = keep(torch.full((rows, cols), float("nan"), dtype=torch.float32, device=DEV))
        ops._lib.check(lib.stp_copy_cols_f32(ops.ptr(D), cols, ops.ptr(Ad) + 4 * c0, total, rows, cols, 0, ops.stream()), "stp_copy_cols_f32")
        np.testing.assert_array_equal(host(D), A[:, c0:c0 + cols])
        G = keep(torch.zeros((rows, total), dtype=torch.float32, device=DEV))
        for _ in range(2):
            ops._lib.check(lib.stp_copy_cols_f32(ops.ptr(G) + 4 * c0, total, ops.ptr(D), cols, rows, cols, 1, ops.stream()), "stp_copy_cols_f32")
        want = np.zeros_like(A)
        want[:, c0:c0 + cols] = 2 * A[:, c0:c0 + cols]
        np.testing.assert_array_equal(host(G), want)
        assert lib.stp_copy_cols_f32(ops.ptr(G) + 4, total, ops.ptr(D), cols, rows, cols, 0, ops.stream()) == -1      # misaligned: STP_E_BADARG


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("geom", [(1, 1, 24, 320), (2, 2, 12, 72), (3, 3, 8, 40), (6, 6, 4, 24), (5, 7, 8, 20), (3, 2, 6, 3), (12, 12, 8, 24), (16, 16, 4, 8),
                                  (5, 7, 2, 8), (9, 3, 3, 16)])
def test_pyramid_pooling_geometry_resize_and_pool(ops, dtype, geom):
    """PSPNet's pyramid at its real aspect: a (H x W) pooled map blown up by a large factor (level 1 is a single pixel whose
    gradient sums a whole feature map), channel counts that take the 16-byte (x8), 8-byte (x4) and scalar paths, and the
    window-per-workgroup average pooling with k*k >= 32."""
    from oracle import nets as onets
    from segmentation_training_pipeline_amd import _lib
    h, w, f, c = geom
    n = 2
    rng = np.random.RandomState(43)
    x = q(rng.randn(n, h, w, c), dtype)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = onets.resize_bilinear_tf1(xt, f)
    gy = q(rng.randn(n, h * f, w * f, c), dtype)
    ref.backward(torch.from_numpy(gy).permute(0, 3, 1, 2))
    xd = dev(x, dtype)
    y = torch.empty((n, h * f, w * f, c), dtype=TD[dtype], device=DEV)
    _lib.call("stp_resize_bilinear", ops.ptr(xd), ops.ptr(y), n, h, w, c, f, c, 0, ops.dt(xd), ops.stream())
    np.testing.assert_allclose(host(y), ref.detach().permute(0, 2, 3, 1).numpy(), atol=tol(x, dtype, 0.5))
    wsb = int(_lib.load().stp_resize_bilinear_bwd_workspace_bytes(n, h, w, c, f))
    assert wsb > 0 or 4 * f * f < 1024                       # the large-factor cases do take the split path
    ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=DEV)
    for acc, use_ws in ((0, True), (1, True), (0, False), (1, False)):
        base = q(rng.randn(n, h, w, c) * f, dtype)
        dx = dev(base, dtype)
        _lib.call("stp_resize_bilinear_bwd", ops.ptr(dev(gy, dtype)), ops.ptr(dx), n, h, w, c, f, c, 0, ops.dt(dx), acc,
                  ops.ptr(ws) if use_ws and wsb else None, wsb if use_ws else 0, ops.stream())
        want = xt.grad.permute(0, 2, 3, 1).numpy() + (base if acc else 0)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype, 1.0))
    # the pooling that produced such a map: window f x f over the (h*f, w*f) feature
    feat = q(rng.randn(n, h * f, w * f, c), dtype)
    ft = torch.from_numpy(feat).permute(0, 3, 1, 2).clone().requires_grad_(True)
    pr = torch.nn.functional.avg_pool2d(ft, f, f)
    gp = q(rng.randn(n, h, w, c), dtype)
    pr.backward(torch.from_numpy(gp).permute(0, 3, 1, 2))
    fd = dev(feat, dtype)
    yp = torch.empty((n, h, w, c), dtype=TD[dtype], device=DEV)
    wsp = int(_lib.load().stp_avgpool_workspace_bytes(n, h * f, w * f, c, f))
    wp = torch.empty(max(wsp, 4) // 4, dtype=torch.float32, device=DEV)
    for use_ws in (True, False):
        yp.fill_(9.0)
        _lib.call("stp_avgpool", ops.ptr(fd), ops.ptr(yp), n, h * f, w * f, c, f, ops.dt(fd), ops.ptr(wp) if use_ws and wsp else None,
                  wsp if use_ws else 0, ops.stream())
        np.testing.assert_allclose(host(yp), pr.detach().permute(0, 2, 3, 1).numpy(), atol=tol(feat, dtype, 0.1))
    for acc in (0, 1):
        base = q(rng.randn(n, h * f, w * f, c), dtype)
        dxp = dev(base, dtype)
        _lib.call("stp_avgpool_bwd", ops.ptr(dev(gp, dtype)), ops.ptr(dxp), n, h * f, w * f, c, f, ops.dt(fd), acc, ops.stream())
        want = ft.grad.permute(0, 2, 3, 1).numpy() + (base if acc else 0)
        np.testing.assert_allclose(host(dxp), want, atol=tol(want, dtype, 1.0))


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("geom", [(3, 1, 1, 1, 1), (3, 2, 0, 0, 1), (3, 1, 2, 2, 2), (3, 1, 4, 4, 4), (3, 2, 1, 1, 1)])
def test_depthwise_convolution_forward_and_gradients(ops, dtype, geom):
    """stp_dwconv / _dgrad / _wgrad (DeepLab model.py:136, 255-259): stride, dilation and explicit top/left padding (TF 'same'
    with stride 2 pads bottom/right only) against torch's grouped convolution and autograd."""
    from segmentation_training_pipeline_amd import _lib
    k, stride, pt, pl, dil = geom
    rng = np.random.RandomState(51)
    n, h, w, c = 2, 12, 10, 24
    keff = (k - 1) * dil + 1
    ho, wo = -(-h // stride), -(-w // stride)                               # 'same'
    pb, pr = max((ho - 1) * stride + keff - h - pt, 0), max((wo - 1) * stride + keff - w - pl, 0)
    x = q(rng.randn(n, h, w, c), dtype)
    wt = rng.randn(k, k, c).astype(np.float32) / 3
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    wtt = torch.from_numpy(wt).permute(2, 0, 1)[:, None].clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(xt, (pl, pr, pt, pb)), wtt, stride=stride, dilation=dil, groups=c)
    assert ref.shape[2:] == (ho, wo)
    gy = q(rng.randn(n, ho, wo, c), dtype)
    ref.backward(torch.from_numpy(gy).permute(0, 3, 1, 2))
    xd, wd, gyd = dev(x, dtype), keep(torch.from_numpy(wt).to(DEV)), dev(gy, dtype)
    y = torch.empty((n, ho, wo, c), dtype=TD[dtype], device=DEV)
    geo = (n, h, w, c, k, stride, pt, pl, dil, ho, wo, ops.dt(xd))
    _lib.call("stp_dwconv", ops.ptr(xd), ops.ptr(wd), ops.ptr(y), *geo, ops.stream())
    np.testing.assert_allclose(host(y), ref.detach().permute(0, 2, 3, 1).numpy(), atol=tol(host(y), dtype, 1.0))
    dx = torch.empty_like(xd)
    _lib.call("stp_dwconv_dgrad", ops.ptr(gyd), ops.ptr(wd), ops.ptr(dx), *geo, 0, ops.stream())
    np.testing.assert_allclose(host(dx), xt.grad.permute(0, 2, 3, 1).numpy(), atol=tol(xt.grad.numpy(), dtype, 1.0))
    dw = torch.full((k, k, c), float("nan"), device=DEV)
    ws = torch.empty(int(_lib.load().stp_dwconv_wgrad_workspace_bytes(c, k)) // 4, device=DEV)
    _lib.call("stp_dwconv_wgrad", ops.ptr(xd), ops.ptr(gyd), ops.ptr(dw), *geo, 0, ops.ptr(ws), ws.numel() * 4, ops.stream())
    refw = wtt.grad[:, 0].permute(1, 2, 0).numpy()
    np.testing.assert_allclose(host(dw), refw, atol=(2e-4 if dtype == "fp32" else 2e-3) * np.abs(refw).max() + 1e-5)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("sizes", [((5, 7), (40, 56)), ((1, 1), (6, 9)), ((6, 6), (6, 6)), ((4, 3), (9, 5))])
def test_align_corners_bilinear_dropout_sigmoid_and_probability_loss(ops, dtype, sizes):
    """The remaining DeepLab ops: resize_bilinear(align_corners=True) fwd/bwd vs torch, inverted dropout (mask reproducible,
    forward == backward map, fresh per tick), sigmoid activation + gradient, and the loss on probabilities."""
    from segmentation_training_pipeline_amd import _lib
    (h, w), (ho, wo) = sizes
    rng = np.random.RandomState(52)
    n, c = 2, 5
    x = q(rng.randn(n, h, w, c), dtype)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    ref = torch.nn.functional.interpolate(xt, size=(ho, wo), mode="bilinear", align_corners=True)
    gy = q(rng.randn(n, ho, wo, c), dtype)
    ref.backward(torch.from_numpy(gy).permute(0, 3, 1, 2))
    xd = dev(x, dtype)
    y = torch.empty((n, ho, wo, c), dtype=TD[dtype], device=DEV)
    _lib.call("stp_resize_bilinear_ac", ops.ptr(xd), ops.ptr(y), n, h, w, c, ho, wo, ops.dt(xd), ops.stream())
    np.testing.assert_allclose(host(y), ref.detach().permute(0, 2, 3, 1).numpy(), atol=tol(x, dtype, 0.5))
    dx = torch.empty_like(xd)
    _lib.call("stp_resize_bilinear_ac_bwd", ops.ptr(dev(gy, dtype)), ops.ptr(dx), n, h, w, c, ho, wo, ops.dt(xd), 0, ops.stream())
    want = xt.grad.permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype, 1.0))
    # dropout
    cnt = n * ho * wo * c
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    yd = dev(gy, dtype)
    o1, o2 = torch.empty_like(yd), torch.empty_like(yd)
    _lib.call("stp_dropout", ops.ptr(yd), ops.ptr(o1), cnt, 0.25, ops.ptr(state), 7, ops.dt(yd), ops.stream())
    _lib.call("stp_dropout", ops.ptr(yd), ops.ptr(o2), cnt, 0.25, ops.ptr(state), 7, ops.dt(yd), ops.stream())
    a1 = host(o1)
    np.testing.assert_array_equal(a1, host(o2))                                # same step, same salt: same mask
    kept = a1 != 0
    np.testing.assert_allclose(a1[kept], (gy / 0.75)[kept], rtol=1e-2 if dtype != "fp32" else 1e-6)
    _lib.call("stp_counter_tick", ops.ptr(state), ops.stream())
    _lib.call("stp_dropout", ops.ptr(yd), ops.ptr(o2), cnt, 0.25, ops.ptr(state), 7, ops.dt(yd), ops.stream())
    assert int(state[0].item()) == 1 and (cnt < 200 or not np.array_equal(host(o2) != 0, kept))
    if cnt > 2000:
        assert 0.18 < 1 - kept.mean() < 0.32
    # sigmoid activation on column 0 of a padded tensor and its gradient
    rows = n * ho * wo
    z = q(rng.randn(rows, 8), dtype)
    p = torch.zeros((rows, 1), dtype=TD[dtype], device=DEV)
    zd = dev(z, dtype)
    _lib.call("stp_sigmoid_act", ops.ptr(zd), ops.ptr(p), rows, 1, 8, 1, ops.dt(zd), ops.stream())
    pr = 1 / (1 + np.exp(-z[:, :1].astype(np.float64)))
    np.testing.assert_allclose(host(p), pr, atol=1e-6 if dtype == "fp32" else 4e-3)
    dp = q(rng.randn(rows, 8), dtype)
    dz = torch.empty((rows, 8), dtype=TD[dtype], device=DEV)
    _lib.call("stp_sigmoid_act_bwd", ops.ptr(p), ops.ptr(dev(dp, dtype)), ops.ptr(dz), rows, 1, 1, 8, ops.dt(zd), ops.stream())
    got = host(dz)
    np.testing.assert_allclose(got[:, 0], dp[:, 0] * host(p)[:, 0] * (1 - host(p)[:, 0]), atol=tol(dp, dtype, 0.5))
    np.testing.assert_array_equal(got[:, 1:], 0)
    # loss on probabilities
    pp = q(np.clip(rng.rand(rows), 0.0, 1.0), dtype)
    pp[:3] = q(np.array([0.0, 1.0, 0.4]), dtype)     # (not exactly 0.5: torch's max(z,0)/|z| formulation has a kink at z = 0)
    yy = (rng.rand(rows) < 0.3).astype(np.uint8)
    pt = torch.from_numpy(pp.astype(np.float32)).requires_grad_(True)
    yt = torch.from_numpy(yy.astype(np.float32))
    loss = olosses.composite_loss("binary_crossentropy+0.5*dice_loss", yt, pt)
    loss.backward()
    scal = torch.empty(10, device=DEV)
    dl = torch.full((rows, 8), float("nan"), dtype=TD[dtype], device=DEV)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    _lib.call("stp_prob_bce_dice", ops.ptr(dev(pp, dtype)), ops.ptr(keep(torch.from_numpy(yy).to(DEV))), rows, ops.dt(zd), 1.0, 0.5,
              ops.ptr(scal), ops.ptr(dl), 8, ops.ptr(ws), ws.numel() * 4, ops.stream())
    sc = host(scal)
    assert abs(sc[0] - float(loss.detach())) < 2e-5 * max(1.0, abs(float(loss.detach())))
    assert abs(sc[2] - float(olosses.dice_loss(yt, pt.detach()))) < 1e-5
    g = host(dl)
    refg = pt.grad.numpy()
    inr = (pp >= 1e-7) & (pp <= 1 - 1e-7) & (pp != 0.5)       # bf16 rounds some samples onto the kink as well
    np.testing.assert_allclose(g[inr, 0], refg[inr], rtol=2e-2 if dtype != "fp32" else 2e-4, atol=1e-6)
    np.testing.assert_array_equal(g[:, 1:], 0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_sigmoid_bce_dice_loss_and_gradient(ops, dtype):
    rng = np.random.RandomState(13)
    count = 2 * 48 * 48
    z = q(rng.randn(count) * 3, dtype)
    z[:4] = q(np.array([30.0, -30.0, 17.0, -17.0]), dtype)  # exercises the probability clip
    y = (rng.rand(count) < 0.3).astype(np.uint8)
    zt = torch.from_numpy(z).requires_grad_(True)
    yt = torch.from_numpy(y.astype(np.float32))
    p = torch.sigmoid(zt)
    loss = olosses.composite_loss("binary_crossentropy+0.5*dice_loss", yt, p)
    loss.backward()
    scal = torch.empty(10, device=DEV)
    C = 8 if dtype != "fp32" else 4
    dl = torch.full((count, C), float("nan"), dtype=TD[dtype], device=DEV)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    ops.sigmoid_bce_dice(dev(z, dtype), keep(torch.from_numpy(y).to(DEV)), count, 1.0, 0.5, scal, dl, C, 1.0, ws)
    s = host(scal)
    assert abs(s[0] - float(loss.detach())) < 1e-5 * max(1, abs(float(loss.detach())))
    assert abs(s[1] - float(olosses.binary_crossentropy(yt, p.detach()))) < 1e-5
    assert abs(s[2] - float(olosses.dice_loss(yt, p.detach()))) < 1e-5      # the north-star 1e-5 Dice bar
    assert abs(s[3] - float(olosses.dice_metric(yt, p.detach()))) < 1e-5
    assert abs(s[4] - float(olosses.binary_accuracy(yt, p.detach()))) < 1e-6
    assert abs(s[8] - float(olosses.iou_coef(yt, p.detach()))) < 1e-5 and abs(s[9] - float(olosses.iot_metric(yt, p.detach()))) < 1e-5
    g = host(dl)
    ref = zt.grad.numpy()
    np.testing.assert_allclose(g[:, 0], ref, atol=(1e-8 if dtype == "fp32" else 1e-2 * np.abs(ref).max()))
    np.testing.assert_array_equal(g[:, 1:], 0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("spec,w5", [
    ("iou_loss", (0, 0, 1, 0, 0)), ("jaccard_loss", (0, 0, 0, 1, 0)), ("focal_loss", (0, 0, 0, 0, 1)),
    ("binary_crossentropy+0.5*dice_loss+0.25*iou_loss+0.01*jaccard_loss+2.0*focal_loss", (1, 0.5, 0.25, 0.01, 2.0))])
def test_sigmoid_registry_losses_and_gradient(ops, dtype, spec, w5):
    """stp_sigmoid_loss_ex: the other entries of the loss registry (reference segmentation.py:15-22) against the oracle's
    formulas through torch autograd."""
    rng = np.random.RandomState(17)
    count = 2 * 48 * 48
    z = q(rng.randn(count) * 3, dtype)
    z[:4] = q(np.array([30.0, -30.0, 17.0, -17.0]), dtype)
    y = (rng.rand(count) < 0.3).astype(np.uint8)
    zt = torch.from_numpy(z).requires_grad_(True)
    yt = torch.from_numpy(y.astype(np.float32))
    p = torch.sigmoid(zt)
    # (the class axis of the oracle's jaccard_loss: one class)
    loss = olosses.composite_loss(spec, yt[:, None], p[:, None])
    loss.backward()
    scal = torch.empty(12, device=DEV)
    C = 8 if dtype != "fp32" else 4
    dl = torch.full((count, C), float("nan"), dtype=TD[dtype], device=DEV)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    ops.sigmoid_loss_ex(dev(z, dtype), keep(torch.from_numpy(y).to(DEV)), count, w5, scal, dl, C, 1.0, ws)
    s = host(scal)
    pd, y1 = p.detach()[:, None], yt[:, None]
    ref_loss = float(loss.detach())
    assert abs(s[0] - ref_loss) < 2e-5 * max(1, abs(ref_loss))
    assert abs(s[1] - float(olosses.binary_crossentropy(y1, pd))) < 1e-5
    assert abs(s[2] - float(olosses.dice_loss(y1, pd))) < 1e-5
    assert abs((1 - s[8]) - float(olosses.iou_loss(y1, pd))) < 1e-5
    assert abs(s[10] - float(olosses.jaccard_loss(y1, pd))) < 2e-5 * max(1, float(olosses.jaccard_loss(y1, pd)))
    assert abs(s[11] - float(olosses.focal_loss(y1, pd))) < 1e-5
    g = host(dl)
    ref = zt.grad.numpy()
    np.testing.assert_allclose(g[:, 0], ref, atol=(2e-8 if dtype == "fp32" else 1e-2 * np.abs(ref).max()), rtol=1e-4 if dtype == "fp32" else 2e-2)
    np.testing.assert_array_equal(g[:, 1:], 0)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("h,w", [(48, 50), (32, 32)])
def test_lovasz_hinge_loss_and_gradient(ops, dtype, h, w):
    """stp_lovasz_hinge (lovasz_loss of the registry, reference segmentation.py:18) against the oracle's per-image Lovasz hinge:
    three images, one of them without positives, image size not a multiple of the scan chunk; the result is ADDED to what
    stp_sigmoid_bce_dice left in scalars[0] and dlogits."""
    from segmentation_training_pipeline_amd import _lib
    rng = np.random.RandomState(23)
    n = 3
    count = n * h * w
    z = q(rng.randn(count) * 3, dtype)
    z[:4] = q(np.array([30.0, -30.0, 17.0, -17.0]), dtype)
    y = (rng.rand(count) < 0.3).astype(np.uint8)
    y[h * w:2 * h * w] = 0                                     # an image without positives
    zt = torch.from_numpy(z).requires_grad_(True)
    yt = torch.from_numpy(y.astype(np.float32))
    p = torch.sigmoid(zt)
    spec = "binary_crossentropy+0.5*lovasz_loss"
    loss = olosses.composite_loss(spec, yt.reshape(n, h, w, 1), p.reshape(n, h, w, 1))
    loss.backward()
    lov = float(olosses.lovasz_loss(yt.reshape(n, h, w, 1), p.detach().reshape(n, h, w, 1)))
    scal = torch.zeros(16, device=DEV)
    C = 8 if dtype != "fp32" else 4
    dl = torch.full((count, C), float("nan"), dtype=TD[dtype], device=DEV)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    zd, yd = dev(z, dtype), keep(torch.from_numpy(y).to(DEV))
    ops.sigmoid_bce_dice(zd, yd, count, 1.0, 0.0, scal, dl, C, 1.0, ws)
    nbytes = int(_lib.load().stp_lovasz_workspace_bytes(count, n))
    assert nbytes > 0
    wl = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    _lib.call("stp_lovasz_hinge", ops.ptr(zd), ops.ptr(yd), n, h * w, ops.dt(zd), 0.5, ops.ptr(scal), ops.ptr(dl), C, ops.ptr(wl), nbytes,
              ops.stream())
    s = host(scal)
    assert abs(s[12] - lov) < 1e-4 * max(1.0, lov)
    assert abs(s[0] - float(loss.detach())) < 1e-4 * max(1.0, abs(float(loss.detach())))
    g = host(dl)
    ref = zt.grad.numpy()
    if dtype == "fp32":
        np.testing.assert_allclose(g[:, 0], ref, atol=2e-8, rtol=1e-4)
    else:
        np.testing.assert_allclose(g[:, 0], ref, atol=1e-2 * np.abs(ref).max(), rtol=2e-2)
    np.testing.assert_array_equal(g[:, 1:], 0)
    # too small a workspace is refused
    assert _lib.load().stp_lovasz_hinge(ops.ptr(zd), ops.ptr(yd), n, h * w, ops.dt(zd), 0.5, ops.ptr(scal), ops.ptr(dl), C, ops.ptr(wl), 1024,
                                        None) != 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("classes,ldc", [(3, 3), (5, 8), (21, 24)])
def test_softmax_categorical_crossentropy_dice_loss_and_gradient(ops, dtype, classes, ldc):
    """stp_softmax_cce_dice / stp_softmax against the oracle's Keras categorical_crossentropy (+ musket dice over all class
    maps) through torch autograd; logits rows carry ldc >= classes channels (padded rows of the head conv)."""
    import ctypes as C
    from segmentation_training_pipeline_amd import _lib
    rng = np.random.RandomState(31)
    pixels = 2 * 24 * 24
    z = q(rng.randn(pixels, ldc) * 3, dtype)
    z[0, :classes] = q(np.array([40.0] + [-40.0] * (classes - 1)), dtype)       # exercises the probability clip
    t = rng.randint(0, classes, size=pixels).astype(np.uint8)
    t[0] = 1
    zt = torch.from_numpy(z[:, :classes].copy()).requires_grad_(True)
    yt = torch.nn.functional.one_hot(torch.from_numpy(t.astype(np.int64)), classes).to(torch.float32)
    p = torch.softmax(zt, dim=-1)
    loss = olosses.composite_loss("categorical_crossentropy+0.5*dice_loss", yt, p)
    loss.backward()
    scal = torch.empty(10, device=DEV)
    dlc = 8 * ((classes + 7) // 8)
    dl = torch.full((pixels, dlc), float("nan"), dtype=TD[dtype], device=DEV)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    zd, td = dev(z, dtype), keep(torch.from_numpy(t).to(DEV))
    _lib.call("stp_softmax_cce_dice", ops.ptr(zd), ops.ptr(td), pixels, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal), ops.ptr(dl), dlc,
              1.0, ops.ptr(ws), ws.numel() * 4, ops.stream())
    s = host(scal)
    pd = p.detach()
    assert abs(s[0] - float(loss.detach())) < 2e-5 * max(1, abs(float(loss.detach())))
    assert abs(s[1] - float(olosses.categorical_crossentropy(yt, pd))) < 2e-5
    assert abs(s[2] - float(olosses.dice_loss(yt, pd))) < 1e-5
    assert abs(s[3] - float(olosses.dice_metric(yt, pd))) < 1e-5
    assert abs(s[4] - float(olosses.binary_accuracy(yt, pd))) < 1e-6
    assert abs(s[8] - float(olosses.iou_coef(yt, pd))) < 1e-5 and abs(s[9] - float(olosses.iot_metric(yt, pd))) < 1e-5
    g = host(dl)
    ref = zt.grad.numpy()
    np.testing.assert_allclose(g[:, :classes], ref, atol=(2e-8 if dtype == "fp32" else 1e-2 * np.abs(ref).max()))
    np.testing.assert_array_equal(g[:, classes:], 0)
    probs = torch.empty((pixels, classes), device=DEV)
    _lib.call("stp_softmax", ops.ptr(zd), ops.ptr(probs), pixels, classes, ldc, ops.dt(zd), ops.stream())
    np.testing.assert_allclose(host(probs), pd.numpy(), atol=2e-6)


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 48, 48, 64, (8, 16, 24, 48)), (1, 36, 72, 24, (6, 12, 36)), (2, 24, 24, 20, (12, 24)), (1, 96, 96, 40, (16, 32, 48, 96))])
def test_pyramid_pooling_in_one_pass_over_the_feature_map(ops, dtype, case):
    """stp_avgpool_pyramid / _bwd (round 6: PSPNet's four AveragePooling2D of one feature map, windows that nest) against torch's
    avg_pool2d per level and against the separate stp_avgpool / stp_avgpool_bwd launches: the means come from the fp32 sums of the finest
    windows (one rounding per output), the gradient pass adds every level's dY / k^2 in one pass (with and without accumulation).  Channel
    counts on the 16-byte and 8-byte vector paths, non-square maps, two to four levels; the shapes it does not serve are refused."""
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    n, h, w, c, ks = case
    rng = np.random.RandomState(17 + c)
    x = q(rng.randn(n, h, w, c), dtype)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).clone().requires_grad_(True)
    kk = list(ks) + [0] * (4 - len(ks))
    assert lib.stp_avgpool_pyramid_ok(n, h, w, c, kk[0], kk[1], kk[2], kk[3], ops.dt(dev(x[:1, :1, :1], dtype))) == 1
    xd = dev(x, dtype)
    ys = [torch.full((n, h // k, w // k, c), 9.0, dtype=TD[dtype], device=DEV) for k in ks]
    wsb = int(lib.stp_avgpool_pyramid_workspace_bytes(n, h, w, c, ks[0], ops.dt(xd)))
    assert wsb > 0
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=DEV)
    yp = [ops.ptr(t) for t in ys] + [None] * (4 - len(ks))
    _lib.call("stp_avgpool_pyramid", ops.ptr(xd), yp[0], yp[1], yp[2], yp[3], kk[0], kk[1], kk[2], kk[3], n, h, w, c, ops.dt(xd), ops.ptr(ws), wsb,
              ops.stream())
    gys, loss = [], 0.0
    for k, y in zip(ks, ys):
        ref = torch.nn.functional.avg_pool2d(xt, k, k)
        np.testing.assert_allclose(host(y), ref.detach().permute(0, 2, 3, 1).numpy(), atol=tol(x, dtype, 0.1))
        # the separate launch: same sums in another order
        y1 = torch.empty_like(y)
        w1 = int(lib.stp_avgpool_workspace_bytes(n, h, w, c, k))
        wb1 = torch.empty(max(w1, 4) // 4, dtype=torch.float32, device=DEV)
        _lib.call("stp_avgpool", ops.ptr(xd), ops.ptr(y1), n, h, w, c, k, ops.dt(xd), ops.ptr(wb1) if w1 else None, w1, ops.stream())
        d = np.abs(host(y) - host(y1))
        assert d.max() <= tol(x, dtype, 0.02) and (dtype == "fp32" or np.mean(d == 0) > 0.95)      # (fp32: the order of the sums shows)
        gy = q(rng.randn(n, h // k, w // k, c), dtype)
        gys.append(gy)
        loss = loss + (ref * torch.from_numpy(gy).permute(0, 3, 1, 2)).sum()
    loss.backward()
    gref = xt.grad.permute(0, 2, 3, 1).numpy()
    gd = [dev(g, dtype) for g in gys]
    gp = [ops.ptr(t) for t in gd] + [None] * (4 - len(ks))
    for acc in (0, 1):
        base = q(rng.randn(n, h, w, c), dtype)
        dx = dev(base, dtype)
        _lib.call("stp_avgpool_pyramid_bwd", gp[0], gp[1], gp[2], gp[3], kk[0], kk[1], kk[2], kk[3], ops.ptr(dx), n, h, w, c, ops.dt(xd), acc, ops.stream())
        want = gref + (base if acc else 0)
        np.testing.assert_allclose(host(dx), want, atol=tol(want, dtype, 1.0))
    # refused: a window that is not a multiple of the finest one, a finest window below the window kernel's size, a missing output
    assert lib.stp_avgpool_pyramid_ok(n, h, w, c, ks[0], ks[0] + 1, 0, 0, ops.dt(xd)) == 0
    assert lib.stp_avgpool_pyramid_ok(n, h, w, c, 2, 4, 0, 0, ops.dt(xd)) == 0
    assert lib.stp_avgpool_pyramid(ops.ptr(xd), yp[0], None, None, None, kk[0], kk[1], 0, 0, n, h, w, c, ops.dt(xd), ops.ptr(ws), wsb, ops.stream()) != 0


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 12, 12, 8, 20, 20, 24), (2, 16, 20, 4, 3, 3, 8), (1, 9, 7, 2, 5, 8, 8), (1, 3, 5, 16, 2, 4, 8),
                                  (3, 1, 6, 4, 9, 12, 16), (1, 6, 1, 8, 30, 32, 32)])
def test_softmax_loss_on_upsampled_logits_without_the_upsampled_tensor(ops, dtype, case):
    """stp_softmax_cce_dice_up (the loss of PSPNet / FPN heads: bilinear resize of the class logits + softmax loss + the gradient of
    both) against the chain it replaces - stp_resize_bilinear, stp_softmax_cce_dice, stp_scale_by_device, stp_resize_bilinear_bwd - run
    on the device: same scalars, same low-resolution gradient up to the order of its fp32 sums (the fused form rounds where the chain
    stores: the resized logit, the per-pixel gradient).  Cases cover every factor, every class bucket, padded rows, one-row / one-column
    maps (both border clamps in one cell) and the device multiplier of the dynamic loss scale."""
    from segmentation_training_pipeline_amd import _lib
    n, h, w, f, classes, ldc, dlc = case
    rng = np.random.RandomState(5 + f + classes)
    ho, wo = h * f, w * f
    z = q(rng.randn(n, h, w, ldc) * 2.5, dtype)
    t = rng.randint(0, classes + 1, size=(n, ho, wo)).astype(np.uint8)      # (one value past the classes: clamped like the unfused kernel)
    zd, td = dev(z, dtype), keep(torch.from_numpy(t).to(DEV))
    gscale = 64.0 if dtype == "fp16" else 1.0
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    mult = keep(torch.tensor([4.0 if dtype == "fp16" else 1.0, 0, 0, 0, 0, 0, 0, 0], device=DEV))
    # the chain
    up = torch.empty((n, ho, wo, ldc), dtype=TD[dtype], device=DEV)
    _lib.call("stp_resize_bilinear", ops.ptr(zd), ops.ptr(up), n, h, w, ldc, f, ldc, 0, ops.dt(zd), ops.stream())
    scal0 = torch.zeros(12, device=DEV)
    dl = torch.full((n, ho, wo, dlc), float("nan"), dtype=TD[dtype], device=DEV)
    _lib.call("stp_softmax_cce_dice", ops.ptr(up), ops.ptr(td), n * ho * wo, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal0), ops.ptr(dl), dlc,
              gscale, ops.ptr(ws), ws.numel() * 4, ops.stream())
    _lib.call("stp_scale_by_device", ops.ptr(dl), n * ho * wo * dlc, ops.dt(zd), ops.ptr(mult), None, ops.stream())
    wsb = int(_lib.load().stp_resize_bilinear_bwd_workspace_bytes(n, h, w, dlc, f))
    wsr = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device=DEV)
    dlow0 = torch.full((n, h, w, dlc), float("nan"), dtype=TD[dtype], device=DEV)
    _lib.call("stp_resize_bilinear_bwd", ops.ptr(dl), ops.ptr(dlow0), n, h, w, dlc, f, dlc, 0, ops.dt(zd), 0, ops.ptr(wsr) if wsb else None, wsb,
              ops.stream())
    # the fused launch
    assert _lib.load().stp_softmax_cce_dice_up_ok(f, classes, ops.dt(zd)) == 1
    nb = int(_lib.load().stp_softmax_cce_dice_up_corner_bytes(n, h, w, classes))
    corners = torch.full((nb // 4,), float("nan"), dtype=torch.float32, device=DEV)
    scal1 = torch.zeros(12, device=DEV)
    dlow1 = torch.full((n, h, w, dlc), float("nan"), dtype=TD[dtype], device=DEV)
    rec = keep(torch.zeros(4, device=DEV))
    _lib.call("stp_softmax_cce_dice_up", ops.ptr(zd), ops.ptr(td), n, h, w, f, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal1), ops.ptr(dlow1),
              dlc, gscale, ops.ptr(mult), ops.ptr(rec), ops.ptr(ws), ws.numel() * 4, ops.ptr(corners), nb, ops.stream())
    s0, s1 = host(scal0), host(scal1)
    np.testing.assert_allclose(s1[:10], s0[:10], rtol=2e-6, atol=2e-6)       # (the partial sums are taken in another order)
    assert host(rec)[0] == host(mult)[0]
    g0, g1 = host(dlow0), host(dlow1)
    assert np.isfinite(g1).all()
    np.testing.assert_array_equal(g1[..., classes:], 0)
    scale = np.abs(g0).max()
    # fp32: the sums only differ in their order; 16-bit: one storage ulp of the result where a sum lands on a rounding boundary
    np.testing.assert_allclose(g1[..., :classes], g0[..., :classes], atol={"fp32": 2e-6, "bf16": 8e-3, "fp16": 1e-3}[dtype] * scale)
    if dtype != "fp32":
        assert np.mean(g1 == g0) > 0.9, np.mean(g1 == g0)
    # and the value against the oracle's loss on the resized logits (float64 lerp of the rounded inputs, rounded as the chain stores it)
    if dtype == "fp32":
        upz = host(up)[..., :classes].reshape(-1, classes)
        zt = torch.from_numpy(upz.copy())
        tt = np.minimum(t.reshape(-1), classes - 1).astype(np.int64)
        yt = torch.nn.functional.one_hot(torch.from_numpy(tt), classes).to(torch.float32)
        loss = olosses.composite_loss("categorical_crossentropy+0.5*dice_loss", yt, torch.softmax(zt, dim=-1))
        assert abs(s1[0] - float(loss)) < 2e-5 * max(1.0, abs(float(loss)))
    # scalars only (no gradient buffer)
    scal2 = torch.zeros(12, device=DEV)
    _lib.call("stp_softmax_cce_dice_up", ops.ptr(zd), ops.ptr(td), n, h, w, f, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal2), None, 0, 1.0,
              None, None, ops.ptr(ws), ws.numel() * 4, None, 0, ops.stream())
    np.testing.assert_array_equal(host(scal2)[:10], s1[:10])
    # argument checks: a factor the kernel does not serve, a corner table that is too small
    lib = _lib.load()
    assert lib.stp_softmax_cce_dice_up(ops.ptr(zd), ops.ptr(td), n, h, w, 3, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal2), None, 0, 1.0, None,
                                       None, ops.ptr(ws), ws.numel() * 4, None, 0, ops.stream()) != 0
    assert lib.stp_softmax_cce_dice_up(ops.ptr(zd), ops.ptr(td), n, h, w, f, classes, ldc, ops.dt(zd), 1.0, 0.5, ops.ptr(scal2), ops.ptr(dlow1), dlc,
                                       1.0, None, None, ops.ptr(ws), ws.numel() * 4, ops.ptr(corners), nb - 16, ops.stream()) != 0


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_overflow_guard_skips_the_optimizer_step(ops, bad):
    """stp_grad_global_scale on an arena that holds an inf / NaN (fp16 overflow under loss scaling) writes the skip marker; Adam,
    SGD, RMSprop and Nadam given that gscale leave parameters, moments and the step counter exactly as they were.  A finite arena
    restores the scale and the step runs."""
    rng = np.random.RandomState(3)
    n = 8192
    f = lambda a: keep(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV))
    g_bad = rng.randn(n).astype(np.float32)
    g_bad[1234] = bad
    gs = torch.tensor([0.5, 0.0], device=DEV)
    ws = torch.empty(1024, dtype=torch.float32, device=DEV)
    gd = f(g_bad)
    ops.grad_global_scale(gd, n, 0.0, 0.25, gs, ws)
    assert host(gs).tolist() == [-1.0, 1.0]
    p0 = rng.randn(n).astype(np.float32)
    lr = torch.tensor([1e-2], device=DEV)
    for name in ("adam", "nadam", "rmsprop", "sgd"):
        p, m, v = f(p0), f(np.full(n, 0.1)), f(np.full(n, 0.2))
        state = torch.tensor([7, 0], dtype=torch.int32, device=DEV)
        fst = torch.ones(8, device=DEV)
        if name == "adam":
            ops.adam(p, gd, m, v, n, lr, 0.9, 0.999, 1e-7, state, gscale=gs)
        elif name == "nadam":
            ops.nadam(p, gd, m, v, n, lr, 0.9, 0.999, 1e-7, 0.004, state, fst, gscale=gs)
        elif name == "rmsprop":
            ops.rmsprop(p, gd, m, n, lr, 0.9, 1e-7, gscale=gs)
        else:
            ops.sgd(p, gd, m, n, lr, 0.9, False, gscale=gs)
        assert np.array_equal(host(p), p0) and np.all(host(m) == np.float32(0.1)) and np.all(host(v) == np.float32(0.2)), name
        assert host(state.float()).tolist()[0] == 7.0 and np.all(host(fst) == 1.0), name
    good = f(rng.randn(n) * 0.1)
    ops.grad_global_scale(good, n, 0.0, 0.25, gs, ws)
    assert host(gs).tolist() == [0.25, 1.0]                     # scale restored, the count of skipped steps stays
    p, m, v = f(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    ops.adam(p, good, m, v, n, lr, 0.9, 0.999, 1e-7, state, gscale=gs)
    assert not np.array_equal(host(p), p0) and int(state[0].item()) == 1


def test_adam_and_sgd_match_keras_rules(ops):
    rng = np.random.RandomState(14)
    n = 4096 + 8
    p0 = rng.randn(n).astype(np.float32)
    grads = [rng.randn(n).astype(np.float32) * 0.1 for _ in range(3)]
    f = lambda a: keep(torch.from_numpy(a.copy()).to(DEV))
    # Adam
    oracle = ooptim.Adam(lr=1e-3)
    P = {"w": p0.copy()}
    p, m, v = f(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV)
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    for g in grads:
        oracle.step(P, {"w": g})
        ops.adam(p, f(g), m, v, n, lr, 0.9, 0.999, 1e-7, state)
    np.testing.assert_allclose(host(p), P["w"], atol=2e-6)
    assert int(state[0].item()) == 3
    # masked (frozen) elements do not move
    mask = torch.zeros(n, dtype=torch.uint8, device=DEV)
    mask[: n // 2] = 1
    before = host(p).copy()
    ops.adam(p, f(grads[0]), m, v, n, lr, 0.9, 0.999, 1e-7, state, mask=mask)
    after = host(p)
    np.testing.assert_array_equal(after[n // 2:], before[n // 2:])
    assert np.abs(after[: n // 2] - before[: n // 2]).max() > 0
    # SGD with momentum + nesterov
    for nesterov in (False, True):
        oracle = ooptim.SGD(lr=0.05, momentum=0.9, nesterov=nesterov)
        P = {"w": p0.copy()}
        p, vel = f(p0), torch.zeros(n, device=DEV)
        lr = torch.tensor([0.05], device=DEV)
        for g in grads:
            oracle.step(P, {"w": g})
            ops.sgd(p, f(g), vel, n, lr, 0.9, nesterov)
        np.testing.assert_allclose(host(p), P["w"], atol=2e-6)
    # RMSprop and Nadam (Keras 2.2.4 forms; Nadam's momentum schedule lives on the device: graph-replayable)
    oracle = ooptim.RMSprop(lr=1e-3)
    P = {"w": p0.copy()}
    p, acc = f(p0), torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], device=DEV)
    for g in grads:
        oracle.step(P, {"w": g})
        ops.rmsprop(p, f(g), acc, n, lr, 0.9, 1e-7)
    np.testing.assert_allclose(host(p), P["w"], atol=2e-6)
    oracle = ooptim.Nadam(lr=2e-3)
    P = {"w": p0.copy()}
    p, m, v = f(p0), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([2e-3], device=DEV)
    state = torch.zeros(2, dtype=torch.int32, device=DEV)
    fstate = torch.zeros(8, device=DEV)
    fstate[0] = 1.0
    for g in grads + grads:
        oracle.step(P, {"w": g})
        ops.nadam(p, f(g), m, v, n, lr, 0.9, 0.999, 1e-7, 0.004, state, fstate)
    np.testing.assert_allclose(host(p), P["w"], atol=5e-6)
    assert int(state[0].item()) == 6 and abs(float(fstate[0].item()) - oracle.m_schedule) < 1e-6
    # clipnorm scale
    g = f(grads[0])
    gs = torch.empty(1, device=DEV)
    ws = torch.empty(1024, device=DEV)
    ops.grad_global_scale(g, n, 0.5, 1.0, gs, ws)
    norm = np.sqrt((grads[0].astype(np.float64) ** 2).sum())
    assert abs(float(gs.item()) - min(1.0, 0.5 / norm)) < 1e-6


def test_augment_fixed_point_warp_bit_exact(ops):
    rng = np.random.RandomState(15)
    n, h, w = 3, 40, 52
    img = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    mask = (rng.rand(n, h, w) < 0.3).astype(np.uint8)
    mats = [oaug.affine_matrix(h, w, 1.2, (0.1, -0.05), 13.0, -7.0, True, False, (32, 48)),
            oaug.affine_matrix(h, w, 0.8, (-0.2, 0.2), -16.0, 16.0, False, True, (32, 48)),
            oaug.affine_matrix(h, w, out_hw=(32, 48))]
    prm = oaug.pack_params(mats, [12, -20, 0], [1.15, 0.8, 1.0])
    ri, rm = oaug.warp_u8(img, mask, prm, (32, 48))
    io = torch.empty((n, 32, 48, 3), dtype=torch.uint8, device=DEV)
    mo = torch.empty((n, 32, 48), dtype=torch.uint8, device=DEV)
    ops.augment_u8(keep(torch.from_numpy(img).to(DEV)), keep(torch.from_numpy(mask).to(DEV)), io, mo, keep(torch.from_numpy(prm).to(DEV)),
                   n, h, w, 32, 48, 3)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(io.cpu().numpy(), ri)   # byte work: bit-exact
    np.testing.assert_array_equal(mo.cpu().numpy(), rm)


@pytest.mark.parametrize("channels", [3, 1])
def test_augment_point_operations_bit_exact(ops, channels):
    """Every point operation of the catalogue (per-channel Add/Multiply, Multiply/AddElementwise, AdditiveGaussianNoise,
    Dropout, Grayscale, Invert) and the crop/pad canvas algebra against the numpy oracle: bytes must be identical."""
    rng = np.random.RandomState(16)
    n, h, w, oh, ow = 6, 36, 44, 32, 40
    img = rng.randint(0, 256, size=(n, h, w, channels)).astype(np.uint8)
    mask = rng.randint(0, 3, size=(n, h, w)).astype(np.uint8)
    mats = [oaug.compose(h, w, [("fliplr",), ("affine", 1.1, (0.05, 0.0), 8.0, 3.0)], (oh, ow)),
            oaug.compose(h, w, [("crop", 4, 6, 24, 30)], (oh, ow)),                 # CropToFixedSize
            oaug.compose(h, w, [("crop", -5, -3, 50, 52)], (oh, ow)),               # PadToFixedSize
            oaug.compose(h, w, [("crop", -2, -4, 41, 50), ("resize", h, w), ("flipud",)], (oh, ow)),   # Pad (keep size) + flip
            oaug.compose(h, w, [], (oh, ow)), oaug.compose(h, w, [], (oh, ow))]
    prm = oaug.pack_params(
        mats, [[3, -4, 5], 0, 0, 10, 0, -7], [[1.1, 0.9, 1.0], 1.0, 1.05, 1.0, 1.0, 1.0],
        invert=[False, True, False, False, True, False], gray_alpha=[0.0, 0.0, 1.0, 0.4, 0.0, 0.0],
        noise_sigma=[0.0, 8.0, 0.0, 12.75, 0.0, 25.0], noise_per_channel=[False, True, False, False, False, True],
        dropout_p=[0.0, 0.0, 0.2, 0.0, 0.1, 0.0], dropout_per_channel=[False, False, True, False, False, False],
        add_elem=[None, (-10, 10), None, (-3, 7), None, (0, 0)], add_elem_per_channel=[False, True, False, False, False, False],
        mul_elem=[None, None, (0.8, 1.2), None, (0.9, 1.1), None], mul_elem_per_channel=[False, False, True, False, False, False],
        seed=[0, 12345, 999, 77, 0xABCDE, 5])
    ri, rm = oaug.warp_u8(img, mask, prm, (oh, ow))
    io = torch.empty((n, oh, ow, channels), dtype=torch.uint8, device=DEV)
    mo = torch.empty((n, oh, ow), dtype=torch.uint8, device=DEV)
    ops.augment_u8(keep(torch.from_numpy(img).to(DEV)), keep(torch.from_numpy(mask).to(DEV)), io, mo, keep(torch.from_numpy(prm).to(DEV)),
                   n, h, w, oh, ow, channels)
    torch.cuda.synchronize()
    got = io.cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(got[i], ri[i], err_msg="sample %d" % i)
    np.testing.assert_array_equal(mo.cpu().numpy(), rm)
    # the generators do what their names say: noise ~ N(0, sigma), dropout rate ~ p
    flat = np.full((1, 64, 64, 3), 128, np.uint8)
    one = oaug.pack_params([oaug.compose(64, 64, [], (64, 64))], [0], [1.0], noise_sigma=[10.0], noise_per_channel=[True], seed=[42])
    o, _ = oaug.warp_u8(flat, None, one, (64, 64))
    d = o.astype(np.float64) - 128
    assert abs(d.mean()) < 0.5 and 9.0 < d.std() < 11.0
    one = oaug.pack_params([oaug.compose(64, 64, [], (64, 64))], [0], [1.0], dropout_p=[0.3], seed=[43])
    o, _ = oaug.warp_u8(flat, None, one, (64, 64))
    assert 0.25 < (o[..., 0] == 0).mean() < 0.35 and np.array_equal(o[..., 0] == 0, o[..., 2] == 0)


def test_neighbourhood_filters_bit_exact(ops):
    """stp_filter_u8 (GaussianBlur / AverageBlur / Sharpen / Emboss / EdgeDetect as K x K linear filters, MedianBlur as rank
    selection; reflect-101 border) against the numpy oracle, records produced by the host sampler."""
    from segmentation_training_pipeline_amd import augment
    rng = np.random.RandomState(17)
    specs = [[{"GaussianBlur": {"sigma": 1.2}}], [{"AverageBlur": {"k": 3}}], [{"AverageBlur": 4}], [{"MedianBlur": {"k": 5}}],
             [{"Sharpen": {"alpha": 0.6, "lightness": 1.3}}], [{"Emboss": {"alpha": 0.5, "strength": 1.5}}], [{"EdgeDetect": 0.4}],
             [{"GaussianBlur": 2.5}, {"Sharpen": {"alpha": 1.0, "lightness": 0.8}}], [], [{"MedianBlur": 3}]]
    per_image = []
    for sp_ in specs:
        _, f = augment.sample_batch_ex(sp_, rng, 1, 8, 8, (8, 8))
        per_image.append(f)
    passes = max(0 if f is None else f.shape[0] for f in per_image)
    recs = np.zeros((passes, len(specs), augment.FILTER_RECORD), np.int32)
    for i, f in enumerate(per_image):
        if f is not None:
            recs[:f.shape[0], i] = f[:, 0]
    assert passes == 2 and recs[0, 0, 0] == 9 and recs[0, 7, 0] == 13 and recs[1, 7, 0] == 3 and recs[0, 8, 0] == 0
    assert recs[0, 0, 4:4 + 81].sum() == 16384 and recs[0, 2, 0] == 5 and recs[0, 3, 1] == 1     # unit DC gain; even box in odd window
    # (round 6: the linear filters run on 64 x 16 tiles staged in LDS - several tiles with ragged edges, four channels, maps smaller than the kernel)
    for (h, w, c) in [(21, 30, 3), (5, 4, 1), (37, 150, 3), (18, 66, 4)]:
        img = rng.randint(0, 256, size=(len(specs), h, w, c)).astype(np.uint8)
        cur = keep(torch.from_numpy(img).to(DEV))
        ref = img
        for ps in range(passes):
            nxt = torch.empty_like(cur)
            ops.filter_u8(cur, nxt, keep(torch.from_numpy(recs[ps]).to(DEV)), len(specs), h, w, c)
            ref = oaug.filter_u8(ref, recs[ps])
            cur = keep(nxt)
        got = cur.cpu().numpy()
        for i in range(len(specs)):
            np.testing.assert_array_equal(got[i], ref[i], err_msg="spec %d at %dx%dx%d" % (i, h, w, c))
        np.testing.assert_array_equal(got[8], img[8])                         # no filter: copied through
    flat = np.full((1, 9, 9, 3), 77, np.uint8)
    for ps in range(passes):                                                  # constant images stay constant (DC gain exactly 1)
        for i in range(len(specs)):
            if recs[ps, i, 0] and (recs[ps, i, 1] == 1 or recs[ps, i, 4:].sum() == 16384):
                np.testing.assert_array_equal(oaug.filter_u8(flat, recs[ps, i:i + 1]), flat)


def test_listed_order_augmentation_passes_bit_exact(ops):
    """imgaug Sequential semantics on the device: augmenters listed in an order one pass cannot serve (colour before geometry, a
    blur between two warps, Add after Multiply, the Resize after colour) run as several passes (DeviceFeeder._run_passes); every
    pass is the bit-exact kernel pair, so the result equals the numpy oracle applied pass by pass with the same records.  Batch
    form (one launch per pass for the whole batch) and per-image form agree."""
    from segmentation_training_pipeline_amd import augment, pipeline
    spec = [{"Add": [-20, 20]}, {"Affine": {"rotate": [-20, 20], "scale": [0.8, 1.2]}}, {"GaussianBlur": {"sigma": [0.5, 1.5]}},
            {"Fliplr": 1.0}, {"Multiply": [0.8, 1.2]}, {"Add": [-5, 5]}, {"CropAndPad": {"percent": [-0.1, 0.1]}},
            {"AdditiveGaussianNoise": {"scale": 8.0}}]
    n, h, w, ch, out = 3, 40, 52, 3, (32, 48)
    rng = np.random.RandomState(23)
    img = rng.randint(0, 256, size=(n, h, w, ch)).astype(np.uint8)
    msk = rng.randint(0, 4, size=(n, h, w)).astype(np.uint8)
    passes, per = augment.sample_batch_staged(spec, np.random.RandomState(5), n, h, w, out)
    assert per is None and len(passes) == 6 and passes[-1][2] == out      # Add | Affine, blur | flip, Multiply | Add | CropAndPad, noise | Resize
    # the oracle, pass by pass
    rimg, rmsk = img, msk
    for prm, filt, hw in passes:
        rimg, rmsk = oaug.warp_u8(rimg, rmsk, prm, hw)
        if filt is not None:
            for ps in range(filt.shape[0]):
                rimg = oaug.filter_u8(rimg, filt[ps])
    feeder = pipeline.DeviceFeeder(DEV, out, spec, seed=0, classes=4, channels=ch)
    xd, yd = keep(torch.from_numpy(img).to(DEV)), keep(torch.from_numpy(msk).to(DEV))
    oi = torch.zeros((n,) + out + (ch,), dtype=torch.uint8, device=DEV)
    om = torch.zeros((n,) + out, dtype=torch.uint8, device=DEV)
    feeder._run_passes(xd, yd, oi, om, passes, n, h, w)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi.cpu().numpy(), rimg)
    np.testing.assert_array_equal(om.cpu().numpy(), rmsk)
    # per-image execution of the same records
    oi2, om2 = torch.zeros_like(oi), torch.zeros_like(om)
    for i in range(n):
        feeder._run_passes(xd[i], yd[i], oi2[i], om2[i], [(p[0][i:i + 1], None if p[1] is None else p[1][:, i:i + 1], p[2]) for p in passes], 1, h, w)
    torch.cuda.synchronize()
    assert torch.equal(oi2, oi) and torch.equal(om2, om)
    # and the order matters: the merged single-pass form of the same list gives a different image
    merged, mf = augment.sample_batch_ex(spec, np.random.RandomState(5), n, h, w, out)
    mi, _ = oaug.warp_u8(img, msk, merged, out)
    assert not np.array_equal(mi, rimg)


def test_displacement_field_augmenters_bit_exact(ops):
    """PiecewiseAffine / ElasticTransformation on the device (stp_field_piecewise / stp_field_elastic -> stp_augment_field_u8)
    against the numpy oracle: the fields themselves and the full listed-order pipeline through DeviceFeeder._run_passes (batch
    and per-image form), bit for bit."""
    from segmentation_training_pipeline_amd import augment, pipeline
    n, h, w, ch, out = 3, 40, 52, 3, (32, 48)
    rng = np.random.RandomState(29)
    img = rng.randint(0, 256, size=(n, h, w, ch)).astype(np.uint8)
    msk = rng.randint(0, 4, size=(n, h, w)).astype(np.uint8)
    # the fields
    grid = rng.randint(-300, 300, size=(n, 4, 5, 2)).astype(np.int32)
    fd = torch.empty((n, h, w), dtype=torch.int32, device=DEV)
    ops.field_piecewise(fd, keep(torch.from_numpy(grid).to(DEV)), n, h, w, 4, 5)
    ref = oaug.field_piecewise(grid, h, w)
    got = fd.cpu().numpy()
    np.testing.assert_array_equal((got << 16) >> 16, ref[..., 0])
    np.testing.assert_array_equal(got >> 16, ref[..., 1])
    bp, per = augment.sample_batch_staged([{"ElasticTransformation": {"alpha": [20, 40], "sigma": [0.0, 5.0]}}], np.random.RandomState(2), n, h, w, (h, w))
    recs = bp[0][3][1]
    recs[0, 2], recs[0, 4:] = 0, 0
    recs[0, 4] = 32768                                                        # sigma 0: the raw noise times alpha
    tmp = torch.empty_like(fd)
    ops.field_elastic(fd, tmp, keep(torch.from_numpy(recs).to(DEV)), n, h, w)
    ref = oaug.field_elastic(recs, h, w)
    got = fd.cpu().numpy()
    np.testing.assert_array_equal((got << 16) >> 16, ref[..., 0])
    np.testing.assert_array_equal(got >> 16, ref[..., 1])
    assert np.abs(ref[1:]).max() > 30                                         # (a visible displacement, > 0.5 px)
    # the pipeline: geometry + field in one pass, a second field after colour, a filter, the resize last
    spec = [{"Affine": {"rotate": [-20, 20], "scale": [0.8, 1.2]}}, {"PiecewiseAffine": {"scale": [0.02, 0.05]}}, {"Add": [-20, 20]},
            {"ElasticTransformation": {"alpha": [20, 40], "sigma": [2.0, 4.0]}}, {"GaussianBlur": {"sigma": [0.5, 1.5]}}]
    passes, per = augment.sample_batch_staged(spec, np.random.RandomState(5), n, h, w, out)
    assert per is None and [len(p) for p in passes] == [4, 4, 3] and passes[-1][2] == out
    rimg, rmsk = img, msk
    for p in passes:
        field = None
        if len(p) > 3:
            field = oaug.field_piecewise(p[3][3].reshape(n, p[3][1], p[3][2], 2), *p[2]) if p[3][0] == "piecewise" else oaug.field_elastic(p[3][1], *p[2])
        rimg, rmsk = oaug.warp_u8(rimg, rmsk, p[0], p[2], field)
        if p[1] is not None:
            for ps in range(p[1].shape[0]):
                rimg = oaug.filter_u8(rimg, p[1][ps])
    feeder = pipeline.DeviceFeeder(DEV, out, spec, seed=0, classes=4, channels=ch)
    xd, yd = keep(torch.from_numpy(img).to(DEV)), keep(torch.from_numpy(msk).to(DEV))
    oi = torch.zeros((n,) + out + (ch,), dtype=torch.uint8, device=DEV)
    om = torch.zeros((n,) + out, dtype=torch.uint8, device=DEV)
    feeder._run_passes(xd, yd, oi, om, passes, n, h, w)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi.cpu().numpy(), rimg)
    np.testing.assert_array_equal(om.cpu().numpy(), rmsk)
    # the fields moved pixels: the same records without them give another image
    plain, _ = oaug.warp_u8(img, msk, passes[0][0], passes[0][2])
    withf, _ = oaug.warp_u8(img, msk, passes[0][0], passes[0][2], oaug.field_piecewise(passes[0][3][3].reshape(n, 4, 4, 2), *passes[0][2]))
    assert (plain != withf).mean() > 0.3
    # per-image execution of the same records
    oi2, om2 = torch.zeros_like(oi), torch.zeros_like(om)
    for i in range(n):
        one = []
        for p in passes:
            q = (p[0][i:i + 1], None if p[1] is None else p[1][:, i:i + 1], p[2])
            if len(p) > 3:
                q += ((p[3][0], p[3][1], p[3][2], p[3][3][i:i + 1]) if p[3][0] == "piecewise" else (p[3][0], p[3][1][i:i + 1]),)
            one.append(q)
        feeder._run_passes(xd[i], yd[i], oi2[i], om2[i], one, 1, h, w)
    torch.cuda.synchronize()
    assert torch.equal(oi2, oi) and torch.equal(om2, om)


def test_background_replacer_bit_exact(ops, tmp_path):
    """stp_background_replace_u8 against the oracle (erosion 0 / 1 / 3, masks touching the border) and the whole BackgroundReplacer
    pass through DeviceFeeder._run_passes: sub-augmenters, background resized to the item, composite, then the pass's warp."""
    import os
    from PIL import Image
    from segmentation_training_pipeline_amd import augment, pipeline
    rng = np.random.RandomState(31)
    n, h, w = 2, 37, 45
    img = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    bgi = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    msk = np.stack([((yy - 18) ** 2 + (xx - 20) ** 2 < 150).astype(np.uint8), (xx > 30).astype(np.uint8) * 3])
    for er in (0, 1, 3):
        out = torch.zeros((n, h, w, 3), dtype=torch.uint8, device=DEV)
        ops.background_replace_u8(keep(torch.from_numpy(img).to(DEV)), keep(torch.from_numpy(msk).to(DEV)), keep(torch.from_numpy(bgi).to(DEV)),
                                  out, n, h, w, 3, er)
        np.testing.assert_array_equal(out.cpu().numpy(), oaug.background_replace_u8(img, msk, bgi, er))
    os.makedirs(str(tmp_path / "bg"))
    for i in range(2):
        Image.fromarray(rng.randint(0, 256, size=(50 + 7 * i, 64, 3)).astype(np.uint8)).save(str(tmp_path / "bg" / ("b%d.png" % i)))
    spec = [{"BackgroundReplacer": {"path": str(tmp_path / "bg"), "rate": 0.0, "erosion": 2, "augmenters": {"Fliplr": 1.0}}},
            {"Affine": {"rotate": [-10, 10]}}, {"Add": [-10, 10]}]
    out_hw = (32, 40)
    passes = augment.sample_staged(spec, np.random.RandomState(3), h, w, out_hw)
    assert [len(p) for p in passes] == [3, 5, 3]                  # Fliplr | background, Affine, Add | the trailing Resize
    rimg, rmsk = img[:1], msk[:1]
    for p in passes:
        if len(p) > 4:
            bg, er = p[4]
            bgr, _ = oaug.warp_u8(bg[None], None, augment.identity_batch(1, bg.shape[0], bg.shape[1], rimg.shape[1:3]), rimg.shape[1:3])
            rimg = oaug.background_replace_u8(rimg, rmsk, bgr, er)
        rimg, rmsk = oaug.warp_u8(rimg, rmsk, p[0][None], p[2])
    feeder = pipeline.DeviceFeeder(DEV, out_hw, spec, seed=0, classes=4, channels=3)
    oi = torch.zeros(out_hw + (3,), dtype=torch.uint8, device=DEV)
    om = torch.zeros(out_hw, dtype=torch.uint8, device=DEV)
    feeder._run_passes(keep(torch.from_numpy(img[0]).to(DEV)), keep(torch.from_numpy(msk[0]).to(DEV)), oi, om,
                       [augment.batch_of_one(p) for p in passes], 1, h, w)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(oi.cpu().numpy(), rimg[0])
    np.testing.assert_array_equal(om.cpu().numpy(), rmsk[0])


@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_batched_weight_prepare_equals_per_layer(ops, dtype):
    """The once-per-step batched launch (LDS tile transpose) must write exactly what stp_weight_prepare writes
    layer by layer: forward copies with channel/tap padding, flipped + transposed data-gradient copies."""
    import ctypes as C
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    vec = 8 if dtype != "fp32" else 4
    rng = np.random.RandomState(3)
    # Cout, KH, KW, Cin(master), KWp, Cinp, want_bwd
    layers = [(64, 7, 7, 3, 8, 4, False), (64, 3, 3, 64, 3, 64, True), (40, 3, 3, 24, 3, 24, True), (1, 3, 3, 16, 3, 16, True),
              (136, 1, 1, 72, 1, 72, True), (16, 3, 3, 48, 3, 48, False)]
    dsz = int(lib.stp_weight_prepare_desc_bytes())
    hostbuf = (C.c_char * (dsz * len(layers)))()
    total, outs = 0, []
    for i, (co, kh, kw, ci, kwp, cinp, want_bwd) in enumerate(layers):
        coutb = (co + vec - 1) // vec * vec
        master = keep(torch.from_numpy(rng.randn(co, kh, kw, ci).astype(np.float32)).to(DEV))
        rows_f, rows_b = (co + 15) // 16 * 16, (cinp + 15) // 16 * 16
        mk = lambda n: torch.full((n,), float("nan"), dtype=TD[dtype], device=DEV)
        f1, f2 = mk(rows_f * kh * kwp * cinp), mk(rows_f * kh * kwp * cinp)
        b1 = b2 = None
        if want_bwd:
            b1, b2 = mk(rows_b * kh * kw * coutb), mk(rows_b * kh * kw * coutb)
        ops.weight_prepare(master, f1, b1, co, kh, kw, ci, kwp, cinp, coutb, ops.dt(f1))
        total += int(lib.stp_weight_prepare_desc_fill(C.cast(hostbuf, C.c_void_p), i, total, ops.ptr(master), ops.ptr(f2), ops.ptr(b2),
                                                      co, kh, kw, ci, kwp, cinp, coutb))
        outs.append((f1, f2, b1, b2))
    dev_desc = keep(torch.frombuffer(bytearray(bytes(hostbuf)), dtype=torch.uint8).to(DEV))
    _lib.call("stp_weight_prepare_batched", dev_desc.data_ptr(), len(layers), total, ops.dt(outs[0][0]), ops.stream())
    for f1, f2, b1, b2 in outs:
        np.testing.assert_array_equal(host(f2), host(f1))
        if b1 is not None:
            np.testing.assert_array_equal(host(b2), host(b1))


def test_bf16_wire_casts(ops):
    x = torch.randn(10007, device=DEV)
    b = torch.empty(10007, dtype=torch.bfloat16, device=DEV)
    ops.cast_f32_to_bf16(x, b, x.numel())
    torch.cuda.synchronize()
    assert torch.equal(b, x.to(torch.bfloat16))
    y = torch.empty_like(x)
    ops.cast_bf16_to_f32(b, y, x.numel(), 0.5)
    torch.cuda.synchronize()
    assert torch.equal(y, b.to(torch.float32) * 0.5)


def test_each_build_rejects_the_other_builds_16bit_code(ops):
    """libstp_hip.so serves STP_F32 + STP_BF16, libstp_hip_f16.so STP_F32 + STP_F16 (include/stp_hip.h): the other 16-bit code must
    come back as STP_E_BADARG from the plain-argument entry points and from the parameter-struct ones, not run as something else."""
    from segmentation_training_pipeline_amd import _lib
    x = torch.zeros(1024, dtype=torch.float16, device=DEV)
    pr = torch.empty(1024, dtype=torch.float32, device=DEV)
    y = torch.empty((1, 8, 8, 16), dtype=torch.float16, device=DEV)
    w = torch.zeros(16 * 9 * 16, dtype=torch.float16, device=DEV)
    for storage, good, bad in (("bf16", _lib.BF16, _lib.F16), ("fp16", _lib.F16, _lib.BF16)):
        lib = _lib.load(storage)
        assert lib.stp_storage_dtype() == good
        assert lib.stp_sigmoid(ops.ptr(x), ops.ptr(pr), 1024, good, ops.stream()) == 0
        assert lib.stp_sigmoid(ops.ptr(x), ops.ptr(pr), 1024, bad, ops.stream()) == -1
        assert lib.stp_add_inplace(ops.ptr(x), ops.ptr(x), 1024, bad, ops.stream()) == -1
        assert lib.stp_sigmoid(ops.ptr(x), ops.ptr(pr), 1024, 7, ops.stream()) == -1
        P = ops.conv_params(x.view(1, 8, 8, 16), w, y, N=1, Hs0=8, Ws0=8, Hv=8, Wv=8, C0=16, KH=3, KW=3, stride=1, pad=1, Ho=8, Wo=8, Cout=16,
                            dtype=bad)
        import ctypes
        assert lib.stp_conv2d(ctypes.byref(P), ops.stream()) == -1
        P.dtype = good
        assert lib.stp_conv2d(ctypes.byref(P), ops.stream()) == 0
    torch.cuda.synchronize()


def test_lean_kernels_equal_the_generic_ones():
    """conv_sc_lean.hip (lean small-channel kernels, persistent stem kernels) against the generic kernels they replace: the switches are
    read once per process, so the same seeded launches (tests/_lean_vs_generic.py: every configuration of the training step, a shape
    with interior and ragged border tiles) run in two subprocesses.  Outputs: BIT-identical (same MFMA order, one rounding); statistic
    sums / the stem's weight gradient: equal to fp32 rounding (other summation order)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for v in ("0", "1"):
        env = dict(os.environ, STP_SC_LEAN=v, STP_STEM_LEAN=v, STP_STEM_WG_LEAN=v, PYTHONPATH=root)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "_lean_vs_generic.py")], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("LEANJSON ")][-1][9:]))
    gen, lean = res
    assert set(gen) == set(lean)
    for k in gen:
        assert gen[k]["y"] == lean[k]["y"], k
        a, b = np.array(gen[k]["sums"]), np.array(lean[k]["sums"])
        assert a.shape == b.shape, k
        if a.size:
            np.testing.assert_allclose(b, a, atol=3e-5 * np.abs(a).max() + 1e-6, err_msg=k)


def test_opt_in_statistic_groups_run_in_their_own_process():
    """STP_STATS_GROUP=1 (the group-level pre-reduction, opt-in: measured slower) is read once per process: its op tests - exact group
    sums, counters back at zero, replay bit-identical - and a whole hipGraph-vs-eager training step run in a subprocess with the switch on."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STP_STATS_GROUP="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", os.path.join(root, "tests", "test_ops_gpu.py"), "-k", "pre_reduced",
                        os.path.join(root, "tests", "test_model_gpu.py") + "::test_hipgraph_replay_equals_eager"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]


def test_all_operands_epilogue_copies_in_their_own_process():
    """STP_EPILOGUE_SPECIAL=0 (the A/B switch of DESIGN 3.11, read once per process) sends every launch of the halo and per-tap kernels through
    the all-operands copy of the epilogue's element loop instead of the per-combination copies: the same op tests (halo forward / statistics /
    residual / BatchNormalization backward, the per-tap kernel's epilogue forms, the class heads) and a whole training step against the oracle
    run in a subprocess with the switch off, so the copy that the A/Bs compare against stays correct."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STP_EPILOGUE_SPECIAL="0", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", os.path.join(root, "tests", "test_ops_gpu.py"), "-k",
                        "halo_kernel_forward_statistics_residual or halo_kernel_batchnorm_backward_sums or tap_channels or conv2d_bnb or fused_bn_backward or residual",
                        os.path.join(root, "tests", "test_model_gpu.py") + "::test_hipgraph_replay_equals_eager"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]
