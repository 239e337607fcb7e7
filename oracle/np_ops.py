"""Oracle (test infrastructure): naive numpy restatements of the layer arithmetic.

Small-shape cross-checks for the torch primitives used by :mod:`oracle.nets`, and the
direct per-op oracles the HIP unit tests compare against (NHWC activations, HWIO
kernels - the Keras layouts named in SURVEY Appendix A.4).  Float64 accumulation.
PARITY UNPINNED (see oracle/__init__.py).
"""
import numpy as np


def conv2d(x, w, stride=1, pad=0, bias=None):
    """x [N,H,W,Ci], w [KH,KW,Ci,Co]; symmetric zero pad then 'valid' correlation."""
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    xp = np.zeros((n, h + 2 * pad, wd + 2 * pad, ci), np.float64)
    xp[:, pad:pad + h, pad:pad + wd] = x
    ho = (h + 2 * pad - kh) // stride + 1
    wo = (wd + 2 * pad - kw) // stride + 1
    out = np.zeros((n, ho, wo, co), np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride]
            out += patch @ w[i, j].astype(np.float64)
    if bias is not None:
        out += bias
    return out


def conv2d_dgrad(dy, w, in_hw, stride=1, pad=0):
    """Gradient of conv2d w.r.t. x.  dy [N,Ho,Wo,Co] -> [N,H,W,Ci]."""
    n, ho, wo, co = dy.shape
    kh, kw, ci, _ = w.shape
    h, wd = in_hw
    dxp = np.zeros((n, h + 2 * pad, wd + 2 * pad, ci), np.float64)
    for i in range(kh):
        for j in range(kw):
            dxp[:, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride] += \
                dy.astype(np.float64) @ w[i, j].astype(np.float64).T
    return dxp[:, pad:pad + h, pad:pad + wd]


def conv2d_wgrad(x, dy, ksize, stride=1, pad=0):
    """Gradient of conv2d w.r.t. w.  Returns [KH,KW,Ci,Co]."""
    n, h, wd, ci = x.shape
    _, ho, wo, co = dy.shape
    kh, kw = ksize
    xp = np.zeros((n, h + 2 * pad, wd + 2 * pad, ci), np.float64)
    xp[:, pad:pad + h, pad:pad + wd] = x
    dw = np.zeros((kh, kw, ci, co), np.float64)
    d2 = dy.reshape(-1, co).astype(np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + (ho - 1) * stride + 1:stride, j:j + (wo - 1) * stride + 1:stride]
            dw[i, j] = patch.reshape(-1, ci).T @ d2
    return dw


def bn_train(x, gamma, beta, eps):
    """Keras BatchNormalization, training phase (biased batch variance)."""
    x = x.astype(np.float64)
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    y = (x - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * gamma
    return y + beta, mean, var


def bn_train_bwd(x, dy, gamma, eps):
    x = x.astype(np.float64)
    dy = dy.astype(np.float64)
    m = x.shape[0] * x.shape[1] * x.shape[2]
    mean = x.mean(axis=(0, 1, 2))
    var = x.var(axis=(0, 1, 2))
    inv = 1.0 / np.sqrt(var + eps)
    xh = (x - mean) * inv
    dbeta = dy.sum(axis=(0, 1, 2))
    dgamma = (dy * xh).sum(axis=(0, 1, 2))
    g = inv if gamma is None else inv * gamma
    dx = g * (dy - dbeta / m - xh * dgamma / m)
    return dx, dgamma, dbeta


def maxpool3x3s2(x):
    """ZeroPadding2D(1) + MaxPooling2D(3, strides=2, 'valid') (classification_models stem)."""
    n, h, w, c = x.shape
    xp = np.zeros((n, h + 2, w + 2, c), x.dtype)
    xp[:, 1:-1, 1:-1] = x
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    out = np.full((n, ho, wo, c), -np.inf, np.float64)
    for i in range(3):
        for j in range(3):
            out = np.maximum(out, xp[:, i:i + 2 * ho - 1:2, j:j + 2 * wo - 1:2])
    return out


def upsample2x(x):
    return x.repeat(2, axis=1).repeat(2, axis=2)


def upsample2x_bwd(dy):
    n, h, w, c = dy.shape
    return dy.reshape(n, h // 2, 2, w // 2, 2, c).sum(axis=(2, 4))
