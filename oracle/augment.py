"""Oracle (test infrastructure): numpy restatement of the augmentation stage.

The reference runs ``imgaug.augmenters.Sequential(augmentation + Resize)`` jointly on image and
segmentation map in worker processes (README.md:247-268; augmenter catalogue
schemas/augmenters.raml:43-133).  imgaug 0.3.0 (requires.txt:11) is not vendored or installable
here: PARITY UNPINNED.  Parity is defined per sample *given explicit parameters* (imgaug's numpy
RandomState stream inside worker processes is not reproducible by design).

``affine_matrix`` restates imgaug's ``Affine`` parameterisation (skimage AffineTransform about
the image centre: scale, translate_percent, rotate, shear) composed with Fliplr/Flipud, as the
OUTPUT->INPUT map that ``cv2.warpAffine`` consumes.  ``warp_u8`` restates the sampling with
cv2-style fixed point (10 fractional coordinate bits, 5-bit bilinear weights, constant-0 border,
nearest for masks), in pure integer arithmetic so that it is bit-reproducible.
"""
import numpy as np


def affine_matrix(h, w, scale=1.0, translate_percent=(0.0, 0.0), rotate=0.0, shear=0.0,
                  fliplr=False, flipud=False, out_hw=None):
    """3x3 matrix mapping OUTPUT pixel coords (x,y,1) to INPUT pixel coords."""
    sx = sy = float(scale)
    rot, sh = np.deg2rad(rotate), np.deg2rad(shear)
    tx, ty = translate_percent[0] * w, translate_percent[1] * h
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    to_origin = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1]], np.float64)
    aff = np.array([[sx * np.cos(rot), -sy * np.sin(rot + sh), tx],
                    [sx * np.sin(rot), sy * np.cos(rot + sh), ty],
                    [0, 0, 1]], np.float64)
    back = np.array([[1, 0, cx], [0, 1, cy], [0, 0, 1]], np.float64)
    fwd = back @ aff @ to_origin                      # input -> output (imgaug/skimage convention)
    if fliplr:
        fwd = np.array([[-1, 0, w - 1], [0, 1, 0], [0, 0, 1]], np.float64) @ fwd
    if flipud:
        fwd = np.array([[1, 0, 0], [0, -1, h - 1], [0, 0, 1]], np.float64) @ fwd
    inv = np.linalg.inv(fwd)                          # output -> input
    if out_hw is not None and tuple(out_hw) != (h, w):
        # trailing Resize to the network shape: output pixel centres map linearly onto the warped image
        oh, ow = out_hw
        rs = np.array([[w / ow, 0, 0.5 * w / ow - 0.5], [0, h / oh, 0.5 * h / oh - 0.5], [0, 0, 1]], np.float64)
        inv = inv @ rs
    return inv


AUG_RECORD = 24          # floats per sample, layout in include/stp_hip.h (stp_augment_u8)
F_INVERT, F_NOISE_PC, F_DROP_PC, F_ADDE_PC, F_MULE_PC, F_ADDE, F_MULE = 1, 2, 4, 8, 16, 32, 64
IRWIN_HALL_STD = 147.8   # std of the sum of 4 uniform bytes: sqrt(4 * (256^2 - 1) / 12)


def pack_params(mats, adds, muls, **point):
    """float32 [N,24] rows (the kernel's per-sample record).  ``adds`` / ``muls``: scalar or 3 per sample.  Optional
    per-sample lists in ``point``: invert, gray_alpha, noise_sigma, noise_per_channel, dropout_p, dropout_per_channel,
    add_elem (lo, hi), add_elem_per_channel, mul_elem (lo, hi), mul_elem_per_channel, seed."""
    n = len(mats)
    out = np.zeros((n, AUG_RECORD), np.float32)
    g = lambda k, i, d: (point[k][i] if k in point and point[k] is not None else d)
    for i in range(n):
        out[i, 0:3] = mats[i][0]
        out[i, 3:6] = mats[i][1]
        out[i, 6:9] = adds[i]
        out[i, 9:12] = muls[i]
        flags = 0
        if g("invert", i, False):
            flags |= F_INVERT
        out[i, 13] = int(round(float(g("gray_alpha", i, 0.0)) * 256))
        sig = float(g("noise_sigma", i, 0.0))
        out[i, 14] = int(round(sig * 65536.0 / IRWIN_HALL_STD))
        if g("noise_per_channel", i, False):
            flags |= F_NOISE_PC
        out[i, 15] = int(round(float(g("dropout_p", i, 0.0)) * (1 << 24)))
        if g("dropout_per_channel", i, False):
            flags |= F_DROP_PC
        ae = g("add_elem", i, None)
        if ae is not None:
            flags |= F_ADDE | (F_ADDE_PC if g("add_elem_per_channel", i, False) else 0)
            out[i, 16], out[i, 17] = int(ae[0]), int(ae[1])
        me = g("mul_elem", i, None)
        if me is not None:
            flags |= F_MULE | (F_MULE_PC if g("mul_elem_per_channel", i, False) else 0)
            out[i, 18], out[i, 19] = me[0], me[1]
        out[i, 12] = flags
        out[i, 20] = int(g("seed", i, 0)) & 0xffffff
    return out


def aug_hash(seed, pix, ch, op):
    """The kernel's counter-based hash in uint32 arithmetic (pix, ch: arrays)."""
    with np.errstate(over="ignore"):
        h = (np.uint32(seed) ^ (pix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (ch.astype(np.uint32) * np.uint32(0x85EBCA77))
             ^ np.uint32((op * 0xC2B2AE3D) & 0xffffffff))
        h ^= h >> np.uint32(16); h = h * np.uint32(0x7feb352d)
        h ^= h >> np.uint32(15); h = h * np.uint32(0x846ca68b)
        h ^= h >> np.uint32(16)
    return h


def point_ops(v, rec):
    """v: int64 [H,W,C] after the warp; rec: one float32[24] record -> uint8 [H,W,C] (bit-exact restatement)."""
    oh, ow, c = v.shape
    flags = int(rec[12])
    seed = int(rec[20])
    pix = (np.arange(oh, dtype=np.uint32)[:, None] * np.uint32(ow) + np.arange(ow, dtype=np.uint32)[None, :])[..., None]
    pix = np.broadcast_to(pix, (oh, ow, c))
    chan = np.broadcast_to(np.arange(c, dtype=np.uint32)[None, None, :], (oh, ow, c))
    zero = np.zeros_like(chan)
    cc = np.minimum(np.arange(c), 2)
    v = np.clip(v + rec[6:9].astype(np.int64)[cc][None, None, :], 0, 255)
    mul = rec[9:12][cc]
    for k in range(c):
        if mul[k] != np.float32(1.0):
            v[..., k] = np.clip(np.rint(v[..., k].astype(np.float32) * mul[k]).astype(np.int64), 0, 255)
    if flags & F_MULE:
        h = aug_hash(seed, pix, chan if flags & F_MULE_PC else zero, 1)
        u = (h >> np.uint32(8)).astype(np.float32) * np.float32(5.9604644775390625e-8)
        m = rec[18] + (rec[19] - rec[18]) * u                      # float32, one rounding per operation
        v = np.clip(np.rint(v.astype(np.float32) * m).astype(np.int64), 0, 255)
    if flags & F_ADDE:
        lo, nvals = int(rec[16]), int(rec[17]) - int(rec[16]) + 1
        h = aug_hash(seed, pix, chan if flags & F_ADDE_PC else zero, 2)
        v = np.clip(v + lo + (h % np.uint32(nvals)).astype(np.int64), 0, 255)
    k = int(rec[14])
    if k:
        h = aug_hash(seed, pix, chan if flags & F_NOISE_PC else zero, 3).astype(np.int64)
        ssum = (h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)
        v = np.clip(v + (((ssum - 510) * k + 32768) >> 16), 0, 255)
    t = int(rec[15])
    if t:
        h = aug_hash(seed, pix, chan if flags & F_DROP_PC else zero, 4)
        v = np.where((h >> np.uint32(8)).astype(np.int64) < t, 0, v)
    gq = int(rec[13])
    if gq and c == 3:
        gray = (v[..., 0] * 4899 + v[..., 1] * 9617 + v[..., 2] * 1868 + 8192) >> 14
        v = (gq * gray[..., None] + (256 - gq) * v + 128) >> 8
    if flags & F_INVERT:
        v = 255 - v
    return v.astype(np.uint8)


# ---- sequential geometric composition (imgaug Sequential applies the augmenters in the listed order) -------------
class Canvas(object):
    """Tracks the current canvas size and the 3x3 map from current-canvas pixel coords to ORIGINAL input pixel coords."""

    def __init__(self, h, w):
        self.h, self.w = float(h), float(w)
        self.M = np.eye(3)

    def _push(self, cur_to_prev):
        self.M = self.M @ cur_to_prev

    def fliplr(self):
        self._push(np.array([[-1, 0, self.w - 1], [0, 1, 0], [0, 0, 1]], np.float64))

    def flipud(self):
        self._push(np.array([[1, 0, 0], [0, -1, self.h - 1], [0, 0, 1]], np.float64))

    def affine(self, scale=1.0, translate_percent=(0.0, 0.0), rotate=0.0, shear=0.0):
        self._push(affine_matrix(self.h, self.w, scale, translate_percent, rotate, shear))

    def crop(self, top, left, new_h, new_w):
        """Keep the window [top, top+new_h) x [left, left+new_w): negative offsets pad with the constant border."""
        self._push(np.array([[1, 0, left], [0, 1, top], [0, 0, 1]], np.float64))
        self.h, self.w = float(new_h), float(new_w)

    def resize(self, new_h, new_w):
        sy, sx = self.h / new_h, self.w / new_w
        self._push(np.array([[sx, 0, 0.5 * sx - 0.5], [0, sy, 0.5 * sy - 0.5], [0, 0, 1]], np.float64))
        self.h, self.w = float(new_h), float(new_w)


def compose(h, w, steps, out_hw):
    """steps: [("fliplr",), ("flipud",), ("affine", scale, (tx, ty), rotate, shear), ("crop", top, left, nh, nw),
    ("resize", nh, nw)] applied in order, then the trailing Resize to ``out_hw``.  Returns the 3x3 output->input map."""
    c = Canvas(h, w)
    for st in steps:
        getattr(c, st[0])(*st[1:])
    if (int(c.h), int(c.w)) != tuple(out_hw):
        c.resize(*out_hw)
    return c.M


def field_piecewise(grid, h, w):
    """PiecewiseAffine displacement field (imgaug / skimage PiecewiseAffineTransform, schemas/augmenters.raml:126-129): control
    points at linspace(0, h, rows) x linspace(0, w, cols) moved by ``grid`` int [n, rows, cols, 2] = (dx, dy) in 1/64 pixel;
    affine on each of the two triangles the (0,0)-(1,1) diagonal cuts a cell into, cell position with 10 fractional bits.
    Returns int64 [n, h, w, 2]."""
    grid = np.asarray(grid, np.int64)
    n, R, Cg, _ = grid.shape
    x, y = np.arange(w, dtype=np.int64)[None, :], np.arange(h, dtype=np.int64)[:, None]
    u, v = x * (Cg - 1) * 1024 // w, y * (R - 1) * 1024 // h
    cx, cy = np.minimum(u >> 10, Cg - 2), np.minimum(v >> 10, R - 2)
    fu, fv = np.broadcast_to(u - (cx << 10), (h, w)), np.broadcast_to(v - (cy << 10), (h, w))
    cxb, cyb = np.broadcast_to(cx, (h, w)), np.broadcast_to(cy, (h, w))
    out = np.zeros((n, h, w, 2), np.int64)
    for i in range(n):
        g = grid[i]
        d00, d10, d01, d11 = g[cyb, cxb], g[cyb, cxb + 1], g[cyb + 1, cxb], g[cyb + 1, cxb + 1]
        lower = (fu >= fv)[..., None]
        t = np.where(lower, d00 * 1024 + fu[..., None] * (d10 - d00) + fv[..., None] * (d11 - d10),
                     d00 * 1024 + fv[..., None] * (d01 - d00) + fu[..., None] * (d11 - d01))
        out[i] = np.clip((t + 512) >> 10, -32768, 32767)
    return out


def background_replace_u8(img, mask, bg, erosion=0):
    """BackgroundReplacer (reference README.md:270-278): img [N,H,W,C], mask [N,H,W], bg [N,H,W,C] (already at the item's size) ->
    img inside the mask eroded by a (2e+1)^2 minimum (image border does not erode), bg elsewhere."""
    n, h, w = mask.shape
    fg = mask != 0
    er = fg.copy()
    for dy in range(-erosion, erosion + 1):
        for dx in range(-erosion, erosion + 1):
            sh = np.ones_like(fg)
            ys, ye, xs, xe = max(0, -dy), min(h, h - dy), max(0, -dx), min(w, w - dx)
            sh[:, ys:ye, xs:xe] = fg[:, ys + dy:ye + dy, xs + dx:xe + dx]
            er &= sh
    return np.where(er[..., None], img, bg).astype(np.uint8)


ELASTIC_RECORD = 69


def field_elastic(recs, h, w):
    """ElasticTransformation displacement field (imgaug: uniform(-1, 1) noise per pixel and axis, gaussian_filter(sigma,
    mode='constant'), times alpha; schemas/augmenters.raml:130-133).  recs int [n, 69]: seed, alpha (1/64 pixel), radius,
    0, one-sided weights in 1/32768.  The kernel's integer arithmetic: noise = 16 hash bits - 32768, two separable passes
    each rounded to 1/32768, displacement = (alpha * v + 2^14) >> 15.  Returns int64 [n, h, w, 2]."""
    recs = np.asarray(recs, np.int64)
    n = recs.shape[0]
    out = np.zeros((n, h, w, 2), np.int64)
    pix = (np.arange(h, dtype=np.int64)[:, None] * w + np.arange(w, dtype=np.int64)[None, :])
    for i in range(n):
        seed, alpha, r = int(recs[i, 0]), int(recs[i, 1]), int(recs[i, 2])
        wk = recs[i, 4:4 + r + 1]
        for a in range(2):
            nz = (aug_hash(seed, pix, np.asarray(a), 7).astype(np.int64) >> 16) - 32768
            pad = np.zeros((h, w + 2 * r), np.int64)
            pad[:, r:r + w] = nz
            s = sum(int(wk[abs(k)]) * pad[:, r + k:r + k + w] for k in range(-r, r + 1))
            t = np.clip((s + 16384) >> 15, -32768, 32767)
            pad = np.zeros((h + 2 * r, w), np.int64)
            pad[r:r + h] = t
            s = sum(int(wk[abs(k)]) * pad[r + k:r + k + h] for k in range(-r, r + 1))
            v = (s + 16384) >> 15
            out[i, :, :, a] = np.clip((alpha * v + 16384) >> 15, -32768, 32767)
    return out


def warp_u8(img, mask, params, out_hw, field=None):
    """img [N,H,W,C] u8, mask [N,H,W] u8 or None, params float32 [N,24] -> (img_out, mask_out).  ``field``: int [N,oh,ow,2]
    displacement (dx, dy) in 1/64 pixel added to the output pixel position before the matrix (field_piecewise / field_elastic)."""
    n, h, w, c = img.shape
    oh, ow = out_hw
    img_out = np.zeros((n, oh, ow, c), np.uint8)
    mask_out = None if mask is None else np.zeros((n, oh, ow), np.uint8)
    for i in range(n):
        xo = np.arange(ow, dtype=np.float64)[None, :]
        yo = np.arange(oh, dtype=np.float64)[:, None]
        if field is not None:
            xo = xo + field[i, :, :, 0].astype(np.float64) * 0.015625
            yo = yo + field[i, :, :, 1].astype(np.float64) * 0.015625
        m = params[i, :6].astype(np.float64)
        X0 = np.rint(m[0] * xo * 1024.0).astype(np.int64) + np.rint((m[1] * yo + m[2]) * 1024.0).astype(np.int64)
        Y0 = np.rint(m[3] * xo * 1024.0).astype(np.int64) + np.rint((m[4] * yo + m[5]) * 1024.0).astype(np.int64)
        X, Y = (X0 + 16) >> 5, (Y0 + 16) >> 5
        ix, iy = X >> 5, Y >> 5
        fx, fy = (X & 31), (Y & 31)
        w00, w01, w10, w11 = (32 - fx) * (32 - fy), fx * (32 - fy), (32 - fx) * fy, fx * fy

        def tap(yy, xx):
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            v = img[i][np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64)
            return v * ok[..., None]

        v = (w00[..., None] * tap(iy, ix) + w01[..., None] * tap(iy, ix + 1) +
             w10[..., None] * tap(iy + 1, ix) + w11[..., None] * tap(iy + 1, ix + 1) + 512) >> 10
        img_out[i] = point_ops(v, params[i])
        if mask is not None:
            mx, my = (X0 + 512) >> 10, (Y0 + 512) >> 10
            ok = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
            mask_out[i] = mask[i][np.clip(my, 0, h - 1), np.clip(mx, 0, w - 1)] * ok
    return img_out, mask_out


FILTER_RECORD = 173      # int32 per image: K, mode, 0, 0, 13*13 weights (include/stp_hip.h, stp_filter_u8)


def filter_u8(img, recs):
    """img [N,H,W,C] u8, recs int32 [N,173] -> filtered copy; bit-exact restatement of stp_filter_u8."""
    out = img.copy()
    n, h, w, c = img.shape
    for i in range(n):
        K, mode = int(recs[i, 0]), int(recs[i, 1])
        if K <= 0:
            continue
        r = K // 2
        pad = np.pad(img[i].astype(np.int64), ((r, r), (r, r), (0, 0)), mode="reflect")    # numpy's reflect == reflect-101
        win = np.stack([pad[ky:ky + h, kx:kx + w] for ky in range(K) for kx in range(K)], axis=0)   # [K*K,H,W,C]
        if mode == 0:
            wts = recs[i, 4:4 + K * K].astype(np.int64)
            acc = (win * wts[:, None, None, None]).sum(axis=0)
            out[i] = np.clip((acc + 8192) >> 14, 0, 255).astype(np.uint8)
        else:
            out[i] = np.sort(win, axis=0)[(K * K) // 2].astype(np.uint8)
    return out
