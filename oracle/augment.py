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


def pack_params(mats, adds, muls):
    """float32 [N,10] rows: m00 m01 m02 m10 m11 m12 add mul 0 0 (the kernel's per-sample record)."""
    n = len(mats)
    out = np.zeros((n, 10), np.float32)
    for i in range(n):
        out[i, 0:3] = mats[i][0]
        out[i, 3:6] = mats[i][1]
        out[i, 6] = adds[i]
        out[i, 7] = muls[i]
    return out


def warp_u8(img, mask, params, out_hw):
    """img [N,H,W,C] u8, mask [N,H,W] u8 or None, params float32 [N,10] -> (img_out, mask_out)."""
    n, h, w, c = img.shape
    oh, ow = out_hw
    xo = np.arange(ow, dtype=np.float64)[None, :]
    yo = np.arange(oh, dtype=np.float64)[:, None]
    img_out = np.zeros((n, oh, ow, c), np.uint8)
    mask_out = None if mask is None else np.zeros((n, oh, ow), np.uint8)
    for i in range(n):
        m = params[i, :6].astype(np.float64)
        add, mul = int(params[i, 6]), np.float32(params[i, 7])
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
        v = np.clip(v + add, 0, 255)
        if mul != np.float32(1.0):
            v = np.clip(np.rint(v.astype(np.float32) * mul).astype(np.int64), 0, 255)
        img_out[i] = v.astype(np.uint8)
        if mask is not None:
            mx, my = (X0 + 512) >> 10, (Y0 + 512) >> 10
            ok = (mx >= 0) & (mx < w) & (my >= 0) & (my < h)
            mask_out[i] = mask[i][np.clip(my, 0, h - 1), np.clip(mx, 0, w - 1)] * ok
    return img_out, mask_out
