"""Run-length mask codec with the reference's conventions (reference
``segmentation_pipeline/impl/rle.py:10-35``): pixels are numbered 1-based in COLUMN-major
order, the string is ``"start length start length ..."``, and ``rle_decode(s, shape)``
returns ``reshape(shape).T``.  Pinned by ``tests/golden/rle_golden.json`` (vectors produced by
the reference's own module).  Connected-component splitting uses scipy.ndimage instead of
the reference's skimage dependency (``rle.py:1,4-6``).
"""
import numpy as np


def rle_encode(img):
    """img: 2-D array, non-zero = mask.  Returns the run-length string."""
    flat = (np.asarray(img).T.reshape(-1) != 0).astype(np.int8)
    if flat.size == 0:
        return ""
    edges = np.flatnonzero(np.diff(np.concatenate(([0], flat, [0])))) + 1
    starts, stops = edges[0::2], edges[1::2]
    return " ".join("%d %d" % (s, e - s) for s, e in zip(starts, stops))


def rle_decode(mask_rle, shape):
    """Inverse of :func:`rle_encode`; returns uint8 array of shape ``shape[::-1]`` transposed
    the way the reference does (``reshape(shape).T``)."""
    tok = np.asarray(mask_rle.split(), dtype=np.int64)
    starts, lengths = tok[0::2] - 1, tok[1::2]
    flat = np.zeros(int(shape[0]) * int(shape[1]), np.uint8)
    for s, l in zip(starts, lengths):
        flat[s:s + l] = 1
    return flat.reshape(shape).T


def multi_rle_encode(img):
    """One RLE string per connected component of ``img[:, :, 0]`` (reference rle.py:4-6
    uses skimage.morphology.label, default full connectivity for 2-D = 8-neighbourhood)."""
    from scipy import ndimage
    labels, n = ndimage.label(np.asarray(img)[:, :, 0] != 0, structure=np.ones((3, 3), int))
    return [rle_encode(labels == k) for k in range(1, n + 1)]


def masks_as_image(in_mask_list, shape):
    """Sum of the decoded masks, ``[H', W', 1]`` int16 (reference rle.py:38-44)."""
    total = np.zeros(shape, np.int16)
    for m in in_mask_list:
        if isinstance(m, str):
            total += rle_decode(m, shape)
    return total[..., None]


def masks_as_images(in_mask_list, shape):
    """List of decoded float32 masks (reference rle.py:46-53)."""
    return [rle_decode(m, shape).astype(np.float32) for m in in_mask_list if isinstance(m, str)]
