"""Dataset classes under the reference's import path (reference ``segmentation_pipeline/impl/datasets.py:1``
re-exports ``musket_core.datasets``; usage README.md:121-125, 313-340, 417-427).

Contract reproduced here: a dataset is any object with ``__len__`` and ``__getitem__(i) ->
PredictionItem(id, x, y)`` where ``x`` is an H x W x 3 uint8 image (0..255) and ``y`` an H x W x 1 mask in
{0,1}; ``isPositive(i)`` is optional (``negatives:`` balancing).  Images are decoded on the host with
Pillow; everything after decoding (resize, augmentation, normalisation) happens on the GPU.
"""
import os

import numpy as np

AUGMENTER_QUEUE_LIMIT = 50   # kept for source compatibility (FAQ.md:15-22); the HIP path has no worker queue

_IMG_EXT = (".jpg", ".jpeg", ".png", ".bmp", ".gif", ".tif", ".tiff")


class PredictionItem(object):
    def __init__(self, path, x, y):
        self.x = x
        self.y = y
        self.id = path

    def original(self):
        return self

    def rootItem(self):
        return self


class DataSet(object):
    def __len__(self):
        raise NotImplementedError

    def __getitem__(self, item):
        raise NotImplementedError

    def isPositive(self, item):
        return True


def _imread_rgb(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def _imread_mask(path):
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im)
    if a.ndim == 3:                      # colour-coded binary mask: any non-zero channel is foreground
        return (a.astype(np.int32).sum(axis=2) != 0).astype(np.uint8)[:, :, None]
    # single-channel PNG: the usual 0/255 (or 0/1) binary mask -> {0,1}; a label image whose values are small class
    # indices (<= 32, `classes` > 1 experiments) keeps them - the training feeder binarises for the 1-class head anyway
    if a.max(initial=0) > 32:
        return (a != 0).astype(np.uint8)[:, :, None]
    return a.astype(np.uint8)[:, :, None]


class SimplePNGMaskDataSet(DataSet):
    """Folder of images + folder of same-named PNG masks (README.md:116-125)."""

    def __init__(self, path, mask, in_ext="jpg", out_ext="png", generate=False):
        self.path, self.mask = path, mask
        self.in_ext, self.out_ext = in_ext, out_ext
        names = sorted(f for f in os.listdir(path) if f.lower().endswith(_IMG_EXT))
        self.files = {os.path.splitext(f)[0]: f for f in names}
        self.ids = sorted(self.files)
        self.name = os.path.basename(os.path.normpath(path))

    def __len__(self):
        return len(self.ids)

    def _mask_path(self, ident):
        for ext in (self.out_ext, "png", "PNG"):
            p = os.path.join(self.mask, ident + "." + ext)
            if os.path.exists(p):
                return p
        raise FileNotFoundError("no mask for %r in %s" % (ident, self.mask))

    def __getitem__(self, item):
        ident = self.ids[item]
        x = _imread_rgb(os.path.join(self.path, self.files[ident]))
        y = _imread_mask(self._mask_path(ident))
        return PredictionItem(ident, x, y)

    def isPositive(self, item):
        return True


class NotzeroSimplePNGMaskDataSet(SimplePNGMaskDataSet):
    """Variant whose ``isPositive`` looks at the mask (FAQ.md:100-106 uses this name for extra data)."""

    def isPositive(self, item):
        return bool(_imread_mask(self._mask_path(self.ids[item])).any())


class DirectoryDataSet(DataSet):
    """Images of a folder without masks (prediction input)."""

    def __init__(self, path):
        self.path = path
        self.ids = sorted(f for f in os.listdir(path) if f.lower().endswith(_IMG_EXT))

    def __len__(self):
        return len(self.ids)

    def __getitem__(self, item):
        f = self.ids[item]
        return PredictionItem(f, _imread_rgb(os.path.join(self.path, f)), None)


# ------------------------------------------------------------------------------------------ writeable datasets
# musket_core.datasets (un-vendored; imported by reference segmentation.py:12) supplies WriteableDataSet / DirectWriteableDS /
# CompressibleWriteableDS.  What is reproduced is what the reference's call sites need (segmentation.py:196-208): a dataset of
# per-item PREDICTIONS that lives in a folder, is filled with ``append`` + ``commit`` and read back as
# ``PredictionItem(id, parent[i].x, prediction)``; the compressible form stores ``round-down(p * scale)`` as uint8 (scale <= 255)
# or uint16 inside a compressed .npz and divides on load (``compressPredictionsAsInts`` / ``compressScale`` of the YAML).
class WriteableDataSet(DataSet):
    def append(self, item):
        raise NotImplementedError

    def commit(self):
        raise NotImplementedError


class DirectWriteableDS(WriteableDataSet):
    """One file per item under ``dsPath`` (``<index>.npy``); ``count`` = items already present (re-opening a folder)."""

    def __init__(self, orig, name, dsPath, count=0):
        self.parent, self.name, self.dsPath, self.count = orig, name, dsPath, int(count)
        os.makedirs(dsPath, exist_ok=True)

    def item_path(self, i):
        return os.path.join(self.dsPath, str(int(i)))

    def saveItem(self, path, item):
        np.save(path + ".npy", np.asarray(item))

    def loadItem(self, path):
        return np.load(path + ".npy")

    def append(self, item):
        self.saveItem(self.item_path(self.count), item)
        self.count += 1

    def commit(self):
        return self

    def __len__(self):
        return self.count

    def __getitem__(self, item):
        i = int(item)
        if i < 0 or i >= self.count:
            raise IndexError(i)
        src = self.parent[i] if self.parent is not None else None
        return PredictionItem(src.id if src is not None else i, src.x if src is not None else None, self.loadItem(self.item_path(i)))

    def isPositive(self, item):
        return self.parent.isPositive(item) if self.parent is not None and hasattr(self.parent, "isPositive") else True


class CompressibleWriteableDS(DirectWriteableDS):
    def __init__(self, orig, name, dsPath, count=0, asUints=True, scale=255):
        super().__init__(orig, name, dsPath, count)
        self.asUints, self.scale = bool(asUints), scale

    def saveItem(self, path, item):
        a = np.asarray(item)
        if self.asUints:
            a = (a * self.scale).astype(np.uint8 if self.scale <= 255 else np.uint16)
        np.savez_compressed(path + ".npy.npz", arr=a)

    def loadItem(self, path):
        with np.load(path + ".npy.npz") as z:
            a = z["arr"]
        return a.astype(np.float32) / self.scale if self.asUints else a
