"""Host-side training engine behind ``cfg.fit()``: the counterpart of the un-vendored
``musket_core.generic_config.GenericImageTaskConfig`` whose methods the reference calls at
``segmentation_pipeline/segmentation.py:35-47,65,131,151,159-169`` and documents in README.md
(folds ``:172-175``, ``testSplit`` ``:196``, stages ``:353-383``, negatives ``:385-427``, freeze /
unfreeze ``:281-299``, callbacks ``:721-743``, resume ``:651-660``, ``cfg.gpus`` ``:756-760``).

What runs where: this module only schedules - k-fold split, stage loop, epoch loop, callbacks,
checkpoints and metric files (paths next to the YAML, README.md:173-174, ``weights/best-<fold>.<stage>.weights``
README.md:382).  Every pixel and every FLOP is handled by the HIP plan (``backend.HipSegModel``): images
are uploaded as raw uint8, resized/augmented by ``stp_augment_u8`` and trained by the captured graph.
"""
import csv
import os
import time

import numpy as np
import torch
import yaml

from . import augment, distributed, ops

# YAML keys flagged (meta.custom) in schemas/segmentation.raml:26-120 (+ the keys used by the examples that
# the RAML does not list): consumed by the pipeline, never forwarded to the model constructor.
CUSTOM_KEYS = {
    "architecture", "crops", "augmentation", "transforms", "optimizer", "lr", "clipnorm", "clipvalue", "loss", "batch",
    "metrics", "primary_metric", "primary_metric_mode", "callbacks", "stages", "folds_count", "random_state",
    "extra_train_data", "dataset_augmenter", "classifier", "classifier_lr", "testSplit", "dataset", "datasets", "fit_with",
    "imports", "import_tasks", "run_tasks", "copyWeights", "dtype", "loss_scale", "gpus", "inference_batch", "testTimeAugmentation",
    "compressPredictionsAsInts", "compressScale", "showDataExamples", "bgr", "stratified", "validationSplit", "draw_examples",
}
# (meta.alias) renames, schemas/segmentation.raml:50-51,67-68,175-176
ALIASES = {"backbone": "backbone_name", "shape": "input_shape", "use_batchnorm": "decoder_use_batchnorm"}


def load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f) or {}


def aug_list(spec):
    """``augmentation:`` may be a mapping (README.md:130-133) or a list of single-key mappings."""
    if spec is None:
        return []
    if isinstance(spec, dict):
        return [{k: v} for k, v in spec.items()]
    return list(spec)


def metric_mode(name, mode="auto"):
    if mode in ("min", "max"):
        return mode
    return "min" if "loss" in name else "max"


# Keras logs a metric under the NAME OF ITS FUNCTION: the registry of segmentation.py:15-22 maps `iou` -> iou_coef,
# `iot` -> iot_coef, `dice_loss` -> dice_coef_loss, so reference YAMLs monitor e.g. `val_iou_coef`
# (examples/people/ds_1.yaml:19-27).  Both spellings resolve to the same log entry.
LOG_ALIASES = {"iou_coef": "iou", "iot_coef": "iot", "dice_coef_loss": "dice_loss", "iou_coef_loss": "iou_loss", "acc": "binary_accuracy"}


def log_value(logs, name):
    if name in logs:
        return logs[name]
    pre, base = ("val_", name[4:]) if name.startswith("val_") else ("", name)
    return logs.get(pre + LOG_ALIASES.get(base, base))


# ------------------------------------------------------------------------------------------ callbacks
class EarlyStopping(object):
    """keras.callbacks.EarlyStopping subset (schemas/callbacks.raml:8-23)."""

    def __init__(self, patience=0, monitor="val_loss", mode="auto", verbose=0, min_delta=0.0):
        self.patience, self.monitor, self.mode = int(patience), monitor, metric_mode(monitor, mode)
        self.best, self.wait, self.stop = None, 0, False

    def on_epoch_end(self, trainer, epoch, logs):
        v = log_value(logs, self.monitor)
        if v is None:
            return
        if self.best is None or (v < self.best if self.mode == "min" else v > self.best):
            self.best, self.wait = v, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stop = True


class ReduceLROnPlateau(object):
    """keras.callbacks.ReduceLROnPlateau subset (schemas/callbacks.raml:24-33)."""

    def __init__(self, patience=10, factor=0.1, monitor="val_loss", mode="auto", cooldown=0, verbose=0, min_lr=0.0):
        self.patience, self.factor, self.monitor = int(patience), float(factor), monitor
        self.mode, self.cooldown, self.min_lr = metric_mode(monitor, mode), int(cooldown), float(min_lr)
        self.best, self.wait, self.cool = None, 0, 0
        self.stop = False

    def on_epoch_end(self, trainer, epoch, logs):
        v = log_value(logs, self.monitor)
        if v is None:
            return
        if self.cool > 0:
            self.cool -= 1
            self.wait = 0
        if self.best is None or (v < self.best if self.mode == "min" else v > self.best):
            self.best, self.wait = v, 0
        elif self.cool <= 0:
            self.wait += 1
            if self.wait >= self.patience:
                old = trainer.model.get_lr()
                if old > self.min_lr:
                    trainer.model.set_lr(max(old * self.factor, self.min_lr))
                    self.cool, self.wait = self.cooldown, 0


class CyclicLR(object):
    """bckenstler CLR (schemas/callbacks.raml:34-48, README.md:433-452): per-batch triangular policy."""

    def __init__(self, base_lr=0.001, max_lr=0.006, step_size=2000.0, mode="triangular", gamma=1.0):
        self.base_lr, self.max_lr, self.step_size, self.mode, self.gamma = float(base_lr), float(max_lr), float(step_size), mode, float(gamma)
        self.it = 0
        self.stop = False

    def lr(self):
        cycle = np.floor(1 + self.it / (2 * self.step_size))
        x = abs(self.it / self.step_size - 2 * cycle + 1)
        scale = 1.0 if self.mode == "triangular" else (1 / (2.0 ** (cycle - 1)) if self.mode == "triangular2" else self.gamma ** self.it)
        return self.base_lr + (self.max_lr - self.base_lr) * max(0.0, 1 - x) * scale

    def on_batch_end(self, trainer):
        self.it += 1
        trainer.model.set_lr(self.lr())

    def on_epoch_end(self, trainer, epoch, logs):
        pass


CALLBACKS = {"EarlyStopping": EarlyStopping, "ReduceLROnPlateau": ReduceLROnPlateau, "CyclicLR": CyclicLR}


class LRFinder(object):
    """Result of ``cfg.lr_find`` with the accessors of keras_lr_finder (README.md:461-468): ``lrs`` / ``losses`` per batch,
    ``plot_loss`` and ``plot_loss_change`` (matplotlib, imported on use), ``get_derivatives``."""

    def __init__(self, start_lr, end_lr, num_batches):
        self.lrs, self.losses = [], []
        self.best_loss = 1e9
        self.mult = (end_lr / start_lr) ** (1.0 / max(1, num_batches))

    def record(self, lr, loss):
        """Appends one batch; True = stop (loss diverged: NaN or > 4x the best so far)."""
        self.lrs.append(lr)
        self.losses.append(loss)
        if not np.isfinite(loss) or loss > self.best_loss * 4:
            return True
        self.best_loss = min(self.best_loss, loss)
        return False

    def get_derivatives(self, sma=1):
        d = [0.0] * sma
        for i in range(sma, len(self.lrs)):
            d.append((self.losses[i] - self.losses[i - sma]) / sma)
        return d

    def plot_loss(self, n_skip_beginning=10, n_skip_end=5, x_scale="log"):
        import matplotlib.pyplot as plt
        plt.ylabel("loss"); plt.xlabel("learning rate (log scale)")
        plt.plot(self.lrs[n_skip_beginning:len(self.lrs) - n_skip_end], self.losses[n_skip_beginning:len(self.losses) - n_skip_end])
        plt.xscale(x_scale)

    def plot_loss_change(self, sma=1, n_skip_beginning=10, n_skip_end=5, y_lim=(-0.01, 0.01)):
        import matplotlib.pyplot as plt
        d = self.get_derivatives(sma)[n_skip_beginning:len(self.lrs) - n_skip_end]
        plt.ylabel("rate of loss change"); plt.xlabel("learning rate (log scale)")
        plt.plot(self.lrs[n_skip_beginning:len(self.lrs) - n_skip_end], d)
        plt.xscale("log"); plt.ylim(y_lim)


class DrawResults(object):
    """Per-epoch example sheets (reference segmentation.py:216-247, 251-257): up to ``limit`` samples of the fold's
    validation set (or, with ``train=True`` = ``cfg.showDataExamples``, augmented training samples) go through
    ``model.predict``; image | ground truth | prediction > 0.5 are written side by side to
    ``examples/<stage>/<fold>/t_epoch_<epoch>.<n>.jpg`` (``t_epoch_train<epoch>.<n>.jpg`` for the training variant)."""

    def __init__(self, cfg, ds, indexes, fold, stage, limit=16, train=False, drawingFunction=None):
        self.cfg, self.ds, self.fold, self.stage, self.train = cfg, ds, fold, stage, train
        self.indexes = [int(i) for i in list(indexes)[:limit]]
        self.drawingFunction = drawingFunction or draw_test_batch
        self.stop = False

    def on_epoch_end(self, trainer, epoch, logs=None):
        if trainer.rank != 0 or not self.indexes:
            return
        m = trainer.model
        items = [self.ds[i] for i in self.indexes]
        B = m.batch
        dr = os.path.join(os.path.dirname(os.path.abspath(self.cfg.path)), "examples", str(self.stage), str(self.fold))
        os.makedirs(dr, exist_ok=True)
        for num, s in enumerate(range(0, len(items), B)):
            chunk = items[s:s + B]
            ip = m.eval_plan()
            trainer.feeder.feed(ip, chunk, training=self.train)       # resize only, or the training augmentation
            xs = ip.inputs["image"].buf[:len(chunk)].cpu().numpy()
            ys = ip.inputs["mask"].buf[:len(chunk)].cpu().numpy()
            pred = m.predict(xs)
            name = ("t_epoch_train" if self.train else "t_epoch_") + str(epoch) + "." + str(num) + ".jpg"
            self.drawingFunction(EvalSheet(xs, ys, pred > 0.5), os.path.join(dr, name))


class EvalSheet(object):
    """What a drawing function receives: network-shape ``images_aug`` (uint8), ``segmentation_maps_aug`` (ground truth) and
    ``heatmaps_aug`` (thresholded predictions), the attribute names of the imgaug batch the reference passes."""

    def __init__(self, images, masks, heat):
        self.images_aug, self.segmentation_maps_aug, self.heatmaps_aug = images, masks, heat


def draw_test_batch(batch, path):
    """Default drawing function: one row per sample, image | ground truth | prediction (class maps scaled to 0..255)."""
    from PIL import Image
    rows = []
    for x, y, p in zip(batch.images_aug, batch.segmentation_maps_aug, batch.heatmaps_aug):
        y = np.asarray(y).reshape(x.shape[0], x.shape[1], -1)
        p = np.asarray(p).reshape(x.shape[0], x.shape[1], -1)
        gt = (y[:, :, 0].astype(np.float32) * (255.0 / max(1, int(y.max())))).astype(np.uint8)
        pr = ((p.argmax(axis=2) * (255 // max(1, p.shape[2] - 1))) if p.shape[2] > 1 else p[:, :, 0] * 255).astype(np.uint8)
        rows.append(np.concatenate([x[:, :, :3], np.repeat(gt[:, :, None], 3, 2), np.repeat(pr[:, :, None], 3, 2)], axis=1))
    Image.fromarray(np.concatenate(rows, axis=0)).save(path, quality=90)


def make_callbacks(spec):
    out = []
    for item in aug_list(spec):
        (name, args), = item.items()
        if name not in CALLBACKS:
            raise ValueError("callback %r is not available (have: %s)" % (name, ", ".join(sorted(CALLBACKS))))
        out.append(CALLBACKS[name](**(args or {})))
    return out


# name -> dataset whose samples are added to the TRAINING indexes of every fold and never to validation
# (README.md:698-709: ``segmentation.extra_train["people"] = ds`` + ``extra_train_data: people`` in the YAML)
extra_train = {}


class ConcatDataSet(object):
    """Items of ``a`` followed by the items of ``b`` (the fold split runs over ``a`` only; see extra_train)."""

    def __init__(self, a, b):
        self.a, self.b, self.na = a, b, len(a)
        self.name = getattr(a, "name", "")

    def __len__(self):
        return self.na + len(self.b)

    def __getitem__(self, i):
        i = int(i)
        return self.a[i] if i < self.na else self.b[i - self.na]

    def isPositive(self, i):
        i = int(i)
        d, j = (self.a, i) if i < self.na else (self.b, i - self.na)
        return d.isPositive(j) if hasattr(d, "isPositive") else True


# ------------------------------------------------------------------------------------------ folds
class KFoldedDataSet(object):
    """Train/validation index sets per fold from a fixed seed (README.md:172-175); optional hold-out
    ``testSplit`` (README.md:196).  ``negatives`` handling follows README.md:385-427."""

    def __init__(self, ds, indexes, folds_count=5, random_state=33, test_split=0.0):
        self.ds = ds
        idx = np.array(list(indexes), dtype=np.int64)
        rng = np.random.RandomState(random_state)
        perm = rng.permutation(len(idx))
        idx = idx[perm]
        self.test_indexes = np.array([], np.int64)
        if test_split and test_split > 0:
            n_test = int(round(len(idx) * float(test_split)))
            self.test_indexes, idx = idx[:n_test], idx[n_test:]
        self.folds = []
        if folds_count <= 1 or len(idx) < 2:
            n_val = max(1, len(idx) // 5) if len(idx) > 1 else 0
            self.folds.append((idx[n_val:], idx[:n_val]))
        else:
            parts = np.array_split(idx, folds_count)
            for f in range(folds_count):
                val = parts[f]
                train = np.concatenate([parts[j] for j in range(folds_count) if j != f]) if folds_count > 1 else idx
                self.folds.append((train, val))

    extra_train_indexes = ()     # appended to the training indexes of every fold (extra_train_data)

    def sampledIndexes(self, fold, isTrain, negatives="all"):
        idx = self.folds[fold][0 if isTrain else 1]
        if isTrain and len(self.extra_train_indexes):
            idx = np.concatenate([idx, np.asarray(self.extra_train_indexes, np.int64)])
        if negatives in ("all", "real", None):
            return idx
        pos = [i for i in idx if self.ds.isPositive(int(i))]
        if negatives == "none":
            return np.array(pos, np.int64)
        neg = [i for i in idx if not self.ds.isPositive(int(i))]
        k = int(negatives) * len(pos)
        return np.array(pos + neg[:k], np.int64)


class ItemBatch(object):
    """Stand-in for the ``imgaug.Batch`` that ``folds.load`` returns in the reference (musket_core, un-vendored; call sites
    segmentation.py:41,228): ``images`` (original uint8 arrays), ``segmentation_maps`` (ground-truth masks), ``data`` (ids); an
    augmentor adds ``images_aug`` / ``segmentation_maps_aug`` at the network shape, ``evaluate`` adds ``heatmaps_aug``."""

    def __init__(self, images, segmentation_maps, data, items=None):
        self.images, self.segmentation_maps, self.data, self.items = images, segmentation_maps, data, items
        self.images_aug = self.segmentation_maps_aug = self.heatmaps_aug = None


def _folds_load(self, fold, isTrain, negatives="all", limit=16):
    """Up to ``limit`` items of the fold's train / validation indexes as one batch (reference segmentation.py:41, 228)."""
    idx = [int(i) for i in list(self.sampledIndexes(fold, isTrain, negatives))[:limit]]
    items = [self.ds[i] for i in idx]
    return ItemBatch([it.x for it in items], [it.y for it in items], [it.id for it in items], items)


def _folds_augmentor(self, isTrain=True):
    """The fold's augmentor (reference segmentation.py:223 ``folds.augmentor(isTrain=True)``): set by ``cfg.kfold`` - the training
    pipeline ``augmentation`` + ``transforms`` + Resize, or the validation one."""
    if getattr(self, "cfg", None) is None:
        raise ValueError("this fold set was not created by cfg.kfold(): no augmentation pipeline is attached")
    return TransformAugmentor(self.cfg, train=bool(isTrain))


KFoldedDataSet.load = _folds_load
KFoldedDataSet.augmentor = _folds_augmentor


class _FeedTarget(object):
    """The two input buffers a DeviceFeeder fills (what it reads of a plan)."""

    class _T(object):
        def __init__(self, buf):
            self.buf = buf

    def __init__(self, n, H, W, ch, device):
        self.inputs = {"image": self._T(torch.empty((n, H, W, ch), dtype=torch.uint8, device=device)),
                       "mask": self._T(torch.empty((n, H, W, 1), dtype=torch.uint8, device=device))}


class TransformAugmentor(object):
    """``cfg.transformAugmentor()`` (reference segmentation.py:39, 224): ``transforms`` + Resize to the network shape, applied jointly
    to images and masks - on the device (``stp_augment_u8``), like every other resize of this backend.  ``augment_batches`` takes
    ItemBatch objects (``folds.load``) and yields them with ``images_aug`` uint8 [n, H, W, C] and ``segmentation_maps_aug``
    uint8 [n, H, W, 1] filled in.  ``train=True``: the training pipeline (``augmentation`` as well)."""

    def __init__(self, cfg, train=False, seed=None):
        self.cfg, self.train = cfg, bool(train)
        s = cfg.shape
        c = int(cfg.crops) if cfg.crops else 1
        self.H, self.W, self.ch = int(s[0]) // c, int(s[1]) // c, int(s[2]) if len(s) > 2 else 3
        _, local_rank, _ = distributed.env_world()
        self.device = "cuda:%d" % distributed.device_index(local_rank)
        spec = cfg._aug_spec() if self.train else augment.resolve_paths(
            cfg.transforms, os.path.dirname(os.path.abspath(cfg.path)) if cfg.path else None)
        self.feeder = DeviceFeeder(self.device, (self.H, self.W), spec, seed=cfg.random_state if seed is None else seed,
                                   classes=cfg.classes, channels=self.ch)

    def augment_batches(self, batches):
        for b in batches:
            items = b.items if getattr(b, "items", None) is not None else [
                _Item(i, x, y) for i, x, y in zip(b.data, b.images, b.segmentation_maps)]
            n = len(items)
            if n == 0:
                b.images_aug = np.zeros((0, self.H, self.W, self.ch), np.uint8)
                b.segmentation_maps_aug = np.zeros((0, self.H, self.W, 1), np.uint8)
                yield b
                continue
            tgt = _FeedTarget(n, self.H, self.W, self.ch, self.device)
            # validation = Resize only (+ `transforms`): the feeder's `training` flag selects whether its spec is sampled at all
            self.feeder.feed(tgt, items, training=True if (self.train or self.feeder.spec) else False)
            torch.cuda.synchronize(self.device)
            b.images_aug = tgt.inputs["image"].buf.cpu().numpy()
            b.segmentation_maps_aug = tgt.inputs["mask"].buf.cpu().numpy()
            yield b


class _Item(object):
    def __init__(self, ident, x, y):
        self.id, self.x, self.y = ident, x, y


def crop_bounds(size, crops):
    """Cell boundaries along one axis for ``crops`` cells (README.md:476-491): floor(k * size / crops)."""
    return [(k * size) // crops for k in range(crops + 1)]


class CropsDataSet(object):
    """``crops: N`` (README.md:476-491): every image / mask is split into N x N cells and the model trains on the cells
    (augmentations run per cell).  Item i is cell ``i % N^2`` (row-major) of base item ``i // N^2``."""

    def __init__(self, ds, crops):
        self.ds, self.crops = ds, int(crops)
        self.name = getattr(ds, "name", "")

    def __len__(self):
        return len(self.ds) * self.crops * self.crops

    def _cell(self, a, k):
        if a is None:
            return None
        a = np.asarray(a)
        ys, xs = crop_bounds(a.shape[0], self.crops), crop_bounds(a.shape[1], self.crops)
        r, c = divmod(k, self.crops)
        return a[ys[r]:ys[r + 1], xs[c]:xs[c + 1]]

    def __getitem__(self, i):
        from segmentation_pipeline.impl.datasets import PredictionItem
        base, k = divmod(int(i), self.crops * self.crops)
        it = self.ds[base]
        return PredictionItem("%s.%d" % (it.id, k), self._cell(it.x, k), self._cell(it.y, k))

    def isPositive(self, i):
        y = self[i].y
        return bool(np.asarray(y).any()) if y is not None else True


# ------------------------------------------------------------------------------------------ device feeding
class HostItem(object):
    """A dataset item prepared for the device: raw uint8 pixels and the label plane in PINNED host memory."""
    __slots__ = ("id", "x", "y", "h", "w", "src")

    def __init__(self, ident, x, y, h, w, src):
        self.id, self.x, self.y, self.h, self.w, self.src = ident, x, y, h, w, src


def prepare_item(it, classes, pin, channels=3):
    """PredictionItem -> HostItem: uint8 pixels [h,w,channels] (RGB, or the first ``channels`` <= 7 bands of an N-channel image:
    ``shape: [H, W, C]`` in the YAML, reference segmentation.py:135-155) and label uint8 [h,w] ({0,1} for the sigmoid head, class
    index for the softmax head; one-hot maps are arg-maxed).  This is the CPU work per sample; everything else happens on the GPU."""
    x = np.asarray(it.x)
    if x.ndim == 2:
        x = x[:, :, None]
    if x.shape[2] < channels:
        if x.shape[2] != 1:
            raise ValueError("item %r has %d channels, the network expects %d" % (it.id, x.shape[2], channels))
        x = np.repeat(x, channels, axis=2)     # grey image for a multi-channel network
    x = np.ascontiguousarray(x[:, :, :channels], dtype=np.uint8)
    if not x.flags.writeable:              # PIL-backed arrays are read-only; torch.from_numpy wants a writable buffer
        x = x.copy()
    h, w = x.shape[:2]
    y = it.y if it.y is not None else np.zeros((h, w, 1), np.uint8)
    y = np.asarray(y).reshape(h, w, -1)
    if classes == 1:
        y = (y[:, :, 0] != 0).astype(np.uint8)
    elif y.shape[2] == classes:            # one-hot maps (what a Keras softmax head is fed) -> class index
        y = y.argmax(axis=2).astype(np.uint8)
    else:                                  # label image
        y = np.minimum(y[:, :, 0], classes - 1).astype(np.uint8)
    xt, yt = torch.from_numpy(x), torch.from_numpy(np.ascontiguousarray(y))
    if pin:
        xt, yt = xt.pin_memory(), yt.pin_memory()
    return HostItem(it.id, xt, yt, h, w, it)


class HostBatch(object):
    """A full plan batch of equally sized items, already wrapped around to the plan's batch size: pinned pixel / label
    blocks and the sampled augmentation passes (augment.sample_batch_staged: ``passes`` for the whole batch, or ``per_item``
    when images differ in pass structure) - one H2D copy per block and, for one-pass pipelines, one kernel launch."""
    __slots__ = ("items", "X", "Y", "passes", "per_item", "h", "w")

    def __init__(self, items, X, Y, passes, per_item, h, w):
        self.items, self.X, self.Y, self.passes, self.per_item, self.h, self.w = items, X, Y, passes, per_item, h, w

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


class HostPrefetcher(object):
    """Background thread that reads (decodes) the next batches from the dataset and stages them in pinned memory while the
    GPU trains on the current one - the replacement of the reference's imgaug worker processes + bounded queue
    (FAQ.md:15-22; ``AUGMENTER_QUEUE_LIMIT``), minus the augmentation itself, which runs on the device."""

    def __init__(self, ds, indexes, batch, classes, pin, depth=2, sampler=None, channels=3):
        """``sampler(n, h, w) -> (batch passes, per-image passes)`` (augment.sample_batch_staged): when given and a batch's items
        share one size, the thread also packs the batch into two pinned blocks and samples its augmentation passes (HostBatch)."""
        import queue
        import threading
        self.q = queue.Queue(maxsize=max(1, depth))
        self._err = None

        def pack(items):
            h, w = items[0].h, items[0].w
            if sampler is None or any((it.h, it.w) != (h, w) for it in items):
                return items
            X = torch.empty((batch, h, w, channels), dtype=torch.uint8, pin_memory=pin)
            Y = torch.empty((batch, h, w), dtype=torch.uint8, pin_memory=pin)
            xv, yv = X.numpy(), Y.numpy()
            for i in range(batch):                      # a short last batch wraps around (static plan batch)
                xv[i], yv[i] = items[i % len(items)].x.numpy(), items[i % len(items)].y.numpy()
            passes, per_item = sampler(batch, h, w)
            return HostBatch(items, X, Y, passes, per_item, h, w)

        def work():
            try:
                for s in range(0, len(indexes), batch):
                    self.q.put(pack([prepare_item(ds[int(i)], classes, False if sampler is not None else pin, channels)
                                     for i in indexes[s:s + batch]]))
            except BaseException as e:      # surfaced on the consumer side
                self._err = e
            self.q.put(None)

        self.t = threading.Thread(target=work, daemon=True)
        self.t.start()

    def __iter__(self):
        while True:
            b = self.q.get()
            if b is None:
                if self._err is not None:
                    raise self._err
                return
            yield b


class DeviceFeeder(object):
    """HostItems -> the plan's input buffers: asynchronous H2D copies of the raw pixels from pinned memory on a COPY
    stream (they overlap the previous step's kernels), then one ``stp_augment_u8`` launch per item on the compute stream
    that resizes to the network shape and augments when training (+ ``stp_filter_u8`` passes for neighbourhood filters)."""

    def __init__(self, device, out_hw, spec, seed, classes=1, channels=3):
        self.device, self.out_hw, self.spec, self.classes = torch.device(device), out_hw, spec, int(classes)
        self.channels = int(channels)          # image channels of the network input (3, or 4..7 for N-channel models)
        self.rng = np.random.RandomState(seed)
        self.pin = self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(device=self.device) if self.pin else None
        self._keep = []

    def feed(self, plan, items, training):
        self._keep = []
        img_buf, msk_buf = plan.inputs["image"].buf, plan.inputs["mask"].buf
        n = img_buf.shape[0]
        oh, ow = self.out_hw
        if isinstance(items, HostBatch) and items.X.shape[0] == n:
            return self._feed_block(plan, items)
        items = [it if isinstance(it, HostItem) else prepare_item(it, self.classes, self.pin, self.channels) for it in items]
        ch = self.channels
        main = torch.cuda.current_stream()
        staged = []
        with torch.cuda.stream(self.copy_stream):
            for i in range(n):
                it = items[i % len(items)]             # a short last batch wraps around (static plan batch)
                xd = torch.empty((it.h, it.w, ch), dtype=torch.uint8, device=self.device)
                yd = torch.empty((it.h, it.w), dtype=torch.uint8, device=self.device)
                xd.copy_(it.x, non_blocking=True)
                yd.copy_(it.y, non_blocking=True)
                xd.record_stream(main); yd.record_stream(main)     # consumed by kernels of the compute stream
                staged.append((it, xd, yd))
        main.wait_stream(self.copy_stream)
        for i, (it, xd, yd) in enumerate(staged):
            passes = augment.sample_staged(self.spec if training else [], self.rng, it.h, it.w, (oh, ow))
            self._keep += [xd, yd, it]
            self._run_passes(xd, yd, img_buf[i], msk_buf[i], [augment.batch_of_one(p) for p in passes], 1, it.h, it.w)

    def _run_passes(self, x, y, img_out, msk_out, passes, n, h, w):
        """Executes the device passes of a sampled pipeline (augment.sample_staged / sample_batch_staged) on ``n`` images of
        ``h`` x ``w``: per pass one ``stp_augment_u8`` launch (warp + point operations, image and mask) and one ``stp_filter_u8``
        launch per neighbourhood filter (ping-pong buffers); the last launch writes the plan's input buffers."""
        ch = self.channels
        lead = (n,) if x.dim() == 4 else ()
        for k, ps_ in enumerate(passes):
            prm, filt, (ph, pw) = ps_[:3]
            final = k == len(passes) - 1
            pd = torch.from_numpy(np.ascontiguousarray(prm, dtype=np.float32)).to(self.device, non_blocking=True)
            ydst = msk_out if final else torch.empty(lead + (ph, pw), dtype=torch.uint8, device=self.device)
            self._keep += [pd, ydst]
            field = None
            if len(ps_) > 4:      # BackgroundReplacer heads this pass: resize the background to the item, composite outside the mask
                bg, erosion = ps_[4]
                if ch != 3 or n != 1:
                    raise ValueError("BackgroundReplacer works on single 3-channel items")
                bgd = torch.from_numpy(bg).to(self.device, non_blocking=True)
                bgr = torch.empty((h, w, ch), dtype=torch.uint8, device=self.device)
                rp = torch.from_numpy(augment.identity_batch(1, bg.shape[0], bg.shape[1], (h, w))).to(self.device, non_blocking=True)
                xr = torch.empty((h, w, ch), dtype=torch.uint8, device=self.device)
                self._keep += [bgd, bgr, rp, xr]
                ops.augment_u8(bgd, None, bgr, None, rp, 1, bg.shape[0], bg.shape[1], h, w, ch)
                ops.background_replace_u8(x, y, bgr, xr, 1, h, w, ch, erosion)
                x = xr
            if len(ps_) > 3 and ps_[3] is not None:      # PiecewiseAffine / ElasticTransformation: the displacement this pass's warp adds
                disp = ps_[3]
                field = torch.empty((n, ph, pw), dtype=torch.int32, device=self.device)
                rd = torch.from_numpy(np.ascontiguousarray(disp[-1], dtype=np.int32)).to(self.device, non_blocking=True)
                self._keep += [field, rd]
                if disp[0] == "piecewise":
                    ops.field_piecewise(field, rd, n, ph, pw, disp[1], disp[2])
                else:
                    tmp = torch.empty_like(field)
                    self._keep.append(tmp)
                    ops.field_elastic(field, tmp, rd, n, ph, pw)
            if filt is None:
                xdst = img_out if final else torch.empty(lead + (ph, pw, ch), dtype=torch.uint8, device=self.device)
                self._keep.append(xdst)
                ops.augment_u8(x, y, xdst, ydst, pd, n, h, w, ph, pw, ch, field)
            else:
                fd = torch.from_numpy(filt).to(self.device, non_blocking=True)
                bufs = [torch.empty(lead + (ph, pw, ch), dtype=torch.uint8, device=self.device) for _ in range(2)]
                self._keep += [fd] + bufs
                ops.augment_u8(x, y, bufs[0], ydst, pd, n, h, w, ph, pw, ch, field)
                src = 0
                for ps in range(filt.shape[0]):
                    last_f = ps == filt.shape[0] - 1
                    dst = img_out if (final and last_f) else bufs[1 - src]
                    ops.filter_u8(bufs[src], dst, fd[ps], n, ph, pw, ch)
                    src = 1 - src
                xdst = img_out if final else bufs[src]
            x, y, h, w = xdst, ydst, ph, pw


def _feed_block(self, plan, hb):
    """Equal-size batch: two H2D copies (pixels, labels) on the copy stream, then the batch's device passes (one
    ``stp_augment_u8`` launch for the common one-pass pipelines); images whose pass structure differs run one by one."""
    img_buf, msk_buf = plan.inputs["image"].buf, plan.inputs["mask"].buf
    n = img_buf.shape[0]
    main = torch.cuda.current_stream()
    with torch.cuda.stream(self.copy_stream):
        xd = torch.empty(hb.X.shape, dtype=torch.uint8, device=self.device)
        yd = torch.empty(hb.Y.shape, dtype=torch.uint8, device=self.device)
        xd.copy_(hb.X, non_blocking=True)
        yd.copy_(hb.Y, non_blocking=True)
        xd.record_stream(main); yd.record_stream(main)
    main.wait_stream(self.copy_stream)
    self._keep += [xd, yd, hb]
    if hb.passes is not None:
        self._run_passes(xd, yd, img_buf, msk_buf, hb.passes, n, hb.h, hb.w)
        return
    for i, item_passes in enumerate(hb.per_item):
        self._run_passes(xd[i], yd[i], img_buf[i], msk_buf[i], [augment.batch_of_one(p) for p in item_passes], 1, hb.h, hb.w)


DeviceFeeder._feed_block = _feed_block


def derived_metrics(scal, classes=1, extended=False):
    """scal: the loss scalars of stp_sigmoid_bce_dice -> Keras-style log entries (metric names of
    schemas/segmentation.raml:98-105: binary_accuracy, dice, iou, iot).  ``extended``: the loss names one of the other registry
    entries (stp_sigmoid_loss_ex / stp_lovasz_hinge leave them in scalars 10..12): they are logged too, so that
    ``primary_metric: val_focal_loss`` or a callback monitoring it sees the quantity it names."""
    loss, bce, dice_l, dice_m, acc, _sp, _sy, _spy, iou, iot = (float(v) for v in scal[:10])
    out = {"loss": loss, ("binary_crossentropy" if classes == 1 else "categorical_crossentropy"): bce, "dice_loss": dice_l,
           "dice": dice_m, "binary_accuracy": acc, "iou": iou, "iot": iot}
    if extended and len(scal) >= 13:
        out.update(iou_loss=1.0 - iou, jaccard_loss=float(scal[10]), focal_loss=float(scal[11]), lovasz_loss=float(scal[12]))
    return out


def _extended(model):
    return len(getattr(model, "loss_w", ())) > 2


class Trainer(object):
    def __init__(self, model, feeder, ds, callbacks, rank=0, world=1):
        self.model, self.feeder, self.ds, self.callbacks = model, feeder, ds, callbacks
        self.rank, self.world = rank, world

    def _aux_stream(self, device):
        if getattr(self, "_aux", None) is None:
            self._aux = torch.cuda.Stream(device=device)
        return self._aux

    def _batches(self, indexes, batch, training):
        f = self.feeder
        oh_ow = f.out_hw
        # the loader thread samples the augmentation records: it gets its OWN generator (seeded from the feeder's), so the
        # training thread's draws (per-item path of DeviceFeeder.feed, DrawResults) never interleave with it
        rng = np.random.RandomState(f.rng.randint(0, 2 ** 31 - 1))
        sampler = lambda n, h, w: augment.sample_batch_staged(f.spec if training else [], rng, n, h, w, oh_ow)
        return HostPrefetcher(self.ds, [int(i) for i in indexes], batch, f.classes, f.pin, sampler=sampler, channels=f.channels)

    def run_epoch_sums(self, indexes, training):
        """One pass over ``indexes`` -> ({log name: sum over batches of value * real samples of the batch}, real samples).
        Keras weights a batch's metric value by the batch's size; a short last batch is padded by wrapping around (static plan
        batch), so for EVALUATION the loss / metric reduction is repeated over the real samples only (Plan.rerun_loss) -
        the duplicates never reach val_loss, the best-checkpoint choice or the callbacks."""
        m = self.model
        plan = m.plan if training else m.eval_plan()
        snaps, counts = [], []
        # Training: the NEXT batch's device passes (resize + augmentation kernels, ~60 us at 16 x 512 x 512) are issued on an auxiliary
        # stream as soon as this step's forward + backward has consumed the input buffers, and run next to the optimizer
        # (HBM-bound, ~100 us) instead of in front of the next forward.  OPT-IN (STP_FEED_OVERLAP=1): measured 0.03 ms SLOWER per step on
        # U-Net/ResNet34 bs16 (profiles/r04j_schedule_ab.txt) - the two stream hand-offs per step cost more than the 57 us kernel they hide.
        overlap = training and plan.device.type == "cuda" and os.environ.get("STP_FEED_OVERLAP", "0") == "1"
        it = iter(self._batches(indexes, m.batch, training))
        items = next(it, None)
        fed = False
        while items is not None:
            n_real = min(len(items), m.batch)
            if not fed:
                self.feeder.feed(plan, items, training)
            nxt, fed = next(it, None), False
            if training and overlap:
                m.forward_backward()
                main = torch.cuda.current_stream()
                if nxt is not None:
                    aux = self._aux_stream(plan.device)
                    aux.wait_stream(main)                       # the forward + backward above has read the input buffers
                    with torch.cuda.stream(aux):
                        self.feeder.feed(plan, nxt, training)
                    fed = True
                m.apply_gradients()
                if fed:
                    main.wait_stream(aux)
                for cb in self.callbacks:
                    if hasattr(cb, "on_batch_end"):
                        cb.on_batch_end(self)
            elif training:
                m.train_on_batch(None, None, fetch=False)
                for cb in self.callbacks:
                    if hasattr(cb, "on_batch_end"):
                        cb.on_batch_end(self)
            else:
                plan.run(plan.prep); plan.run(plan.fwd)
                if n_real < m.batch:
                    plan.rerun_loss(n_real)
            # the step's scalars stay on the device (no host sync per batch: the host keeps running ahead, so the next
            # batch's H2D copies overlap this step); they are fetched once per epoch
            snaps.append(plan.loss_scalars.clone())
            counts.append(n_real)
            items = nxt
        sums = {}
        if snaps:
            for scal, n in zip(torch.stack(snaps).cpu().numpy(), counts):
                for k, v in derived_metrics(scal, getattr(m, "classes", 1), _extended(m)).items():
                    sums[k] = sums.get(k, 0.0) + v * n
        return sums, int(sum(counts))

    def run_epoch(self, indexes, training):
        """Sample-weighted epoch means, combined over all ranks (one small SUM-all-reduce per call): every rank returns
        the same values bit for bit."""
        sums, n = self.run_epoch_sums(indexes, training)
        return reduce_epoch_sums(sums, n, getattr(self.model, "classes", 1), _extended(self.model))


def epoch_log_names(classes=1, extended=False):
    return sorted(derived_metrics(np.zeros(16, np.float32), classes, extended))


def reduce_epoch_sums(sums, n, classes=1, extended=False):
    """{name: weighted sum}, samples -> {name: mean over the samples of ALL ranks}.  The vector layout is fixed by the
    metric names (not by what a rank happened to see), so a rank with an empty shard still takes part in the collective."""
    names = epoch_log_names(classes, extended)
    vec = distributed.allreduce_sums([sums.get(k, 0.0) for k in names] + [float(n)])
    total = vec[-1]
    if total <= 0:
        return {}
    return {k: float(vec[i] / total) for i, k in enumerate(names)}


# ------------------------------------------------------------------------------------------ config
class Stage(object):
    def __init__(self, dict_, cfg):
        self.dict, self.cfg = dict(dict_ or {}), cfg
        d = self.dict
        self.epochs = int(d.get("epochs", 1))
        self.lr = d.get("lr")
        self.loss = d.get("loss")
        self.negatives = d.get("negatives", "real")
        self.validation_negatives = d.get("validation_negatives", self.negatives)
        self.initial_weights = d.get("initial_weights")
        self.unfreeze_encoder = bool(d.get("unfreeze_encoder", False))

    def callbacks(self):
        if "callbacks" in self.dict:
            return make_callbacks(self.dict["callbacks"])
        return make_callbacks(self.cfg.all.get("callbacks")) + make_callbacks(self.dict.get("extra_callbacks"))

    def unfreeze(self, model):
        model.freeze_encoder = False


class GenericTaskConfig(object):
    """Attribute surface of the parsed experiment (``self.all`` keeps the raw YAML, read by createNet1)."""

    def __init__(self, **atrs):
        self.all = dict(atrs)
        a = self.all
        self.path = None
        self.architecture = a.get("architecture")
        self.backbone = a.get("backbone")
        self.encoder_weights = a.get("encoder_weights")
        self.classes = int(a.get("classes", 1))
        self.shape = a.get("shape")
        self.crops = a.get("crops")
        self.batch = int(a.get("batch", 16))
        self.optimizer = a.get("optimizer", "Adam")
        self.lr = float(a.get("lr", 0.001))
        self.clipnorm, self.clipvalue = a.get("clipnorm"), a.get("clipvalue")
        self.loss = a.get("loss", "binary_crossentropy")
        self.metrics = list(a.get("metrics", []) or [])
        self.primary_metric = a.get("primary_metric", "val_loss")
        self.primary_metric_mode = a.get("primary_metric_mode", "auto")
        self.folds_count = int(a.get("folds_count", 5))
        self.random_state = int(a.get("random_state", 33))
        self.testSplit = float(a.get("testSplit", 0.0) or 0.0)
        self.freeze_encoder = bool(a.get("freeze_encoder", False))
        self.augmentation = aug_list(a.get("augmentation"))
        self.transforms = aug_list(a.get("transforms"))
        self.dtype = a.get("dtype", "bf16")          # "bf16" | "fp16" | "fp32" (a key of this backend)
        self.loss_scale = a.get("loss_scale")        # fp16 only: static loss scale (default 2^14)
        self.gpus = int(a.get("gpus", 1))
        self.inference_batch = int(a.get("inference_batch", self.batch))
        self.showDataExamples = False
        # storage of predictions in writeable datasets (reference segmentation.py:196-208 reads both attributes)
        self.compressPredictionsAsInts = bool(a.get("compressPredictionsAsInts", True))
        self.compressScale = a.get("compressScale")
        self.drawingFunction = None
        # the reference always draws validation examples each epoch; `draw_examples: false` in the YAML (a key of this
        # backend) switches the per-epoch JPG sheets off for throughput runs
        self.draw_examples = bool(a.get("draw_examples", True))
        self.resume = False
        self.stages = [self.createStage(s) for s in (a.get("stages") or [{"epochs": 1}])]
        self.dataset_clazz = KFoldedDataSet

    # --- hooks the subclass provides (reference segmentation.py:49-50,93-155)
    def createStage(self, x):
        return Stage(x, self)

    def createNet(self):
        raise NotImplementedError

    def setAllowResume(self, v):
        self.resume = bool(v)

    def clean(self, cleaned):
        cleaned.pop("datasets", None)
        return cleaned

    # --- paths (next to the YAML)
    def _aug_spec(self):
        """``augmentation`` + ``transforms`` with folder arguments (BackgroundReplacer ``path: ./bg``, README.md:275) resolved
        against the experiment's directory."""
        base = os.path.dirname(os.path.abspath(self.path)) if self.path else None
        return augment.resolve_paths(self.augmentation + self.transforms, base)

    def _dir(self, name):
        d = os.path.join(os.path.dirname(os.path.abspath(self.path)), name)
        os.makedirs(d, exist_ok=True)
        return d

    def weightsPath(self, fold, stage):
        return os.path.join(self._dir("weights"), "best-%d.%d.weights" % (fold, stage))

    def metricsPath(self, fold, stage):
        return os.path.join(self._dir("metrics"), "metrics-%d.%d.csv" % (fold, stage))

    def kfold(self, ds, indexes=None):
        if indexes is None:
            indexes = range(len(ds))
        kf = self.dataset_clazz(ds, indexes, self.folds_count, self.random_state, self.testSplit)
        kf.cfg = self            # folds.augmentor(isTrain) builds this experiment's pipeline (reference segmentation.py:223)
        return kf

    def transformAugmentor(self):
        """Validation-time pipeline: ``transforms`` + Resize to the network shape (reference segmentation.py:39, 224)."""
        return TransformAugmentor(self, train=False)

    # --- model lifecycle
    def _compiled(self, stage=None, use_graph=True):
        model = self.createNet()
        loss = (stage.loss if stage is not None and stage.loss else self.loss)
        lr = float(stage.lr) if stage is not None and stage.lr is not None else self.lr
        if stage is not None and stage.unfreeze_encoder:
            stage.unfreeze(model)
        rank, local_rank, world = distributed.env_world()
        device = "cuda:%d" % distributed.device_index(local_rank)
        model.compile(optimizer=self.optimizer, loss=loss, lr=lr, batch=self.batch, dtype=self.dtype, clipnorm=self.clipnorm,
                      clipvalue=self.clipvalue, metrics=self.metrics, device=device, use_graph=use_graph,
                      loss_scale=float(self.loss_scale) if self.loss_scale else None)
        return model

    def load_model(self, fold=0, stage=-1):
        if stage < 0:
            stage = len(self.stages) + stage
        model = self._compiled(self.stages[stage])
        model.load_weights(self.weightsPath(fold, stage))
        return model

    def fit(self, dataset=None, subsample=1.0, foldsToExecute=None, start_from_stage=0):
        """Trains one model per fold and stage; returns the list of per-(fold, stage) summaries.

        ``cfg.gpus = N`` (README.md:756-760) in a process that was not started by torchrun: the reference replicates the graph
        inside the process; here the experiment is re-launched as N ranks through the fit launcher (one process per GPU).
        That needs the dataset to be declared in the YAML (``fit_with`` / ``datasets``) - a live Python dataset object cannot be
        handed to other processes - otherwise a RuntimeError says how to launch."""
        if int(getattr(self, "gpus", 1)) > 1 and "WORLD_SIZE" not in os.environ:
            if dataset is not None:
                raise RuntimeError("cfg.gpus = %d: start the script with `python -m torch.distributed.run --nproc-per-node %d ...` "
                                   "(one process per GPU), or declare the dataset in the YAML (fit_with / datasets) so that "
                                   "fit() can launch the ranks itself" % (self.gpus, self.gpus))
            return self._fit_multi_gpu(foldsToExecute)
        if dataset is None:
            dataset = self._dataset_from_yaml()
        rank, local_rank, world = distributed.init() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else (0, 0, 1)
        if self.crops:
            dataset = CropsDataSet(dataset, self.crops)       # the network is built for shape / crops (createNet1)
        indexes = list(range(len(dataset)))
        if subsample < 1.0:
            indexes = indexes[: max(1, int(len(indexes) * subsample))]
        extra_name = self.all.get("extra_train_data")
        extra_idx = ()
        if extra_name:
            if extra_name not in extra_train:
                raise ValueError("extra_train_data: %r is not registered; set segmentation.extra_train[%r] = dataset before fit() "
                                 "(registered: %s)" % (extra_name, extra_name, ", ".join(sorted(extra_train)) or "none"))
            extra = CropsDataSet(extra_train[extra_name], self.crops) if self.crops else extra_train[extra_name]
            extra_idx = range(len(dataset), len(dataset) + len(extra))
            dataset = ConcatDataSet(dataset, extra)           # folds are drawn from the original indexes only
        kf = self.kfold(dataset, indexes)
        kf.extra_train_indexes = extra_idx
        folds = range(len(kf.folds)) if foldsToExecute is None else foldsToExecute
        summaries = []
        for fold in folds:
            prev = None
            for si, stage in enumerate(self.stages):
                if si < start_from_stage:
                    continue
                wp = self.weightsPath(fold, si)
                if self.resume and os.path.exists(wp) and self._stage_done(fold, si, stage):
                    prev = wp
                    continue
                summaries.append(self._run_stage(kf, dataset, fold, si, stage, prev, rank, world))
                prev = wp
        if rank == 0:
            with open(os.path.join(os.path.dirname(os.path.abspath(self.path)), "summary.yaml"), "w") as f:
                yaml.safe_dump({"primary_metric": self.primary_metric, "stages": summaries}, f)
        return summaries

    def _fit_multi_gpu(self, folds):
        import subprocess
        from . import fit as launcher
        job = {"name": os.path.basename(self.path), "config": os.path.abspath(self.path), "nproc": int(self.gpus), "port": 29500 + os.getpid() % 2000,
               "devices": list(range(int(self.gpus)))}
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        rc = subprocess.call(launcher.command(job, allow_resume=self.resume, folds=folds), env=env)
        if rc:
            raise RuntimeError("multi-GPU fit failed (exit %d)" % rc)
        with open(os.path.join(os.path.dirname(os.path.abspath(self.path)), "summary.yaml")) as f:
            return yaml.safe_load(f)["stages"]

    def _stage_done(self, fold, si, stage):
        mp = self.metricsPath(fold, si)
        if not os.path.exists(mp):
            return False
        with open(mp) as f:
            return sum(1 for _ in f) - 1 >= stage.epochs

    def _run_stage(self, kf, ds, fold, si, stage, prev_weights, rank, world):
        model = self._compiled(stage)
        impl = model.impl
        init = stage.initial_weights or prev_weights
        if init and rank == 0:
            # only rank 0 touches the file (it is the rank that wrote it); the others receive the tensors below
            impl.load_weights(os.path.join(os.path.dirname(os.path.abspath(self.path)), init) if not os.path.isabs(init) else init)
        if world > 1:
            # replicas start the stage bit-identical: parameters, BatchNormalization statistics and optimizer state of rank 0
            impl.broadcast_state(src=0)
            ov = os.environ.get("STP_DP_OVERLAP", "auto")
            # ("buckets" = the per-bucket overlap, as bench.py passes it; "0" / "1" fix the serialised / two-phase schedule)
            impl.set_data_parallel(distributed.make_reducer(), overlap={"0": False, "1": True, "auto": True}.get(ov, "buckets"))
            if ov not in ("0", "1", "buckets"):
                # overlapped or serialised gradient all-reduce: measured on this node ONCE per run, before the first epoch of the first
                # stage (state restored afterwards); later stages and folds reuse the decision
                choice = getattr(self, "_dp_schedule_choice", None)
                if choice is None:
                    rec = impl.calibrate_dp_schedule()
                    self._dp_schedule_choice = choice = (rec or {}).get("chosen", "overlapped")
                elif choice == "serialised":
                    impl.dp_overlap, impl._segments, impl._graphs = False, None, None
                    impl.reducer.reset_bounds()
            elif ov == "0":
                impl.dp_overlap, impl._segments, impl._graphs = False, None, None
                impl.reducer.reset_bounds()
        H, W = impl.H, impl.W                                  # = shape, or shape / crops
        feeder = DeviceFeeder(impl.device, (H, W), self._aug_spec(), seed=self.random_state * 7919 + fold * 101 + si,
                              classes=self.classes, channels=impl.in_ch)
        cbs = stage.callbacks()
        trainer = Trainer(impl, feeder, ds, cbs, rank, world)
        train_idx = kf.sampledIndexes(fold, True, stage.negatives)
        val_idx = kf.sampledIndexes(fold, False, stage.validation_negatives)
        if self.draw_examples and rank == 0:
            # SegmentationStage.add_visualization_callbacks (reference segmentation.py:251-257); rank 0 only: the sheets are files
            # under one fold / stage name, and drawing takes values from the feeder's generator
            cbs.append(DrawResults(self, ds, val_idx, fold, si, drawingFunction=self.drawingFunction))
            if self.showDataExamples:
                cbs.append(DrawResults(self, ds, train_idx, fold, si, train=True, drawingFunction=self.drawingFunction))
        mode = metric_mode(self.primary_metric, self.primary_metric_mode)
        # primary_metric / callback monitors are checked against the names an epoch will log BEFORE the first epoch trains (a typo used
        # to surface only after a whole epoch: advisor finding, round 3)
        known = {}
        for k in epoch_log_names(self.classes, _extended(impl)) + ["lr"]:
            known[k] = 0.0
            known["val_" + k] = 0.0
        for what, name in [("primary_metric", self.primary_metric)] + [("%s.monitor" % type(cb).__name__, cb.monitor) for cb in cbs if hasattr(cb, "monitor")]:
            if log_value(known, name) is None:
                raise ValueError("%s %r is not among the quantities an epoch logs: %s" % (what, name, sorted(known)))
        best, best_epoch, rows = None, -1, []
        t0 = time.time()
        for epoch in range(stage.epochs):
            order = np.array(train_idx)[distributed.shard_indices(len(train_idx), rank, world, epoch, self.random_state + fold)] \
                if len(train_idx) else np.array([], np.int64)
            # Multi-GPU: every quantity a decision depends on is reduced over the ranks first (run_epoch ends in one small
            # SUM-all-reduce), so `best`, EarlyStopping and ReduceLROnPlateau see the same numbers on every rank and no rank
            # can leave the epoch loop or change its learning rate alone.  The validation set is sharded by rank (each sample
            # evaluated once) and the replicas' BatchNormalization moving statistics are averaged before it.
            logs = trainer.run_epoch(order, True)
            if world > 1:
                distributed.average_tensor(impl.plan.S)
            if len(val_idx):
                logs.update({"val_" + k: v for k, v in trainer.run_epoch(distributed.shard_list(val_idx, rank, world), False).items()})
            logs["lr"] = impl.get_lr()
            rows.append(dict(epoch=epoch, **logs))
            cur = log_value(logs, self.primary_metric)
            if cur is None and not len(val_idx) and self.primary_metric.startswith("val_"):
                cur = log_value(logs, self.primary_metric[4:])     # a fold without validation samples: the training value stands in
            if cur is None:      # choosing the best checkpoint on another quantity than the one named would be silent and wrong
                raise ValueError("primary_metric %r is not among the epoch logs %s" % (self.primary_metric, sorted(logs)))
            if best is None or (cur < best if mode == "min" else cur > best):
                best, best_epoch = cur, epoch
                if rank == 0:
                    impl.save_weights(self.weightsPath(fold, si))
            stop = False
            for cb in cbs:
                cb.on_epoch_end(trainer, epoch, logs)
                stop = stop or getattr(cb, "stop", False)
            if rank == 0:
                self._write_metrics(fold, si, rows)
            if world > 1:
                # belt and braces: the decisions were taken on identical inputs; rank 0's learning rate and the OR of the stop
                # flags are made authoritative anyway, so a replica can neither drift nor wait alone in a collective
                v = distributed.allreduce_sums([1.0 if stop else 0.0, impl.get_lr() if rank == 0 else 0.0])
                stop = bool(v[0] > 0)
                if abs(float(v[1]) - impl.get_lr()) > 0:
                    impl.set_lr(float(v[1]))
            if stop:
                break
        if world > 1 and not distributed.replicas_equal(impl.plan.P):
            raise RuntimeError("data-parallel replicas diverged during fold %d stage %d (parameter checksums differ)" % (fold, si))
        distributed.barrier()        # rank 0's checkpoint of this stage is complete before any rank moves on
        return {"fold": int(fold), "stage": int(si), "epochs_run": len(rows), "best_epoch": int(best_epoch),
                self.primary_metric: float(best) if best is not None else None, "seconds": round(time.time() - t0, 3)}

    def _write_metrics(self, fold, si, rows):
        keys = ["epoch"] + sorted(k for k in rows[0] if k != "epoch")
        with open(self.metricsPath(fold, si), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=keys)
            w.writeheader()
            for r in rows:
                w.writerow({k: r.get(k) for k in keys})

    def lr_find(self, d=None, foldsToExecute=None, stage=0, subsample=1.0, start_lr=0.00001, end_lr=1.0, epochs=5):
        """Learning-rate range test (README.md:455-470, the keras_lr_finder procedure): trains fold 0's training set for
        ``epochs`` epochs while multiplying the learning rate after every batch so that it sweeps ``start_lr -> end_lr``
        geometrically; records the loss per batch and stops early once it exceeds 4x the best one.  No weights are saved."""
        if d is None:
            d = self._dataset_from_yaml()
        if self.crops:
            d = CropsDataSet(d, self.crops)       # as fit(): the network is built for shape / crops (createNet1)
        fold = (foldsToExecute or [0])[0]
        kf = self.kfold(d, range(len(d)))
        idx = [int(i) for i in kf.sampledIndexes(fold, True, "all")]
        idx = idx[: max(1, int(len(idx) * subsample))]
        st = self.stages[stage]
        model = self._compiled(st)
        impl = model.impl
        H, W = impl.H, impl.W                                  # = shape, or shape / crops: the size of the plan's input buffers
        feeder = DeviceFeeder(impl.device, (H, W), self._aug_spec(), seed=self.random_state, classes=self.classes, channels=impl.in_ch)
        trainer = Trainer(impl, feeder, d, [], 0, 1)
        nb = max(1, -(-len(idx) // impl.batch)) * int(epochs)
        finder = LRFinder(float(start_lr), float(end_lr), nb)
        lr = float(start_lr)
        impl.set_lr(lr)
        rng = np.random.RandomState(self.random_state)
        for _ in range(int(epochs)):
            order = [idx[i] for i in rng.permutation(len(idx))]
            for items in trainer._batches(order, impl.batch, True):
                feeder.feed(impl.plan, items, True)
                loss = impl.train_on_batch(None, None)["loss"]
                if finder.record(lr, loss):
                    return finder
                lr *= finder.mult
                impl.set_lr(lr)
        return finder

    def info(self, metric=None):
        """Primary metric per fold/stage from the metrics files (README.md:711-718)."""
        metric = metric or self.primary_metric
        mode = metric_mode(metric, self.primary_metric_mode)
        out = []
        mdir = self._dir("metrics")
        for fn in sorted(os.listdir(mdir)):
            if not fn.startswith("metrics-"):
                continue
            fold, stage = (int(v) for v in fn[len("metrics-"):-4].split("."))
            with open(os.path.join(mdir, fn)) as f:
                vals = [float(r[metric]) for r in csv.DictReader(f) if r.get(metric) not in (None, "")]
            if vals:
                out.append({"fold": fold, "stage": stage, metric: (min(vals) if mode == "min" else max(vals))})
        return out

    def _dataset_from_yaml(self):
        """``fit_with: name`` + ``datasets: {name: {input_path, output_path}}`` (examples/people/ds_1.yaml:33-38)."""
        from segmentation_pipeline.impl.datasets import SimplePNGMaskDataSet
        name = self.all.get("fit_with")
        dss = self.all.get("datasets") or {}
        if name and name in dss and "input_path" in dss[name]:
            base = os.path.dirname(os.path.abspath(self.path))
            d = dss[name]
            return SimplePNGMaskDataSet(os.path.join(base, d["input_path"]), os.path.join(base, d["output_path"]))
        raise ValueError("fit() needs a dataset: pass one or declare `fit_with` + `datasets` in the YAML")
