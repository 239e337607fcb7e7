"""Drop-in ``segmentation_pipeline.segmentation``: same names as the reference module
(``/root/reference/segmentation_pipeline/segmentation.py``), backed by the MI355X-native HIP path.

    from segmentation_pipeline import segmentation
    cfg = segmentation.parse("config.yaml")      # reference :211-214
    cfg.fit(ds)                                   # reference README.md:125

What is mirrored: ``parse`` (sets ``cfg.path``), ``PipelineConfig`` with ``createNet`` / ``createNet1``
(the YAML -> constructor-kwargs rules of reference :96-155: ``activation: none``, architecture lookup
order, backbone check and its error messages, alias renaming, signature filtering, ``crops``),
``createStage`` / ``SegmentationStage.unfreeze`` (:49-50, :249-260), ``custom_models`` (:31-33), the
loss/metric name registry (:15-22), ``evaluate`` / ``update`` (:37-47, :58-60), ``predict_to_directory`` (:62-79) and
``load_writeable_dataset`` / ``create_writeable_dataset`` (:196-208).  The Keras / imgaug objects
behind those names are replaced by the HIP plan; nothing here computes on the CPU.
"""
import inspect
import os

import numpy as np

from segmentation_training_pipeline_amd import models as _models
from segmentation_training_pipeline_amd import pipeline as generic
from segmentation_training_pipeline_amd.pipeline import ALIASES, CUSTOM_KEYS

# name -> what the HIP loss/metric kernel provides (reference :15-22 registers these names with Keras)
custom_objects = {
    "dice": "dice", "iou": "iou", "iot": "iot", "dice_loss": "dice_loss", "binary_crossentropy": "binary_crossentropy",
    "categorical_crossentropy": "categorical_crossentropy", "binary_accuracy": "binary_accuracy",
    "iou_loss": "iou_loss", "jaccard_loss": "jaccard_loss", "focal_loss": "focal_loss",        # sigmoid head (stp_sigmoid_loss_ex)
    "lovasz_loss": "lovasz_loss",                                                                # sigmoid head (stp_lovasz_hinge)
}
unsupported_objects = ()

extra_train = generic.extra_train     # name -> dataset added to every fold's training indexes (reference :29, README.md:698-709)
dataset_augmenters = {}

# name -> fn(**arch_kwargs) -> model; the reference ships one entry, its in-tree DeepLabV3+ (reference :31-33), and lets users
# add more (README.md:636-643)
custom_models = {"DeepLabV3": _models.Deeplabv3}


def ansemblePredictions(sourceFolder, folders, cb, data, weights=None):
    """Averages per-image .npy predictions of several folders (README.md:745-754)."""
    for f in sorted(os.listdir(sourceFolder)):
        stem = f[0:f.index(".")] if "." in f else f
        arrs = [np.load(os.path.join(d, stem + ".npy")) for d in folders]
        w = weights or [1.0] * len(arrs)
        cb(f, sum(a * wi for a, wi in zip(arrs, w)) / float(sum(w)), data)


class SegmentationStage(generic.Stage):
    def unfreeze(self, model):
        # reference :259-260 -> segmentation_models.utils.set_trainable(model): every layer trainable
        model.freeze_encoder = False


class PipelineConfig(generic.GenericTaskConfig):
    def __init__(self, **atrs):
        super().__init__(**atrs)
        self.dataset_clazz = generic.KFoldedDataSet
        self.flipPred = True

    def createStage(self, x):
        return SegmentationStage(x, self)

    def evaluate(self, d, fold, stage, negatives="all", limit=16):
        """Reference :37-47: up to ``limit`` validation items of ``fold`` go through the validation pipeline
        (``transformAugmentor``: ``transforms`` + Resize, on the device) and ``model.predict``; yields the batch with
        ``images_aug`` and ``heatmaps_aug`` (one probability map per item, ``.arr`` H x W x classes) filled in."""
        mdl = self.load_model(fold, stage)
        ta = self.transformAugmentor()
        folds = self.kfold(d, range(0, len(d)))
        rs = folds.load(fold, False, negatives, limit)
        for z in ta.augment_batches([rs]):
            res = mdl.predict(np.array(z.images_aug))
            z.heatmaps_aug = [PredictedMap(x) for x in res]
            yield z

    def update(self, z, res):
        """Reference :58-60: attaches predictions ``res`` to the batch ``z`` as its ``segmentation_maps_aug``."""
        z.segmentation_maps_aug = [PredictedMap(x) for x in res]

    def _writeable(self, ds, path, count):
        from segmentation_pipeline.impl.datasets import CompressibleWriteableDS
        resName = (ds.name if hasattr(ds, "name") else "") + "_predictions"
        if self.compressScale is not None:
            return CompressibleWriteableDS(ds, resName, path, count, asUints=self.compressPredictionsAsInts, scale=self.compressScale)
        return CompressibleWriteableDS(ds, resName, path, count, asUints=self.compressPredictionsAsInts)

    def load_writeable_dataset(self, ds, path):
        """Reference :196-201: re-opens the predictions stored for ``ds`` under ``path`` (one entry per item of ``ds``)."""
        return self._writeable(ds, path, len(ds))

    def create_writeable_dataset(self, dataset, dsPath):
        """Reference :203-208: an empty predictions dataset over ``dataset`` stored under ``dsPath`` (``append`` + ``commit``)."""
        return self._writeable(dataset, dsPath, 0)

    def createNet(self):
        return self.createNet1(False)

    def createNet1(self, forInference):
        ac = self.all.get("activation")
        if ac == "none":
            ac = None
        self.all["activation"] = ac
        if self.architecture in custom_models:
            clazz = custom_models[self.architecture]
        else:
            if self.architecture not in _models.ARCHITECTURES:
                print("Unknown architecture:" + str(self.architecture))
                print("Known architectures:", sorted(_models.ARCHITECTURES))
                raise ValueError("Unknown architecture")
            clazz = _models.ARCHITECTURES[self.architecture]
            if str(self.backbone).lower() not in _models.known_backbones():
                print("Unknown backbone:" + str(self.backbone))
                print("Known backbones:", _models.known_backbones())
                raise ValueError("Unknown backbone")
        self.backbone = str(self.backbone).lower()
        self.all["backbone"] = self.backbone
        cleaned = {}
        sig = inspect.signature(clazz)
        for arg in self.all:
            pynama = ALIASES.get(arg, arg)
            if arg not in CUSTOM_KEYS and pynama in sig.parameters:
                cleaned[pynama] = self.all[arg]
        self.clean(cleaned)
        if forInference and "weights" in cleaned:
            cleaned["weights"] = None
        if forInference and "encoder_weights" in cleaned:
            cleaned["encoder_weights"] = None     # trained weights are loaded right after (load_model)
        if self.crops is not None and "input_shape" in cleaned:
            s = cleaned["input_shape"]
            cleaned["input_shape"] = (s[0] // self.crops, s[1] // self.crops, s[2])
        nchannel = None
        if "input_shape" in cleaned and cleaned["input_shape"][2] > 3:
            if cleaned["input_shape"][2] > 7:
                raise ValueError("the HIP backend takes images of up to 7 channels")
            # reference :135-153: with encoder_weights an N-channel model is built from the 3-channel pretrained one (adaptNet
            # copies the first convolution's kernels into the wider one, `copyWeights` seeds channel 3 from channel 2) and cached
            # as `<experiment>.mdl-nchannel`; without them the N-channel model is simply built.  The adaptation runs when the
            # model is compiled (models.adapt_nchannel), where the pretrained file is read.
            ew = cleaned.get("encoder_weights")
            if ew is not None and len(str(ew)) > 0:
                nchannel = {"cache": str(self.path) + ".mdl-nchannel", "copy": bool(self.all.get("copyWeights", False))}
        model = clazz(**cleaned)
        if nchannel is not None:
            model.nchannel_adapt = nchannel
        return model

    def load_model(self, fold=0, stage=-1):
        if stage < 0:
            stage = len(self.stages) + stage
        model = self.createNet1(True)
        st = self.stages[stage]
        model.compile(optimizer=self.optimizer, loss=(st.loss or self.loss), lr=self.lr, batch=self.inference_batch,
                      dtype=self.dtype)
        model.load_weights(self.weightsPath(fold, stage))
        return model

    def get_eval_batch(self):
        return self.inference_batch

    # ---------------------------------------------------------------- inference (reference :62-91, :158-191)
    def _models(self, fold, stage):
        folds = fold if isinstance(fold, (list, tuple)) else [fold]
        return [self.load_model(f, stage) for f in folds]

    def _resize_to_net(self, impl, images):
        """uint8 HxWxC images of any size -> uint8 [n, H, W, C] at the network shape (stp_augment_u8, identity + resize)."""
        from segmentation_training_pipeline_amd import ops
        import torch
        H, W, ch = impl.H, impl.W, impl.in_ch
        xs = np.zeros((len(images), H, W, ch), np.uint8)
        for i, img in enumerate(images):
            img = np.asarray(img)
            if img.ndim == 2:
                img = img[:, :, None]
            if img.shape[2] < ch:
                if img.shape[2] != 1:
                    raise ValueError("image has %d channels, the network expects %d" % (img.shape[2], ch))
                img = np.repeat(img, ch, axis=2)
            img = np.ascontiguousarray(img[:, :, :ch], dtype=np.uint8)
            h, w = img.shape[:2]
            prm = torch.from_numpy(generic.augment.identity_batch(1, h, w, (H, W))).to(impl.device)
            src = torch.from_numpy(img).to(impl.device)
            dst = torch.empty((1, H, W, ch), dtype=torch.uint8, device=impl.device)
            ops.augment_u8(src, None, dst, None, prm, 1, h, w, H, W, ch)
            xs[i] = dst[0].cpu().numpy()
        return xs

    def predict_on_batch(self, models, ttflips, xs):
        """Mean of the fold models' probabilities, optionally with flip test-time augmentation (README.md:519-534).
        ``models``: one model or a list; ``xs``: uint8 [n, H, W, 3] at the network shape, n <= inference batch."""
        models = models if isinstance(models, (list, tuple)) else [models]
        acc = np.zeros((len(xs),) + tuple(xs.shape[1:3]) + (self.classes,), np.float32)
        k = 0
        for m in models:
            acc += m.predict(xs); k += 1
            if ttflips:
                acc += m.predict(xs[:, :, ::-1])[:, :, ::-1]; k += 1
                acc += m.predict(xs[:, ::-1])[:, ::-1]; k += 1
        return acc / k

    @staticmethod
    def _scale_back(p, h, w):
        """Prediction at the network shape -> original image size (nearest, like the reference's Scale of a map)."""
        yy = (np.arange(h) * p.shape[0] // h)[:, None]
        xx = (np.arange(w) * p.shape[1] // w)[None, :]
        return p[yy, xx]

    def predict_on_directory(self, spath, fold=0, stage=0, limit=-1, batch_size=32, ttflips=False):
        from segmentation_pipeline.impl.datasets import DirectoryDataSet
        nets_ = self._models(fold, stage)
        ds = DirectoryDataSet(spath)
        n = len(ds) if limit < 0 else min(limit, len(ds))
        impl0 = nets_[0].impl
        B = impl0.batch
        for s in range(0, n, B):
            items = [ds[i] for i in range(s, min(s + B, n))]
            if self.crops:
                yield items, [self._predict_cells(nets_, ttflips, it.x) for it in items]
                continue
            xs = self._resize_to_net(impl0, [it.x for it in items])
            yield items, self.predict_on_batch(nets_, ttflips, xs)

    def _predict_cells(self, models, ttflips, img):
        """``crops: N`` at prediction time (README.md:488-491): the image is split into the N x N cells the model was trained
        on, every cell is predicted, scaled back to its own size and the map is assembled - invisible to the caller."""
        from segmentation_training_pipeline_amd.pipeline import crop_bounds
        c = int(self.crops)
        h, w = img.shape[:2]
        ys, xs_ = crop_bounds(h, c), crop_bounds(w, c)
        cells = [img[ys[r]:ys[r + 1], xs_[q]:xs_[q + 1]] for r in range(c) for q in range(c)]
        impl0 = models[0].impl
        out = np.zeros((h, w, self.classes), np.float32)
        for s in range(0, len(cells), impl0.batch):
            chunk = cells[s:s + impl0.batch]
            probs = self.predict_on_batch(models, ttflips, self._resize_to_net(impl0, chunk))
            for k, (cell, p) in enumerate(zip(chunk, probs), start=s):
                r, q = divmod(k, c)
                out[ys[r]:ys[r + 1], xs_[q]:xs_[q + 1]] = self._scale_back(p, *cell.shape[:2])
        return out

    def predict_to_directory(self, spath, tpath, fold=0, stage=0, limit=-1, batchSize=32, binaryArray=False, ttflips=False):
        os.makedirs(tpath, exist_ok=True)
        from PIL import Image
        for items, probs in self.predict_on_directory(spath, fold=fold, stage=stage, limit=limit, batch_size=batchSize, ttflips=ttflips):
            for it, p in zip(items, probs):
                scaled = self._scale_back(p, *it.x.shape[:2])
                stem = it.id[0:it.id.index(".")] if "." in it.id else it.id
                if binaryArray:
                    np.save(os.path.join(tpath, stem), scaled)
                else:
                    Image.fromarray((scaled[:, :, 0] * 255).astype(np.uint8)).save(os.path.join(tpath, stem + ".png"))

    def predict_in_directory(self, spath, fold, stage, cb=None, data=None, limit=-1, batchSize=32, ttflips=False):
        """Calls ``cb(file_name, map, data)`` per image with ``map.arr`` = probabilities at the ORIGINAL image size
        (reference :81-91; README.md:498-527).  ``fold`` may be a list (ensemble).  The README's ensembling example
        omits ``stage`` (``predict_in_directory(path, folds, cb, data)``): that call shape is accepted too."""
        if callable(stage):
            stage, cb, data = 0, stage, cb
        for items, probs in self.predict_on_directory(spath, fold=fold, stage=stage, limit=limit, batch_size=batchSize, ttflips=ttflips):
            for it, p in zip(items, probs):
                cb(it.id, PredictedMap(self._scale_back(p, *it.x.shape[:2])), data)

    def evaluateAll(self, ds, fold, stage=-1, negatives="real", ttflips=None):
        """Iterator over validation batches of ``fold`` (reference :158-191): each carries the original ``images``,
        ``data`` (ids), ``segmentation_maps`` (ground truth) and ``predicted_maps_aug`` (probabilities, original size)."""
        folds = self.kfold(ds, range(0, len(ds)))
        indexes = [int(i) for i in folds.sampledIndexes(fold, False, negatives)]
        m = self.load_model(fold, stage)
        B = m.impl.batch
        for s in range(0, len(indexes), B):
            items = [ds[i] for i in indexes[s:s + B]]
            if self.crops:
                probs = [self._predict_cells([m], ttflips, it.x) for it in items]
            else:
                probs = self.predict_on_batch(m, ttflips, self._resize_to_net(m.impl, [it.x for it in items]))
            yield EvalBatch(images=[it.x for it in items], data=[it.id for it in items],
                            segmentation_maps=[PredictedMap(np.asarray(it.y)) for it in items],
                            predicted_maps_aug=[PredictedMap(self._scale_back(p, *it.x.shape[:2])) for it, p in zip(items, probs)])


class PredictedMap(object):
    """What callbacks receive in place of imgaug's SegmentationMapOnImage: ``.arr`` is the HxWxC array."""

    def __init__(self, arr):
        self.arr = arr
        self.shape = arr.shape


class EvalBatch(object):
    """Stand-in for imgaug.Batch as produced by evaluateAll (reference :184-186)."""

    def __init__(self, images, data, segmentation_maps, predicted_maps_aug):
        self.images, self.data = images, data
        self.segmentation_maps, self.predicted_maps_aug = segmentation_maps, predicted_maps_aug


def parse(path) -> PipelineConfig:
    cfg = PipelineConfig(**generic.load_yaml(path))
    cfg.path = path
    return cfg
