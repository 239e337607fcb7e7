"""Drop-in ``segmentation_pipeline.segmentation``: same names as the reference module
(``/root/reference/segmentation_pipeline/segmentation.py``), backed by the MI355X-native HIP path.

    from segmentation_pipeline import segmentation
    cfg = segmentation.parse("config.yaml")      # reference :211-214
    cfg.fit(ds)                                   # reference README.md:125

What is mirrored: ``parse`` (sets ``cfg.path``), ``PipelineConfig`` with ``createNet`` / ``createNet1``
(the YAML -> constructor-kwargs rules of reference :96-155: ``activation: none``, architecture lookup
order, backbone check and its error messages, alias renaming, signature filtering, ``crops``),
``createStage`` / ``SegmentationStage.unfreeze`` (:49-50, :249-260), ``custom_models`` (:31-33), the
loss/metric name registry (:15-22) and ``predict_to_directory`` (:62-79).  The Keras / imgaug objects
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
    "dice": "dice", "iou": "iou", "dice_loss": "dice_loss", "binary_crossentropy": "binary_crossentropy",
    "binary_accuracy": "binary_accuracy",
}
# registered by the reference but outside the first hot-path bar: named so the error is explicit
unsupported_objects = ("iot", "lovasz_loss", "iou_loss", "jaccard_loss", "focal_loss")

extra_train = {}
dataset_augmenters = {}

custom_models = {}          # user-registered: name -> fn(**arch_kwargs) -> model  (reference :31-33, README.md:636-643)


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
        if "input_shape" in cleaned and cleaned["input_shape"][2] > 3:
            raise ValueError("more than 3 input channels is not available in the HIP backend yet")
        return clazz(**cleaned)

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

    # ---------------------------------------------------------------- inference to disk (reference :62-79)
    def predict_on_directory(self, spath, fold=0, stage=0, limit=-1, batch_size=32, ttflips=False):
        from segmentation_pipeline.impl.datasets import DirectoryDataSet
        from segmentation_training_pipeline_amd import ops
        import torch
        folds = fold if isinstance(fold, (list, tuple)) else [fold]
        nets_ = [self.load_model(f, stage) for f in folds]
        ds = DirectoryDataSet(spath)
        n = len(ds) if limit < 0 else min(limit, len(ds))
        H, W = int(self.shape[0]), int(self.shape[1])
        impl0 = nets_[0].impl
        B = impl0.batch
        for s in range(0, n, B):
            items = [ds[i] for i in range(s, min(s + B, n))]
            xs = np.zeros((len(items), H, W, 3), np.uint8)
            for i, it in enumerate(items):          # Resize to the network shape on the device
                h, w = it.x.shape[:2]
                prm = torch.from_numpy(generic.augment.identity_batch(1, h, w, (H, W))).to(impl0.device)
                src = torch.from_numpy(np.ascontiguousarray(it.x)).to(impl0.device)
                dst = torch.empty((1, H, W, 3), dtype=torch.uint8, device=impl0.device)
                ops.augment_u8(src, None, dst, None, prm, 1, h, w, H, W, 3)
                xs[i] = dst[0].cpu().numpy()
            acc = np.zeros((len(items), H, W, self.classes), np.float32)
            k = 0
            for m in nets_:
                acc += m.predict(xs); k += 1
                if ttflips:                          # flip test-time augmentation (README.md:534)
                    acc += m.predict(xs[:, :, ::-1])[:, :, ::-1]; k += 1
                    acc += m.predict(xs[:, ::-1])[:, ::-1]; k += 1
            yield items, acc / k

    def predict_to_directory(self, spath, tpath, fold=0, stage=0, limit=-1, batchSize=32, binaryArray=False, ttflips=False):
        os.makedirs(tpath, exist_ok=True)
        from PIL import Image
        for items, probs in self.predict_on_directory(spath, fold=fold, stage=stage, limit=limit, batch_size=batchSize, ttflips=ttflips):
            for it, p in zip(items, probs):
                h, w = it.x.shape[:2]
                yy = (np.arange(h) * p.shape[0] // h)[:, None]
                xx = (np.arange(w) * p.shape[1] // w)[None, :]
                scaled = p[yy, xx]                    # back to the original image size
                stem = it.id[0:it.id.index(".")] if "." in it.id else it.id
                if binaryArray:
                    np.save(os.path.join(tpath, stem), scaled)
                else:
                    Image.fromarray((scaled[:, :, 0] * 255).astype(np.uint8)).save(os.path.join(tpath, stem + ".png"))


def parse(path) -> PipelineConfig:
    cfg = PipelineConfig(**generic.load_yaml(path))
    cfg.path = path
    return cfg
