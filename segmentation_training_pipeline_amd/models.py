"""Architecture factories with the constructor signatures the reference resolves by name.

``PipelineConfig.createNet1`` (reference ``segmentation_pipeline/segmentation.py:96-155``) looks the
``architecture`` up in ``custom_models`` / ``segmentation_models`` and calls it with every non-pipeline
YAML key (after alias renaming) that appears in ``inspect.signature`` of the constructor.  The
functions below carry the segmentation_models 0.2.1 keyword names and the defaults of
``schemas/segmentation.raml:158-178`` so that filtering behaves the same; they return a
:class:`SegModel`, which plays the role of the ``keras.Model`` (``compile`` / ``predict`` /
``load_weights`` / ``save_weights`` / ``train_on_batch``) and is backed by the HIP plan.
"""
import os

import numpy as np
import warnings

from . import nets
from .backend import HipSegModel


def pretrained_path(name, backbone):
    root = os.environ.get("STP_PRETRAINED_DIR", os.path.join(os.path.expanduser("~"), ".stp", "pretrained"))
    return os.path.join(root, "%s_%s.weights" % (backbone, name))


def resolve_pretrained(name, backbone):
    """``encoder_weights`` -> checkpoint path or None: an existing file path is taken as is; a name (`imagenet`,
    `pascal_voc`) is looked up as ``$STP_PRETRAINED_DIR/<backbone>_<name>.weights``."""
    if os.path.exists(str(name)):
        return str(name)
    p = pretrained_path(name, backbone)
    return p if os.path.exists(p) else None


def adapt_nchannel(pretrained, current, copy=False):
    """N-channel adaptation of a 3-channel pretrained checkpoint (reference segmentation.py:138-153 builds the 3-channel model
    with its weights and the N-channel one without, then calls musket_core's ``adaptNet(model, model1, self.copyWeights)``).
    musket_core is not vendored in the reference tree, so the rule is restated from its published behaviour (unpinned): every
    tensor of equal shape is copied; a tensor that differs only in the INPUT-CHANNEL axis - the first convolution's kernel
    (kh, kw, 3, f) -> (kh, kw, C, f) and the per-channel vectors of the input BatchNormalization (3,) -> (C,) - keeps the
    pretrained values in channels 0..2, the N-channel model's own fresh values elsewhere (zero kernels, beta 0, mean 0,
    variance 1), and with ``copyWeights: true`` channel 3 starts as a copy of pretrained channel 2.
    ``pretrained`` / ``current``: dict name -> numpy in Keras layouts; returns the dict to load (names ``current`` has)."""
    out = {}
    for name, a in pretrained.items():
        if name not in current:
            continue
        a, cur = np.asarray(a, np.float32), np.asarray(current[name], np.float32)
        if a.shape == cur.shape:
            out[name] = a
        elif a.ndim == 4 and cur.ndim == 4 and a.shape[:2] == cur.shape[:2] and a.shape[3] == cur.shape[3] and a.shape[2] == 3 < cur.shape[2]:
            v = np.zeros_like(cur)
            v[:, :, :3, :] = a
            if copy:
                v[:, :, 3, :] = a[:, :, 2, :]
            out[name] = v
        elif a.ndim == 1 and cur.ndim == 1 and a.shape[0] == 3 < cur.shape[0]:
            v = cur.copy()
            v[:3] = a
            if copy:
                v[3] = a[2]
            out[name] = v
        else:
            raise ValueError("%s: pretrained shape %s cannot be adapted to %s" % (name, a.shape, cur.shape))
    return out


class SegModel(object):
    """Uncompiled model description; ``compile`` builds the HIP training plan."""

    def __init__(self, architecture, backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder,
                 decoder_filters):
        if input_shape is None or input_shape[0] is None or input_shape[1] is None:
            raise ValueError("the HIP backend builds static plans: give `shape: [H, W, C]` in the experiment YAML")
        self.architecture, self.backbone_name = architecture, backbone_name
        self.input_shape = tuple(int(v) for v in input_shape)
        self.classes, self.activation = int(classes), activation
        self.encoder_weights, self.freeze_encoder = encoder_weights, bool(freeze_encoder)
        self.decoder_filters = tuple(decoder_filters)
        self.impl = None
        self._pending_weights = None

    def compile(self, optimizer="Adam", loss="binary_crossentropy", lr=1e-3, batch=16, dtype="bf16", clipnorm=None,
                clipvalue=None, metrics=None, device="cuda", use_graph=True, opt_kwargs=None, loss_scale=None):
        self.impl = HipSegModel(self.architecture, self.backbone_name, self.input_shape, self.classes, self.activation,
                                decoder_block_type=getattr(self, "decoder_block_type", "upsampling"),
                                batch=batch, dtype=dtype, loss=loss, optimizer=optimizer, lr=lr,
                                freeze_encoder=self.freeze_encoder, decoder_filters=self.decoder_filters, clipnorm=clipnorm,
                                clipvalue=clipvalue, use_graph=use_graph, device=device, opt_kwargs=opt_kwargs,
                                net_kwargs=getattr(self, "net_kwargs", None), loss_scale=loss_scale)
        ew = self.encoder_weights
        if ew:
            path = resolve_pretrained(ew, self.backbone_name)
            nch = getattr(self, "nchannel_adapt", None)        # set by createNet1 for C > 3 with encoder_weights (reference :138-153)
            if nch is not None and os.path.exists(nch["cache"]):
                self.impl.load_weights(nch["cache"])              # `<experiment>.mdl-nchannel` written by an earlier createNet
            elif path is not None and nch is not None:
                from safetensors.numpy import load_file
                cur = self.impl.get_weights()
                self.impl.set_weights(adapt_nchannel(load_file(path), cur, copy=nch["copy"]))
                self.impl.save_weights(nch["cache"])
            elif path is not None:
                self.impl.load_weights(path, strict=False)
            elif os.environ.get("STP_ALLOW_RANDOM_ENCODER") == "1" or getattr(self, "allow_random_encoder_init", False):
                warnings.warn("encoder_weights=%r: no pretrained file found; the encoder starts from he_uniform "
                              "initialisation (STP_ALLOW_RANDOM_ENCODER=1)" % (ew,))
            else:
                # the reference downloads these weights; a silent random start would train something else under the same YAML
                raise RuntimeError(
                    "encoder_weights=%r: no pretrained weights for %r are available offline. Put a checkpoint at %s "
                    "(safetensors, float32, Keras layout, keys '<layer>/<weight>': 'conv0/kernel' (7, 7, 3, 64) HWIO, 'bn0/gamma', "
                    "'bn0/moving_mean', 'stage1_unit1_conv1/kernel' ... - the key list is HipSegModel.get_weights().keys(); "
                    "docs/PRETRAINED_WEIGHTS.md has the layout table and a conversion snippet), or pass a path as encoder_weights, "
                    "or set `encoder_weights: null` "
                    "in the YAML, or export STP_ALLOW_RANDOM_ENCODER=1 to train from random initialisation knowingly."
                    % (ew, self.backbone_name, pretrained_path(ew, self.backbone_name)))
        if self._pending_weights is not None:
            self.impl.load_weights(self._pending_weights)
            self._pending_weights = None
        return self

    def _need(self):
        if self.impl is None:
            self.compile()
        return self.impl

    def predict(self, x):
        return self._need().predict(x)

    def train_on_batch(self, x, y):
        return self._need().train_on_batch(x, y)

    def load_weights(self, path):
        if self.impl is None:
            self._pending_weights = path
        else:
            self.impl.load_weights(path)

    def save_weights(self, path):
        self._need().save_weights(path)


def Unet(backbone_name="vgg16", input_shape=(None, None, 3), classes=1, activation="sigmoid", encoder_weights="imagenet",
         freeze_encoder=False, decoder_block_type="upsampling", decoder_filters=(256, 128, 64, 32, 16),
         decoder_use_batchnorm=True, n_upsample_blocks=5, upsample_rates=(2, 2, 2, 2, 2)):
    """segmentation_models.Unet keyword surface (schemas/segmentation.raml:158-178)."""
    if backbone_name not in nets.known_backbones():
        raise ValueError("Unknown backbone")
    if decoder_block_type not in ("upsampling", "transpose") or not decoder_use_batchnorm or int(n_upsample_blocks) != 5 \
            or tuple(upsample_rates) != (2, 2, 2, 2, 2):
        raise ValueError("the HIP Unet implements the upsampling and transpose decoder blocks with BatchNorm, 5 x2 stages")
    m = SegModel("Unet", backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder, decoder_filters)
    m.decoder_block_type = decoder_block_type
    return m


def Linknet(backbone_name="vgg16", input_shape=(None, None, 3), classes=1, activation="sigmoid", encoder_weights="imagenet",
            freeze_encoder=False, decoder_filters=(None, None, None, None, 16), decoder_use_batchnorm=True,
            decoder_block_type="upsampling", n_upsample_blocks=5, upsample_rates=(2, 2, 2, 2, 2)):
    """segmentation_models.Linknet keyword surface (schemas/segmentation.raml:180-203)."""
    if backbone_name not in nets.RESNET_UNITS and backbone_name not in nets.VGG_BLOCKS:
        raise ValueError("Unknown backbone")
    if decoder_block_type not in ("upsampling", "transpose") or not decoder_use_batchnorm or int(n_upsample_blocks) != 5 \
            or tuple(upsample_rates) != (2, 2, 2, 2, 2) or decoder_filters[4] is None:
        raise ValueError("the HIP Linknet implements the upsampling / transpose decoder blocks with BatchNorm, 5 x2 stages")
    m = SegModel("Linknet", backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder, decoder_filters)
    m.decoder_block_type = decoder_block_type
    return m


def FPN(backbone_name="vgg16", input_shape=(None, None, 3), classes=21, activation="softmax", encoder_weights="imagenet",
        freeze_encoder=False, fpn_layers="default", pyramid_block_filters=256, segmentation_block_filters=128,
        upsample_rates=(2, 2, 2), last_upsample=4, interpolation="bilinear", use_batchnorm=True, dropout=None):
    """segmentation_models.FPN keyword surface (schemas/segmentation.raml:180-203)."""
    if backbone_name not in nets.RESNET_UNITS and backbone_name not in nets.VGG_BLOCKS:
        raise ValueError("Unknown backbone")
    if fpn_layers != "default" or tuple(upsample_rates) != (2, 2, 2) or int(last_upsample) != 4 or interpolation not in ("bilinear", "nearest") \
            or not use_batchnorm or (dropout and not 0.0 < float(dropout) < 1.0):
        raise ValueError("the HIP FPN implements the x2-rate pyramid ending at 1/4 resolution with bilinear resizes and BatchNorm")
    pf, sf = int(pyramid_block_filters), int(segmentation_block_filters)
    if pf % 8 or sf % 8 or pf <= 0 or sf <= 0:
        raise ValueError("pyramid_block_filters / segmentation_block_filters must be multiples of 8")
    m = SegModel("FPN", backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder, ())
    m.net_kwargs = {"pyramid_block_filters": pf, "segmentation_block_filters": sf, "dropout": float(dropout) if dropout else None,
                    "interpolation": interpolation}
    return m


def PSPNet(backbone_name="vgg16", input_shape=(384, 384, 3), classes=21, activation="softmax", encoder_weights="imagenet",
           freeze_encoder=False, downsample_factor=8, psp_conv_filters=512, psp_pooling_type="avg", use_batchnorm=True, dropout=None,
           final_interpolation="bilinear"):
    """segmentation_models.PSPNet keyword surface (schemas/segmentation.raml:225-249)."""
    if backbone_name not in nets.RESNET_UNITS and backbone_name not in nets.VGG_BLOCKS:
        raise ValueError("Unknown backbone")
    if int(downsample_factor) not in (4, 8, 16) or psp_pooling_type not in ("avg", "max") or not use_batchnorm or final_interpolation not in ("bilinear", "nearest") \
            or (dropout and not 0.0 < float(dropout) < 1.0):
        raise ValueError("the HIP PSPNet implements downsample_factor 4 / 8 / 16, avg / max pooling, BatchNorm, bilinear / nearest final resize")
    if int(psp_conv_filters) % 8 or int(psp_conv_filters) <= 0:
        raise ValueError("psp_conv_filters must be a multiple of 8")
    m = SegModel("PSPNet", backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder, ())
    m.net_kwargs = {"downsample_factor": int(downsample_factor), "psp_conv_filters": int(psp_conv_filters),
                    "dropout": float(dropout) if dropout else None, "final_interpolation": final_interpolation, "psp_pooling_type": psp_pooling_type}
    return m


def Deeplabv3(encoder_weights="pascal_voc", input_tensor=None, input_shape=(512, 512, 3), classes=21, backbone_name="mobilenetv2", OS=16,
              alpha=1.0, activation=None, freeze_encoder=False):
    """The reference's in-tree constructor (segmentation_pipeline/impl/deeplab/model.py:281; registered in
    ``custom_models`` as ``DeepLabV3``, segmentation.py:31-33), with its own argument checks (:317-328)."""
    if encoder_weights not in ("pascal_voc", None) and not os.path.exists(str(encoder_weights)):
        raise ValueError("The `encoder_weights` argument should be either `None` (random initialization) or `pascal_voc` "
                         "(pre-trained on PASCAL VOC)")
    if backbone_name not in ("xception", "mobilenetv2"):
        raise ValueError("The `backbone_name` argument should be either `xception`  or `mobilenetv2` ")
    if backbone_name == "mobilenetv2" and float(alpha) != 1.0:
        raise ValueError("the HIP DeepLabV3 implements the mobilenetv2 branch with alpha = 1")
    if backbone_name == "xception" and int(OS) not in (8, 16):
        raise ValueError("OS (output stride of the xception backbone) is 8 or 16")
    if not ((activation == "sigmoid" and int(classes) == 1) or (activation == "softmax" and 2 <= int(classes) <= 32)):
        raise ValueError("the HIP DeepLabV3 trains the 1-class sigmoid head and 2..32-class softmax heads")
    mdl = SegModel("DeepLabV3", backbone_name, input_shape, classes, activation, encoder_weights, freeze_encoder, ())
    if backbone_name == "xception":
        mdl.net_kwargs = {"OS": int(OS)}
    return mdl


ARCHITECTURES = {"Unet": Unet, "Linknet": Linknet, "FPN": FPN, "PSPNet": PSPNet}


def known_backbones():
    return nets.known_backbones()
