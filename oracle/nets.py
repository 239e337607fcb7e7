"""Oracle (test infrastructure): U-Net over a pre-activation ResNet encoder, PyTorch-CPU fp32.

Restates the graph that ``segmentation_pipeline/segmentation.py:113,155`` obtains from
``segmentation_models.Unet(backbone_name='resnet34', input_shape=(H,W,3), classes=1,
activation='sigmoid', decoder_block_type='upsampling', decoder_filters=(256,128,64,32,16),
decoder_use_batchnorm=True)`` - kwargs/defaults from
``segmentation_pipeline/schemas/segmentation.raml:158-178`` - i.e. segmentation_models 0.2.1
(``requires.txt:15``) over the classification_models ResNet (``segmentation.py:5``).
Neither package is vendored or installable here: PARITY UNPINNED (see oracle/__init__.py).

Layouts exposed: activations NHWC, conv kernels HWIO (Keras), BN vectors [C].
Internally tensors are moved to NCHW for torch.nn.functional primitives.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

RESNET_UNITS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3),
                "resnet152": (3, 8, 36, 3)}
BOTTLENECK = ("resnet50", "resnet101", "resnet152")   # classification_models residual_bottleneck_block: 1x1 f, 3x3 f (stride), 1x1 4f


VGG_BLOCKS = {"vgg16": (2, 2, 3, 3, 3), "vgg19": (2, 2, 4, 4, 4)}
VGG_FILTERS = (64, 128, 256, 512, 512)


def expansion(backbone):
    return 4 if backbone in BOTTLENECK else 1
STAGE_FILTERS = (64, 128, 256, 512)
BN_EPS_ENCODER = 2e-5   # classification_models ResNet: BatchNormalization(epsilon=2e-5)
BN_EPS_DECODER = 1e-3   # Keras BatchNormalization default
BN_MOMENTUM = 0.99      # Keras default, used by both


def _he_uniform(rng, shape):
    kh, kw, ci, co = shape
    limit = np.sqrt(6.0 / (kh * kw * ci))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _glorot_uniform(rng, shape):
    kh, kw, ci, co = shape
    limit = np.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def _bn(P, name, c, scale=True):
    if scale:
        P[name + "/gamma"] = np.ones(c, np.float32)
    P[name + "/beta"] = np.zeros(c, np.float32)
    P[name + "/moving_mean"] = np.zeros(c, np.float32)
    P[name + "/moving_variance"] = np.ones(c, np.float32)


def init_unet_resnet(backbone="resnet34", in_ch=3, classes=1,
                     decoder_filters=(256, 128, 64, 32, 16), seed=42, decoder_block_type="upsampling"):
    """Random-init parameter set (he_uniform encoder, glorot_uniform decoder/head)."""
    rng = np.random.RandomState(seed)
    if backbone in VGG_BLOCKS:
        return _init_unet_vgg(rng, backbone, in_ch, classes, decoder_filters, decoder_block_type)
    units = RESNET_UNITS[backbone]
    P = OrderedDict()
    _bn(P, "bn_data", in_ch, scale=False)
    P["conv0/kernel"] = _he_uniform(rng, (7, 7, in_ch, 64))
    _bn(P, "bn0", 64)
    cin = 64
    ex = expansion(backbone)
    for s, (n_units, f) in enumerate(zip(units, STAGE_FILTERS), start=1):
        for u in range(1, n_units + 1):
            pre = "stage%d_unit%d_" % (s, u)
            _bn(P, pre + "bn1", cin)
            if ex == 1:
                P[pre + "conv1/kernel"] = _he_uniform(rng, (3, 3, cin, f))
                _bn(P, pre + "bn2", f)
                P[pre + "conv2/kernel"] = _he_uniform(rng, (3, 3, f, f))
            else:
                P[pre + "conv1/kernel"] = _he_uniform(rng, (1, 1, cin, f))
                _bn(P, pre + "bn2", f)
                P[pre + "conv2/kernel"] = _he_uniform(rng, (3, 3, f, f))
                _bn(P, pre + "bn3", f)
                P[pre + "conv3/kernel"] = _he_uniform(rng, (1, 1, f, f * ex))
            if u == 1:
                P[pre + "sc/kernel"] = _he_uniform(rng, (1, 1, cin, f * ex))
            cin = f * ex
    _bn(P, "bn1", cin)
    skip_ch = (STAGE_FILTERS[2] * ex, STAGE_FILTERS[1] * ex, STAGE_FILTERS[0] * ex, 64, 0)
    for i, f in enumerate(decoder_filters):
        pre = "decoder_stage%d_" % i
        if decoder_block_type == "transpose":
            # Transpose2D_block: Conv2DTranspose(f, 4x4, strides 2, 'same') kernel (kh, kw, out, in), BN, ReLU, concat, conv3x3
            lim = np.sqrt(6.0 / (16 * cin + 16 * f))
            P[pre + "upsample/kernel"] = rng.uniform(-lim, lim, size=(4, 4, f, cin)).astype(np.float32)
            _bn(P, pre + "bn1", f)
            P[pre + "conv2/kernel"] = _glorot_uniform(rng, (3, 3, f + skip_ch[i], f))
            _bn(P, pre + "bn2", f)
            cin = f
            continue
        P[pre + "conv1/kernel"] = _glorot_uniform(rng, (3, 3, cin + skip_ch[i], f))
        _bn(P, pre + "bn1", f)
        P[pre + "conv2/kernel"] = _glorot_uniform(rng, (3, 3, f, f))
        _bn(P, pre + "bn2", f)
        cin = f
    P["final_conv/kernel"] = _glorot_uniform(rng, (3, 3, cin, classes))
    P["final_conv/bias"] = np.zeros(classes, np.float32)
    return P


def _init_unet_vgg(rng, backbone, in_ch, classes, decoder_filters, decoder_block_type):
    """U-Net over keras.applications VGG16/19 (Conv2D 3x3 'same' + bias + ReLU, MaxPooling2D(2)); segmentation_models takes
    the last convolution of every block as skip and block5_pool as the decoder input (schemas/segmentation.raml:158-178
    defaults, backbone names README.md:587-589)."""
    if decoder_block_type != "upsampling":
        raise ValueError("VGG oracle: upsampling decoder only")
    P = OrderedDict()
    cin = in_ch
    for b, (n_conv, f) in enumerate(zip(VGG_BLOCKS[backbone], VGG_FILTERS), start=1):
        for c in range(1, n_conv + 1):
            P["block%d_conv%d/kernel" % (b, c)] = _glorot_uniform(rng, (3, 3, cin, f))
            P["block%d_conv%d/bias" % (b, c)] = np.zeros(f, np.float32)
            cin = f
    skip_ch = VGG_FILTERS[::-1]
    for i, f in enumerate(decoder_filters):
        pre = "decoder_stage%d_" % i
        P[pre + "conv1/kernel"] = _glorot_uniform(rng, (3, 3, cin + skip_ch[i], f))
        _bn(P, pre + "bn1", f)
        P[pre + "conv2/kernel"] = _glorot_uniform(rng, (3, 3, f, f))
        _bn(P, pre + "bn2", f)
        cin = f
    P["final_conv/kernel"] = _glorot_uniform(rng, (3, 3, cin, classes))
    P["final_conv/bias"] = np.zeros(classes, np.float32)
    return P


def _vgg_encoder(ctx, x_nhwc, backbone, stop_block=None):
    """``stop_block``: return the last convolution of that block (PSPNet's feature) instead of (block5_pool, skips)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    skips = []
    for b, n_conv in enumerate(VGG_BLOCKS[backbone], start=1):
        for c in range(1, n_conv + 1):
            x = F.relu(_conv(ctx, x, "block%d_conv%d" % (b, c), pad=1))
        ctx.tap("block%d_out" % b, x)
        if stop_block == b:
            return x
        skips.append(x)
        x = ctx.st(F.max_pool2d(x, kernel_size=2, stride=2))
    return x, skips[::-1]


def init_linknet_resnet(backbone="resnet34", in_ch=3, classes=1, decoder_filters=(None, None, None, None, 16), seed=42,
                        decoder_block_type="upsampling"):
    """Linknet over the same encoder (segmentation_models 0.2.1 ``Linknet(decoder_use_batchnorm=True)``, kwargs/defaults
    ``schemas/segmentation.raml:180-203``).  Decoder block i (``decoder_stage{i}``): 1x1 conv to in/4 -> BN -> ReLU ->
    UpSampling2D(2) -> 3x3 conv in/4 -> BN -> ReLU  (``decoder_block_type='transpose'``: Conv2DTranspose 4x4 s2 to in/4 -> BN
    -> ReLU instead of the pair) -> 1x1 conv to the skip's channel count (or ``decoder_filters[i]`` without a skip) -> BN ->
    ReLU -> Add(skip).  VGG encoders: block5_pool + the skips block5 / block4 / block3 / block2 last convolutions."""
    full = init_unet_resnet(backbone, in_ch, classes, seed=seed)
    P = OrderedDict((k, v) for k, v in full.items() if not (k.startswith("decoder_") or k.startswith("final_")))
    rng = np.random.RandomState(seed + 1)
    if backbone in VGG_BLOCKS:
        cin, skip_ch = 512, (512, 512, 256, 128, None)
    else:
        ex = expansion(backbone)
        cin = STAGE_FILTERS[3] * ex
        skip_ch = (STAGE_FILTERS[2] * ex, STAGE_FILTERS[1] * ex, STAGE_FILTERS[0] * ex, 64, None)
    for i in range(5):
        pre = "decoder_stage%d_" % i
        mid = cin // 4
        out = skip_ch[i] if skip_ch[i] is not None else decoder_filters[i]
        for j, (k, ci, co) in enumerate(((1, cin, mid), (3, mid, mid), (1, mid, out)), start=1):
            if j == 2 and decoder_block_type == "transpose":
                P[pre + "upsample/kernel"] = _glorot_uniform(rng, (4, 4, co, ci))        # Keras Conv2DTranspose: (kh, kw, out, in)
            else:
                P[pre + "conv%d/kernel" % j] = _glorot_uniform(rng, (k, k, ci, co))
            _bn(P, pre + "bn%d" % j, co)
        cin = out
    P["final_conv/kernel"] = _glorot_uniform(rng, (3, 3, cin, classes))
    P["final_conv/bias"] = np.zeros(classes, np.float32)
    return P


def init_fpn_resnet(backbone="resnet34", in_ch=3, classes=1, seed=42, pyramid_filters=256, segmentation_filters=128):
    """FPN over the same encoder: segmentation_models 0.2.1 ``FPN(upsample_rates=(2,2,2), last_upsample=4,
    interpolation='bilinear', use_batchnorm=True)`` (``schemas/segmentation.raml:180-203``); see fpn_resnet_forward."""
    full = init_unet_resnet(backbone, in_ch, classes, seed=seed)
    P = OrderedDict((k, v) for k, v in full.items() if not (k.startswith("decoder_") or k.startswith("final_")))
    rng = np.random.RandomState(seed + 2)
    if backbone in VGG_BLOCKS:       # pyramid over block5_pool and the skip layers block5 / block4 / block3 (last convolution of the block)
        level_ch = (VGG_FILTERS[4], VGG_FILTERS[4], VGG_FILTERS[3], VGG_FILTERS[2])
    else:
        ex = expansion(backbone)
        level_ch = (STAGE_FILTERS[3] * ex, STAGE_FILTERS[2] * ex, STAGE_FILTERS[1] * ex, STAGE_FILTERS[0] * ex)
    for i, c in enumerate(level_ch):
        pre = "fpn_stage%d_" % i
        P[pre + "lateral/kernel"] = _glorot_uniform(rng, (1, 1, c, pyramid_filters))
        P[pre + "lateral/bias"] = (rng.randn(pyramid_filters) * 0.01).astype(np.float32)
        P[pre + "segm1/kernel"] = _glorot_uniform(rng, (3, 3, pyramid_filters, segmentation_filters))
        _bn(P, pre + "segm1_bn", segmentation_filters)
        P[pre + "segm2/kernel"] = _glorot_uniform(rng, (3, 3, segmentation_filters, segmentation_filters))
        _bn(P, pre + "segm2_bn", segmentation_filters)
    P["fpn_final/kernel"] = _glorot_uniform(rng, (3, 3, 4 * segmentation_filters, 4 * segmentation_filters))
    _bn(P, "fpn_final_bn", 4 * segmentation_filters)
    P["final_conv/kernel"] = _glorot_uniform(rng, (3, 3, 4 * segmentation_filters, classes))
    P["final_conv/bias"] = np.zeros(classes, np.float32)
    return P


def resize_bilinear_tf1(x, f):
    """tf.image.resize_bilinear(align_corners=False) of TF 1.x by an integer factor on NCHW: src = dst / f, x0 = floor(src),
    x1 = min(x0 + 1, in - 1) - no half-pixel offset (torch's F.interpolate uses half-pixel centres and is NOT this).
    TF's lerp order: along x first (top, bottom), then along y."""
    if f == 1:
        return x
    n, c, h, w = x.shape
    yo, xo = torch.arange(h * f), torch.arange(w * f)
    y0, x0 = yo // f, xo // f
    fy = ((yo - y0 * f).to(x.dtype) / f).view(1, 1, -1, 1)
    fx = ((xo - x0 * f).to(x.dtype) / f).view(1, 1, 1, -1)
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    r0, r1 = x[:, :, y0], x[:, :, y1]
    top = r0[:, :, :, x0] + (r0[:, :, :, x1] - r0[:, :, :, x0]) * fx
    bot = r1[:, :, :, x0] + (r1[:, :, :, x1] - r1[:, :, :, x0]) * fx
    return top + (bot - top) * fy


DECODER_DROPOUT_SALT = 0x5D0D


def _spatial_dropout(y, rate, step):
    """SpatialDropout2D (training phase) with the device kernel's counter-based mask (stp_dropout_spatial): one decision per
    (sample, channel), index n * C + c, inverted scaling."""
    from .deeplab import dropout_mask
    n, c = y.shape[0], y.shape[1]
    keep = dropout_mask(step, DECODER_DROPOUT_SALT, n * c, rate).reshape(n, c, 1, 1)
    return y * torch.from_numpy(keep.astype(np.float32)) / (1.0 - rate)        # (the caller stores it: ctx.st)


def _resize(x, f, interpolation):
    return F.interpolate(x, scale_factor=f, mode="nearest") if (interpolation == "nearest" and f > 1) else resize_bilinear_tf1(x, f)


def fpn_resnet_forward(P, x_nhwc, backbone="resnet34", training=True, taps=None, dropout=None, step=1, interpolation="bilinear",
                       storage=None, grad_scale=1.0, accum64=False):
    """Pyramid over [encoder output, stage4/3/2 unit-1 relu1]: lateral Conv2D 1x1 (bias) + UpSampling2D(2) of the level above,
    two (Conv2D 3x3 no bias, BN, ReLU) per level; the maps resized to 1/4 resolution, concatenated finest first,
    Conv 3x3 + BN + ReLU (4 x 128 filters), Conv2D 3x3 to the classes, bilinear x4.  Returns (logits_nhwc, bn_updates).
    ``storage``: see _Ctx - stored on the device: the lateral convolution, the top-down sum (stp_upsample2x_add, in place), every
    resized slice of the concatenation, the class convolution (tap-channel form: _class_head) and the resized logits."""
    ctx = _Ctx(P, training, taps, storage, grad_scale, accum64)
    if backbone in VGG_BLOCKS:
        x, sk = _vgg_encoder(ctx, x_nhwc, backbone)
        levels = (x, sk[0], sk[1], sk[2])
    else:
        x, skips = _resnet_encoder(ctx, x_nhwc, backbone)
        levels = (x, skips["stage4_unit1_relu1"], skips["stage3_unit1_relu1"], skips["stage2_unit1_relu1"])
    m, pyramid = None, []
    for i, c in enumerate(levels):
        pre = "fpn_stage%d_" % i
        lat = _conv(ctx, c, pre + "lateral")
        if m is not None:
            lat = ctx.st(lat + F.interpolate(m, scale_factor=2, mode="nearest"))
        p = _bn_apply(ctx, _conv(ctx, lat, pre + "segm1", pad=1), pre + "segm1_bn", BN_EPS_DECODER, relu=True)
        p = _bn_apply(ctx, _conv(ctx, p, pre + "segm2", pad=1), pre + "segm2_bn", BN_EPS_DECODER, relu=True)
        ctx.tap(pre + "out", p)
        m = lat
        pyramid.append(p)
    cat = torch.cat([ctx.st(_resize(p, f, interpolation)) for p, f in zip(pyramid[::-1], (1, 2, 4, 8))], dim=1)
    y = _bn_apply(ctx, _conv(ctx, cat, "fpn_final", pad=1), "fpn_final_bn", BN_EPS_DECODER, relu=True)
    if dropout and training:
        y = ctx.st(_spatial_dropout(y, float(dropout), step))
    lo = _class_head(ctx, y)
    ctx.tap("final_conv", lo)
    return ctx.st(_resize(lo, 4, interpolation)).permute(0, 2, 3, 1).contiguous(), ctx.bn_updates


PSP_STAGE = {4: 2, 8: 3, 16: 4}     # downsample_factor -> the stage whose unit1_relu1 is the feature (schemas/segmentation.raml:228-230)


def init_pspnet_resnet(backbone="resnet34", in_ch=3, classes=1, seed=42, conv_filters=512, downsample_factor=8):
    """PSPNet (segmentation_models 0.2.1, ``schemas/segmentation.raml:225-249``): only the encoder up to the feature
    stage<s>_unit1_relu1 (1/downsample_factor resolution; default 1/8 = stage 3) exists; see pspnet_resnet_forward."""
    full = init_unet_resnet(backbone, in_ch, classes, seed=seed)
    st = PSP_STAGE[int(downsample_factor)]
    if backbone in VGG_BLOCKS:       # feature = the last convolution of block st + 1 (1/4: block3_conv3, 1/8: block4, 1/16: block5)
        keep = tuple("block%d_" % b for b in range(1, st + 2))
        c = VGG_FILTERS[st]
    else:
        keep = ("bn_data", "conv0", "bn0") + tuple("stage%d_" % i for i in range(1, st)) + ("stage%d_unit1_bn1" % st,)
        c = STAGE_FILTERS[st - 2] * expansion(backbone)
    P = OrderedDict((k, v) for k, v in full.items() if k.startswith(keep))
    rng = np.random.RandomState(seed + 3)
    for level in (1, 2, 3, 6):
        P["psp_level%d_conv/kernel" % level] = _glorot_uniform(rng, (1, 1, c, conv_filters))
        _bn(P, "psp_level%d_bn" % level, conv_filters)
    P["psp_final/kernel"] = _glorot_uniform(rng, (1, 1, c + 4 * conv_filters, 512))
    _bn(P, "psp_final_bn", 512)
    P["final_conv/kernel"] = _glorot_uniform(rng, (3, 3, 512, classes))
    P["final_conv/bias"] = np.zeros(classes, np.float32)
    return P


def pspnet_resnet_forward(P, x_nhwc, backbone="resnet34", training=True, taps=None, downsample_factor=8, dropout=None, step=1,
                          final_interpolation="bilinear", psp_pooling_type="avg", storage=None, grad_scale=1.0, accum64=False):
    """feature = stage3_unit1_relu1; for level in 1, 2, 3, 6: AveragePooling2D(size / level) -> Conv 1x1 (no bias) -> BN ->
    ReLU -> bilinear resize back; Concatenate([feature, l1, l2, l3, l6]); Conv 1x1 + BN + ReLU (512); Conv2D 3x3 to the
    classes; bilinear x8.  Returns (logits_nhwc, bn_updates).  ``storage``: see _Ctx - stored on the device: the pooled maps, every
    resized slice of the concatenation, the class convolution (tap-channel form: _class_head) and the resized logits."""
    ctx = _Ctx(P, training, taps, storage, grad_scale, accum64)
    if backbone in VGG_BLOCKS:
        f = _vgg_encoder(ctx, x_nhwc, backbone, stop_block=PSP_STAGE[int(downsample_factor)] + 1)
    else:
        feat = "stage%d_unit1_relu1" % PSP_STAGE[int(downsample_factor)]
        _, skips = _resnet_encoder(ctx, x_nhwc, backbone, stop_at=feat)
        f = skips[feat]
    parts = [f]
    for level in (1, 2, 3, 6):
        k = f.shape[2] // level
        p = ctx.st((F.max_pool2d if psp_pooling_type == "max" else F.avg_pool2d)(f, kernel_size=k, stride=k))
        p = _bn_apply(ctx, _conv(ctx, p, "psp_level%d_conv" % level), "psp_level%d_bn" % level, BN_EPS_DECODER, relu=True)
        ctx.tap("psp_level%d_out" % level, p)
        parts.append(ctx.st(resize_bilinear_tf1(p, k)))
    y = _bn_apply(ctx, _conv(ctx, torch.cat(parts, dim=1), "psp_final"), "psp_final_bn", BN_EPS_DECODER, relu=True)
    ctx.tap("psp_final_out", y)
    if dropout and training:
        y = ctx.st(_spatial_dropout(y, float(dropout), step))
    lo = _class_head(ctx, y)
    ctx.tap("final_conv", lo)
    return ctx.st(_resize(lo, int(downsample_factor), final_interpolation)).permute(0, 2, 3, 1).contiguous(), ctx.bn_updates


ENCODER_PREFIXES = ("bn_data", "conv0", "bn0", "stage", "bn1/", "block")


def trainable_names(P, freeze_encoder=False):
    """Names that receive gradients/updates.  Moving statistics never do."""
    out = []
    for k in P:
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            continue
        if freeze_encoder and k.startswith(ENCODER_PREFIXES):
            continue
        out.append(k)
    return out


def conv_param_count(P):
    return int(sum(v.size for k, v in P.items() if k.endswith("/kernel")))


class _StoreRound(torch.autograd.Function):
    """A tensor that the device path STORES in a 16-bit format: forward = round to that format; backward = round the incoming
    gradient the same way (the gradient of a stored tensor is itself a stored tensor on the device).  Straight-through otherwise."""

    @staticmethod
    def forward(fctx, x, dt, gscale=1.0):
        fctx.dt, fctx.gscale = dt, float(gscale)
        return x.to(dt).to(torch.float32)

    @staticmethod
    def backward(fctx, g):
        # (the fp16 build stores gradients TIMES its loss scale - a power of two, divided out by the optimizer: the rounding, and
        #  what falls below the format's range, happen at that scale)
        s = fctx.gscale
        return (g * s).to(fctx.dt).to(torch.float32) / s, None, None


class _Ctx:
    """``storage``: None = the fp32 oracle.  torch.bfloat16 / torch.float16 = the STORAGE-QUANTISED oracle: every tensor
    the HIP path keeps in HBM in that format - convolution outputs (after the fused residual add), BatchNormalization outputs, the
    pooled / upsampled tensors, the logits - and every gradient of such a tensor is rounded where the kernels round it, and the
    convolutions read rounded weight copies; statistics, accumulation, loss and optimizer stay fp32.  Used to hold the 16-bit
    training modes to a tight bar (tests/test_model_gpu.py); U-Net / Linknet / FPN / PSPNet over the ResNet (basic and bottleneck) and
    VGG encoders.  ``grad_scale``: the loss scale the stored GRADIENTS carry (fp16 build: 2^14 by default, backend.HipSegModel)."""

    def __init__(self, P, training, taps, storage=None, grad_scale=1.0, accum64=False):
        self.P = P
        self.training = training
        self.taps = taps
        self.bn_updates = OrderedDict()
        self.storage = storage
        self.grad_scale = float(grad_scale)
        # accum64: every convolution accumulates in float64 (then rounds to fp32, then to the storage format) - a SECOND, equally valid
        # evaluation of the same rounding points: the distance between the two oracles is the noise floor of a 16-bit step (which way
        # the rounding ties of ~10^6 stored values fall), the yardstick the device is held to (tests/test_model_gpu.py)
        self.accum64 = bool(accum64)

    def st(self, t):
        return t if self.storage is None else _StoreRound.apply(t, self.storage, self.grad_scale)

    def wq(self, w):
        """The 16-bit compute copy of an fp32 master weight (the weight gradient is a function of dY and x only: straight-through)."""
        if self.storage is None:
            return w
        return w + (w.detach().to(self.storage).to(torch.float32) - w.detach())

    def tap(self, name, t):
        if self.taps is not None:
            self.taps[name] = t.permute(0, 2, 3, 1)  # NHWC view
        return t


def _conv(ctx, x, name, stride=1, pad=0, store=True):
    # Keras HWIO -> torch OIHW ; explicit symmetric ZeroPadding2D + 'valid'
    w = ctx.wq(ctx.P[name + "/kernel"].permute(3, 2, 0, 1))
    b = ctx.P.get(name + "/bias")
    if ctx.accum64:
        y = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad).float()
    else:
        y = F.conv2d(x, w, b, stride=stride, padding=pad)
    return ctx.st(y) if store else y       # store=False: the epilogue adds a residual before the one rounding (caller stores)


def _conv_transpose(ctx, x, name):
    """Conv2DTranspose(4x4, strides 2, padding='same'), Keras kernel (kh, kw, out, in): the gradient of a stride-2 'same' convolution =
    torch padding 1.  Stored like any convolution output."""
    wt = ctx.wq(ctx.P[name + "/kernel"].permute(3, 2, 0, 1))           # (kh,kw,out,in) -> (in,out,kh,kw)
    return ctx.st(F.conv_transpose2d(x, wt, stride=2, padding=1))


def class_head_uses_taps(cin, classes):
    """segmentation_training_pipeline_amd/nets.py:_class_head - the condition under which the device evaluates ``final_conv`` of the
    FPN / PSPNet decoders in its tap-channel form."""
    return 18 * classes <= cin and cin >= 128


def _class_head(ctx, y, name="final_conv"):
    """``Conv2D(classes, 3x3, padding 1, bias)`` of the FPN / PSPNet decoders.  The fp32 oracle evaluates it directly.  The STORAGE-QUANTISED
    oracle follows the device's tap-channel form where the device uses it (graph.Plan.conv3x3_taps): a 1x1 convolution into 9 x classes
    tap channels that are STORED (one rounding per tap), then the shifted taps and the bias summed in fp32 and stored once more."""
    w = ctx.P[name + "/kernel"]                                        # HWIO
    classes = int(w.shape[3])
    if ctx.storage is None or not class_head_uses_taps(int(y.shape[1]), classes):
        return _conv(ctx, y, name, pad=1)
    wq = ctx.wq(w)
    n, _, h, wd = y.shape
    acc = None
    for kh in range(3):
        for kw in range(3):
            wt = wq[kh, kw].t().reshape(classes, -1, 1, 1)
            z = ctx.st(F.conv2d(y.double(), wt.double()).float() if ctx.accum64 else F.conv2d(y, wt))       # tap channel (kh, kw): stored
            zs = F.pad(z, (1, 1, 1, 1))[:, :, kh:kh + h, kw:kw + wd]                     # out[h, w] += z[h + kh - 1, w + kw - 1]
            acc = zs if acc is None else acc + zs
    return ctx.st(acc + ctx.P[name + "/bias"].view(1, -1, 1, 1))


def _bn_apply(ctx, x, name, eps, relu):
    """Keras BatchNormalization.  Training phase: biased batch variance for the
    normalisation; moving variance updated with the unbiased estimate
    (tf.nn.fused_batch_norm behaviour), momentum 0.99."""
    P = ctx.P
    c = x.shape[1]
    gamma = P.get(name + "/gamma")
    beta = P[name + "/beta"]
    if ctx.training:
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean.view(1, c, 1, 1)) ** 2).mean(dim=(0, 2, 3))
        n = x.numel() // c
        with torch.no_grad():
            mm = P[name + "/moving_mean"] * BN_MOMENTUM + mean * (1 - BN_MOMENTUM)
            mv = P[name + "/moving_variance"] * BN_MOMENTUM + var * (n / max(n - 1, 1)) * (1 - BN_MOMENTUM)
            ctx.bn_updates[name + "/moving_mean"] = mm.detach().clone()
            ctx.bn_updates[name + "/moving_variance"] = mv.detach().clone()
    else:
        mean = P[name + "/moving_mean"]
        var = P[name + "/moving_variance"]
    inv = torch.rsqrt(var + eps)
    scale = inv if gamma is None else inv * gamma
    y = (x - mean.view(1, c, 1, 1)) * scale.view(1, c, 1, 1) + beta.view(1, c, 1, 1)
    return ctx.st(F.relu(y) if relu else y)


def _resnet_encoder(ctx, x_nhwc, backbone, stop_at=None):
    """Pre-activation ResNet of classification_models up to the final bn1+relu; returns (x, skip tensors by layer name)."""
    units = RESNET_UNITS[backbone]
    x = x_nhwc.permute(0, 3, 1, 2)
    x = _bn_apply(ctx, x, "bn_data", BN_EPS_ENCODER, relu=False)
    ctx.tap("bn_data", x)
    x = _conv(ctx, x, "conv0", stride=2, pad=3)
    ctx.tap("conv0", x)
    x = _bn_apply(ctx, x, "bn0", BN_EPS_ENCODER, relu=True)
    skips = {"relu0": x}
    ctx.tap("relu0", x)
    x = ctx.st(F.max_pool2d(F.pad(x, (1, 1, 1, 1)), kernel_size=3, stride=2))  # ZeroPadding2D(1) + valid pool
    ctx.tap("pooling0", x)
    for s, (n_units, f) in enumerate(zip(units, STAGE_FILTERS), start=1):
        for u in range(1, n_units + 1):
            pre = "stage%d_unit%d_" % (s, u)
            stride = 2 if (u == 1 and s > 1) else 1
            a = _bn_apply(ctx, x, pre + "bn1", BN_EPS_ENCODER, relu=True)
            if u == 1:
                skips[pre + "relu1"] = a
                ctx.tap(pre + "relu1", a)
                if stop_at == pre + "relu1":
                    return None, skips
                shortcut = _conv(ctx, a, pre + "sc", stride=stride, pad=0)
            else:
                shortcut = x
            if expansion(backbone) == 1:
                y = _conv(ctx, a, pre + "conv1", stride=stride, pad=1)
                y = _bn_apply(ctx, y, pre + "bn2", BN_EPS_ENCODER, relu=True)
                y = _conv(ctx, y, pre + "conv2", stride=1, pad=1, store=False)
            else:   # bottleneck: the stride sits on the 3x3 convolution
                y = _conv(ctx, a, pre + "conv1")
                y = _bn_apply(ctx, y, pre + "bn2", BN_EPS_ENCODER, relu=True)
                y = _conv(ctx, y, pre + "conv2", stride=stride, pad=1)
                y = _bn_apply(ctx, y, pre + "bn3", BN_EPS_ENCODER, relu=True)
                y = _conv(ctx, y, pre + "conv3", store=False)
            x = ctx.st(y + shortcut)       # the residual Add rides in the convolution's epilogue: one rounding of the sum
            ctx.tap(pre + "out", x)
    x = _bn_apply(ctx, x, "bn1", BN_EPS_ENCODER, relu=True)
    ctx.tap("relu1", x)
    return x, skips


def linknet_resnet_forward(P, x_nhwc, backbone="resnet34", training=True, taps=None, storage=None, grad_scale=1.0, accum64=False):
    """Linknet (see init_linknet_resnet).  Returns (logits_nhwc, bn_updates).  ``storage``: see _Ctx - the Add() with the encoder
    feature is a tensor op of its own on the device (stp_add_inplace): the sum is stored."""
    ctx = _Ctx(P, training, taps, storage, grad_scale, accum64)
    if backbone in VGG_BLOCKS:
        x, sk = _vgg_encoder(ctx, x_nhwc, backbone)
        skips = {"s%d" % i: t for i, t in enumerate(sk)}
        skip_names = ("s0", "s1", "s2", "s3", None)
    else:
        x, skips = _resnet_encoder(ctx, x_nhwc, backbone)
        skip_names = ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0", None)
    for i in range(5):
        pre = "decoder_stage%d_" % i
        x = _bn_apply(ctx, _conv(ctx, x, pre + "conv1"), pre + "bn1", BN_EPS_DECODER, relu=True)
        if pre + "upsample/kernel" in P:      # Conv2DTranspose(4x4, strides 2, padding='same') = torch padding 1
            x = _conv_transpose(ctx, x, pre + "upsample")
        else:
            x = F.interpolate(x, scale_factor=2, mode="nearest")      # (folded into the convolution's gather: no tensor of its own)
            x = _conv(ctx, x, pre + "conv2", pad=1)
        x = _bn_apply(ctx, x, pre + "bn2", BN_EPS_DECODER, relu=True)
        x = _bn_apply(ctx, _conv(ctx, x, pre + "conv3"), pre + "bn3", BN_EPS_DECODER, relu=True)
        if skip_names[i] is not None:
            x = ctx.st(x + skips[skip_names[i]])
        ctx.tap(pre + "out", x)
    x = _conv(ctx, x, "final_conv", pad=1)
    return x.permute(0, 2, 3, 1).contiguous(), ctx.bn_updates


def unet_resnet_forward(P, x_nhwc, backbone="resnet34", training=True, taps=None,
                        decoder_filters=(256, 128, 64, 32, 16), storage=None, grad_scale=1.0, accum64=False):
    """P: dict name -> torch tensor (Keras layouts).  x_nhwc: [N,H,W,C] float32 raw 0..255.
    Returns (logits_nhwc, bn_updates).  Probabilities = sigmoid(logits).  ``storage``: see _Ctx."""
    ctx = _Ctx(P, training, taps, storage, grad_scale, accum64)
    if backbone in VGG_BLOCKS:
        x, sk = _vgg_encoder(ctx, x_nhwc, backbone)
        skips = {"s%d" % i: t for i, t in enumerate(sk)}
        skip_names = ("s0", "s1", "s2", "s3", "s4")
    else:
        x, skips = _resnet_encoder(ctx, x_nhwc, backbone)
        skip_names = ("stage4_unit1_relu1", "stage3_unit1_relu1", "stage2_unit1_relu1", "relu0", None)
    for i, f in enumerate(decoder_filters):
        pre = "decoder_stage%d_" % i
        if pre + "upsample/kernel" in P:
            # Conv2DTranspose(4x4, strides 2, padding='same'): the gradient of a stride-2 'same' conv = torch padding 1
            x = _conv_transpose(ctx, x, pre + "upsample")
            x = _bn_apply(ctx, x, pre + "bn1", BN_EPS_DECODER, relu=True)
            if skip_names[i] is not None:
                x = torch.cat([x, skips[skip_names[i]]], dim=1)
            x = _conv(ctx, x, pre + "conv2", pad=1)
            x = _bn_apply(ctx, x, pre + "bn2", BN_EPS_DECODER, relu=True)
            ctx.tap(pre + "relu2", x)
            continue
        x = ctx.st(F.interpolate(x, scale_factor=2, mode="nearest"))  # UpSampling2D(2) (its high-resolution gradient is a stored tensor)
        if skip_names[i] is not None:
            x = torch.cat([x, skips[skip_names[i]]], dim=1)
        x = _conv(ctx, x, pre + "conv1", pad=1)
        x = _bn_apply(ctx, x, pre + "bn1", BN_EPS_DECODER, relu=True)
        x = _conv(ctx, x, pre + "conv2", pad=1)
        x = _bn_apply(ctx, x, pre + "bn2", BN_EPS_DECODER, relu=True)
        ctx.tap(pre + "relu2", x)
    x = _conv(ctx, x, "final_conv", pad=1)
    logits = x.permute(0, 2, 3, 1).contiguous()
    return logits, ctx.bn_updates


def to_torch(P, requires_grad_names=()):
    out = OrderedDict()
    req = set(requires_grad_names)
    for k, v in P.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).clone()
        if k in req:
            t.requires_grad_(True)
        out[k] = t
    return out
