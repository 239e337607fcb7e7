"""Network definitions executed against a :class:`graph.Plan`.

``unet_resnet`` is the graph that the reference obtains at ``segmentation_pipeline/segmentation.py:155``
from ``segmentation_models.Unet(backbone_name='resnet18|34', ...)`` with the defaults of
``schemas/segmentation.raml:158-178`` (decoder_block_type 'upsampling', decoder_filters
(256,128,64,32,16), BatchNorm in the decoder) over the classification_models pre-activation
ResNet.  Parameter names follow those packages' layer names so weight files map 1:1.

Fusions expressed here (none changes the arithmetic of the Keras graph):
  * ZeroPadding2D + 'valid' conv            -> conv with symmetric pad
  * Add() after conv2                        -> residual operand of the conv epilogue
  * UpSampling2D(2) + Concatenate + conv     -> one conv with a two-source gather
  * batch statistics of a BatchNormalization -> reduced in the epilogue of the conv that feeds it (bn_stats=True)
"""

RESNET_UNITS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3),
                "resnet152": (3, 8, 36, 3)}
BOTTLENECK = ("resnet50", "resnet101", "resnet152")   # residual_bottleneck_block: 1x1 f, 3x3 f (carries the stride), 1x1 4f
STAGE_FILTERS = (64, 128, 256, 512)
BN_EPS_ENCODER = 2e-5
BN_EPS_DECODER = 1e-3

ENCODER_PREFIXES = ("bn_data", "conv0", "bn0", "stage", "bn1", "block", "Conv", "expanded_conv", "entry_flow", "middle_flow", "exit_flow")
VGG_BLOCKS = {"vgg16": (2, 2, 3, 3, 3), "vgg19": (2, 2, 4, 4, 4)}      # keras.applications: 3x3 'same' convs + ReLU per block
VGG_FILTERS = (64, 128, 256, 512, 512)


def known_backbones():
    return sorted(RESNET_UNITS) + sorted(VGG_BLOCKS) + ["mobilenetv2", "xception"]


def _resnet_encoder(plan, backbone, H, W, in_ch, stop_stage=None):
    """Pre-activation ResNet of classification_models; returns (bn1+relu output, relu0, {stage: unit-1 relu1}).
    ``stop_stage``: build only up to that stage's unit-1 relu1 (PSPNet cuts the backbone there); the first value is None."""
    if (H % 32 or W % 32) and stop_stage is None:
        raise ValueError("input height/width must be divisible by 32")
    units = RESNET_UNITS[backbone]
    ex = 4 if backbone in BOTTLENECK else 1
    img = plan.input_u8("image", H, W, in_ch)
    x = plan.input_bn("bn_data", img, BN_EPS_ENCODER)
    x = plan.conv("conv0", x, 64, 7, stride=2, pad=3, bn_stats=True)
    relu0 = x = plan.bn("bn0", x, BN_EPS_ENCODER, relu=True)
    x = plan.maxpool("pooling0", x)
    taps = {}
    for s, (n_units, f) in enumerate(zip(units, STAGE_FILTERS), start=1):
        for u in range(1, n_units + 1):
            pre = "stage%d_unit%d_" % (s, u)
            stride = 2 if (u == 1 and s > 1) else 1
            a = plan.bn(pre + "bn1", x, BN_EPS_ENCODER, relu=True)
            if u == 1:
                taps[s] = a
                if s == stop_stage:
                    return None, relu0, taps
                # (bottleneck units: the shortcut is issued AFTER conv1, so in the backward pass its data gradient arrives FIRST: the
                #  stride-2 scatter then writes into a fresh buffer - no read of what conv1 accumulated, no BatchNormalization input - and
                #  conv1's dense 1x1 launch completes the gradient in its epilogue)
                if ex == 1:
                    shortcut = plan.conv(pre + "sc", a, f * ex, 1, stride=stride, pad=0)
            else:
                shortcut = x
            if ex == 1:
                # (a projection shortcut reads the same tensor: its data gradient rides in conv1's launch - stride 2: parity-class / space-to-
                #  depth form; stride 1, the first unit of stage 1: the centre tap of a second source on the halo kernel)
                y = plan.conv(pre + "conv1", a, f, 3, stride=stride, pad=1, bn_stats=True, fold_shortcut=shortcut if u == 1 else None)
                y = plan.bn(pre + "bn2", y, BN_EPS_ENCODER, relu=True)
                x = plan.conv(pre + "conv2", y, f, 3, stride=1, pad=1, residual=shortcut, bn_stats=True)
            else:
                y = plan.conv(pre + "conv1", a, f, 1, bn_stats=True)
                if u == 1:
                    shortcut = plan.conv(pre + "sc", a, f * ex, 1, stride=stride, pad=0)
                y = plan.bn(pre + "bn2", y, BN_EPS_ENCODER, relu=True)
                y = plan.conv(pre + "conv2", y, f, 3, stride=stride, pad=1, bn_stats=True)
                y = plan.bn(pre + "bn3", y, BN_EPS_ENCODER, relu=True)
                x = plan.conv(pre + "conv3", y, f * ex, 1, residual=shortcut, bn_stats=True)
    x = plan.bn("bn1", x, BN_EPS_ENCODER, relu=True)
    return x, relu0, taps


def _vgg_encoder(plan, backbone, H, W, in_ch, stop_block=None):
    """keras.applications VGG16/VGG19 (include_top=False) fed with raw pixels; returns (block5_pool, the last conv of each
    block = segmentation_models' skip layers block5_conv3 ... block1_conv2, deepest first).  ``stop_block``: build only up to the
    last convolution of that block and return it (PSPNet's feature layer)."""
    if H % 32 or W % 32:
        raise ValueError("input height/width must be divisible by 32")
    img = plan.input_u8("image", H, W, in_ch)
    x = plan.input_cast("input_cast", img)
    skips = []
    for b, (n_conv, f) in enumerate(zip(VGG_BLOCKS[backbone], VGG_FILTERS), start=1):
        for c in range(1, n_conv + 1):
            x = plan.conv("block%d_conv%d" % (b, c), x, f, 3, pad=1, bias=True, relu=True)
        if stop_block == b:
            return x
        skips.append(x)
        x = plan.maxpool2("block%d_pool" % b, x)
    return x, skips[::-1]


def _class_head(plan, y, classes):
    """``final_conv`` = Conv2D(classes, 3x3, padding 1, bias) of the FPN / PSPNet decoders.  With few classes over a wide feature map
    (512 -> 3 / 20) the per-tap kernel reads the input nine times for a handful of output channels: the tap-channel form
    (Plan.conv3x3_taps) reads it once.  STP_TAPSUM=0: the plain launch."""
    import os
    if 18 * classes <= y.C and y.C >= 128 and os.environ.get("STP_TAPSUM", "1") != "0":
        return plan.conv3x3_taps("final_conv", y, classes, bias=True)
    return plan.conv("final_conv", y, classes, 3, pad=1, bias=True)


def _head(plan, x, H, W, classes, loss, with_loss):
    logits = plan.conv("final_conv", x, classes, 3, pad=1, bias=True)
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)       # {0,1} for the sigmoid head, class index for the softmax head
        if classes == 1:
            plan.sigmoid_loss(logits, target, *loss)
        else:
            plan.softmax_loss(logits, target, loss[0], loss[1])
    return logits


def unet_resnet(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=(256, 128, 64, 32, 16),
                loss=(1.0, 1.0), with_loss=True, decoder_block_type="upsampling"):
    """Declares inputs 'image' (uint8 NHWC) and 'mask' (uint8 NHW1); returns the logits tensor.
    ``decoder_block_type``: 'upsampling' (UpSampling2D + concat + 2 x conv3x3) or 'transpose' (Conv2DTranspose 4x4 s2 ->
    BN -> ReLU -> concat -> conv3x3; segmentation_models' Transpose2D_block, schemas/segmentation.raml:166-169)."""
    if backbone in VGG_BLOCKS:
        x, skips = _vgg_encoder(plan, backbone, H, W, in_ch)        # five skips: every decoder stage concatenates one
    else:
        x, relu0, taps = _resnet_encoder(plan, backbone, H, W, in_ch)
        skips = (taps[4], taps[3], taps[2], relu0, None)
    for i, f in enumerate(decoder_filters):
        pre = "decoder_stage%d_" % i
        if decoder_block_type == "transpose":
            x = plan.conv(pre + "upsample", x, f, 4, transpose=True, bn_stats=True)
            x = plan.bn(pre + "bn1", x, BN_EPS_DECODER, relu=True)
            x = plan.conv(pre + "conv2", x, f, 3, pad=1, src1=skips[i], bn_stats=True)
            x = plan.bn(pre + "bn2", x, BN_EPS_DECODER, relu=True)
            continue
        x = plan.conv(pre + "conv1", x, f, 3, pad=1, src1=skips[i], upsample=True, bn_stats=True)
        x = plan.bn(pre + "bn1", x, BN_EPS_DECODER, relu=True)
        x = plan.conv(pre + "conv2", x, f, 3, pad=1, bn_stats=True)
        x = plan.bn(pre + "bn2", x, BN_EPS_DECODER, relu=True)
    return _head(plan, x, H, W, classes, loss, with_loss)


def linknet_resnet(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=(None, None, None, None, 16),
                   loss=(1.0, 1.0), with_loss=True, decoder_block_type="upsampling"):
    """segmentation_models 0.2.1 ``Linknet(decoder_use_batchnorm=True)`` (``schemas/segmentation.raml:180-203``): per decoder
    stage 1x1 conv to in/4, then UpSampling2D(2) folded into the 3x3 conv's gather ('upsampling') or Conv2DTranspose 4x4 s2
    ('transpose', ``schemas/segmentation.raml:166-169``), 1x1 conv to the skip's channels, each followed by BN+ReLU, then
    Add(skip).  VGG encoders: block5_pool + four skips (the last convolutions of block5 / 4 / 3 / 2)."""
    if backbone in VGG_BLOCKS:
        x, sk = _vgg_encoder(plan, backbone, H, W, in_ch)
        skips = (sk[0], sk[1], sk[2], sk[3], None)
    else:
        x, relu0, taps = _resnet_encoder(plan, backbone, H, W, in_ch)
        skips = (taps[4], taps[3], taps[2], relu0, None)
    for i in range(5):
        pre = "decoder_stage%d_" % i
        mid = x.C // 4
        out = skips[i].C if skips[i] is not None else int(decoder_filters[i])
        x = plan.bn(pre + "bn1", plan.conv(pre + "conv1", x, mid, 1, bn_stats=True), BN_EPS_DECODER, relu=True)
        if decoder_block_type == "transpose":
            x = plan.conv(pre + "upsample", x, mid, 4, transpose=True, bn_stats=True)
        else:
            x = plan.conv(pre + "conv2", x, mid, 3, pad=1, upsample=True, bn_stats=True)
        x = plan.bn(pre + "bn2", x, BN_EPS_DECODER, relu=True)
        x = plan.bn(pre + "bn3", plan.conv(pre + "conv3", x, out, 1, bn_stats=True), BN_EPS_DECODER, relu=True)
        if skips[i] is not None:
            x = plan.add(pre + "add", x, skips[i])
    return _head(plan, x, H, W, classes, loss, with_loss)


def fpn_resnet(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=None, loss=(1.0, 1.0), with_loss=True,
               pyramid_block_filters=256, segmentation_block_filters=128, last_upsample=4, dropout=None, interpolation="bilinear"):
    """segmentation_models 0.2.1 ``FPN(..., upsample_rates=(2,2,2), interpolation='bilinear', use_batchnorm=True)``
    (``schemas/segmentation.raml:180-203``).  Pyramid over [encoder output, stage4/3/2 unit1 relu1]: 1x1 lateral conv (+ the
    2x nearest upsampling of the level above), two conv3x3+BN+ReLU segmentation convs per level; the four maps are resized
    to 1/4 resolution (TF 1.x bilinear) and concatenated, conv3x3+BN+ReLU, conv3x3 to the classes, bilinear x4."""
    if last_upsample != 4:
        raise ValueError("the HIP FPN pyramid ends at 1/4 resolution: last_upsample must be 4")
    if backbone in VGG_BLOCKS:      # block5_pool + the skip layers block5_conv3 / block4_conv3 / block3_conv3 (README.md:587-589 backbones)
        x, sk = _vgg_encoder(plan, backbone, H, W, in_ch)
        levels = (x, sk[0], sk[1], sk[2])
    else:
        x, relu0, taps = _resnet_encoder(plan, backbone, H, W, in_ch)
        levels = (x, taps[4], taps[3], taps[2])
    pf, sf = int(pyramid_block_filters), int(segmentation_block_filters)
    m, pyramid = None, []
    for i, c in enumerate(levels):
        pre = "fpn_stage%d_" % i
        lat = plan.conv(pre + "lateral", c, pf, 1, bias=True)
        if m is not None:
            lat = plan.upsample_add(pre + "add", lat, m)
        p = plan.bn(pre + "segm1_bn", plan.conv(pre + "segm1", lat, sf, 3, pad=1, bn_stats=True), BN_EPS_DECODER, relu=True)
        p = plan.bn(pre + "segm2_bn", plan.conv(pre + "segm2", p, sf, 3, pad=1, bn_stats=True), BN_EPS_DECODER, relu=True)
        m = lat
        pyramid.append(p)
    near = interpolation == "nearest"
    cat = plan.concat_resize("fpn_concat", [(pyramid[3], 1), (pyramid[2], 2), (pyramid[1], 4), (pyramid[0], 8)], nearest=near)
    y = plan.bn("fpn_final_bn", plan.conv("fpn_final", cat, sf * 4, 3, pad=1, bn_stats=True), BN_EPS_DECODER, relu=True)
    if dropout:       # SpatialDropout2D between the final block and the class convolution (segmentation_models 0.2.1 fpn builder)
        y = plan.dropout("fpn_dropout", y, float(dropout), DECODER_DROPOUT_SALT, spatial=True)
    lo = _class_head(plan, y, classes)
    logits = plan.resize("logits", lo, 4, nearest=near)
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)
        (plan.sigmoid_loss if classes == 1 else plan.softmax_loss)(logits, target, *loss)
    return logits


def pspnet_resnet(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=None, loss=(1.0, 1.0), with_loss=True,
                  downsample_factor=8, psp_conv_filters=512, dropout=None, final_interpolation="bilinear", psp_pooling_type="avg"):
    """segmentation_models 0.2.1 ``PSPNet(downsample_factor=8, psp_conv_filters=512, psp_pooling_type='avg', use_batchnorm=True,
    final_interpolation='bilinear')`` (``schemas/segmentation.raml:225-249``): the backbone is cut at the 1/8 feature
    (stage3_unit1_relu1; 1/4: stage2, 1/16: stage4); pyramid pooling levels 1, 2, 3, 6 = AveragePooling2D(size/level) ->
    Conv 1x1 + BN + ReLU -> bilinear resize back, concatenated with the feature; Conv 1x1 + BN + ReLU; Conv 3x3 to the classes;
    bilinear x downsample_factor."""
    stage = {4: 2, 8: 3, 16: 4}[int(downsample_factor)]
    if backbone in VGG_BLOCKS:      # feature = last convolution of block3 (1/4), block4 (1/8) or block5 (1/16)
        f = _vgg_encoder(plan, backbone, H, W, in_ch, stop_block=stage + 1)
    else:
        _, _, taps = _resnet_encoder(plan, backbone, H, W, in_ch, stop_stage=stage)
        f = taps[stage]
    if f.H != f.W or f.H % 6:
        raise ValueError("PSPNet needs a square input whose 1/%d feature map is divisible by 6 (got %dx%d)" % (downsample_factor, f.H, f.W))
    import os
    parts = [(f, 1)]
    levels = (1, 2, 3, 6)
    if psp_pooling_type == "max":
        pooled = [plan.maxpool_k("psp_level%d_pool" % level, f, f.H // level) for level in levels]
    else:         # (the four average poolings of the feature map: one pass over it and one over its gradient where the windows nest)
        pooled = plan.avgpool_pyramid(["psp_level%d_pool" % level for level in levels], f, [f.H // level for level in levels])
    for level, p in zip(levels, pooled):
        k = f.H // level
        pre = "psp_level%d_" % level
        p = plan.bn(pre + "bn", plan.conv(pre + "conv", p, int(psp_conv_filters), 1, bn_stats=True), BN_EPS_DECODER, relu=True)
        parts.append((p, k))
    F_ = int(psp_conv_filters)
    if os.environ.get("STP_PSP_SPLIT", "1") != "0" and f.C % 8 == 0 and F_ % 8 == 0:
        # Conv2D(512, 1x1)(Concatenate([f, resize(p1), resize(p2), resize(p3), resize(p6)])) WITHOUT the concatenation (round 6): a 1x1
        # convolution is a per-pixel linear map and commutes with the bilinear resize, so
        #     psp_final(cat) = W[:, :C] f  +  sum over the levels of resize(W[:, level's columns] p_level)
        # - the level terms are 1x1 convolutions of 1x1 ... 6x6 maps, their resized sum (stp_upsample_sum) enters the feature convolution
        # as its residual operand.  Same parameter (`psp_final/kernel`, Keras (1, 1, C + 4 F, 512)), same function; the 2560-channel tensor
        # (377 MB at 8 x 96 x 96), its gradient and 80 % of the head's FLOP are gone.  STP_PSP_SPLIT=0: the concatenated form.
        Ct = f.C + 4 * F_
        zs = [(plan.conv("psp_final_level%d" % level, p, 512, 1, param_name="psp_final", param_cols=(f.C + i * F_, Ct), flops_as=0.0), k)
              for i, (level, (p, k)) in enumerate(zip((1, 2, 3, 6), parts[1:]))]
        r = plan.upsample_sum("psp_pyramid_sum", zs)
        # (algorithmic FLOP = the reference layer's: Conv2D(512, 1x1) over the C + 4 F concatenated channels)
        y = plan.conv("psp_final", f, 512, 1, residual=r, bn_stats=True, param_cols=(0, Ct), flops_as=2.0 * plan.N * f.H * f.W * 512 * Ct)
    else:
        cat = plan.concat_resize("psp_concat", parts)
        y = plan.conv("psp_final", cat, 512, 1, bn_stats=True)
    y = plan.bn("psp_final_bn", y, BN_EPS_DECODER, relu=True)
    if dropout:       # SpatialDropout2D between the final block and the class convolution (segmentation_models 0.2.1 psp builder)
        y = plan.dropout("psp_dropout", y, float(dropout), DECODER_DROPOUT_SALT, spatial=True)
    lo = _class_head(plan, y, classes)
    logits = plan.resize("logits", lo, int(downsample_factor), nearest=final_interpolation == "nearest")
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)
        (plan.sigmoid_loss if classes == 1 else plan.softmax_loss)(logits, target, *loss)
    return logits


# (filters, stride, expansion, block_id, skip_connection, rate): segmentation_pipeline/impl/deeplab/model.py:395-431
MOBILENETV2_BLOCKS = [(16, 1, 1, 0, False, 1), (24, 2, 6, 1, False, 1), (24, 1, 6, 2, True, 1), (32, 2, 6, 3, False, 1), (32, 1, 6, 4, True, 1),
                      (32, 1, 6, 5, True, 1), (64, 1, 6, 6, False, 1), (64, 1, 6, 7, True, 2), (64, 1, 6, 8, True, 2), (64, 1, 6, 9, True, 2),
                      (96, 1, 6, 10, False, 2), (96, 1, 6, 11, True, 2), (96, 1, 6, 12, True, 2), (160, 1, 6, 13, False, 2),
                      (160, 1, 6, 14, True, 4), (160, 1, 6, 15, True, 4), (320, 1, 6, 16, False, 4)]
def deeplab_logits_name(classes):
    """impl/deeplab/model.py:494-497: the class convolution is 'logits_semantic' for the 21 PASCAL-VOC classes (so that the published
    weights load by name) and 'custom_logits_semantic' for any other count."""
    return "logits_semantic" if classes == 21 else "custom_logits_semantic"


DEEPLAB_DROPOUT_SALT = 0x0D0D
DECODER_DROPOUT_SALT = 0x5D0D      # FPN / PSPNet `dropout` (SpatialDropout2D)


def deeplab_mobilenetv2(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=None, loss=(1.0, 1.0), with_loss=True):
    """The reference's in-tree DeepLabV3+ (``segmentation_pipeline/impl/deeplab/model.py:281-519``, registered as
    ``DeepLabV3`` at ``segmentation.py:31-33``), MobileNetV2 branch (alpha 1, output stride 8):
    Conv 3x3 s2 + BN + ReLU6 (:383-390); 17 inverted residual blocks = 1x1 expand, depthwise 3x3 (stride / atrous rate), 1x1
    project, each with BN(eps 1e-3, momentum 0.999) and ReLU6 except after the projection, optional Add (:236-279, :392-431);
    ASPP: image pooling branch (global average -> 1x1 -> BN -> ReLU -> bilinear back) and 1x1 branch (:438-450), concat, 1x1
    projection + BN + ReLU + Dropout(0.1) (:456-461); 1x1 convolution to the classes WITH the activation, then align-corners
    bilinear upsampling of the probabilities to the input size (:485-486).  The loss therefore works on probabilities."""
    if backbone != "mobilenetv2":
        raise ValueError("Unknown backbone")        # (the xception branch is not built)
    if not 1 <= classes <= 32:
        raise ValueError("the HIP DeepLabV3 trains the 1-class sigmoid head and 2..32-class softmax heads")
    if H != W:
        raise ValueError("DeepLabV3 needs a square input (any size: the reference sizes its layers with ceil(input / OS), model.py:440-445)")
    mob = dict(momentum=0.999)
    img = plan.input_u8("image", H, W, in_ch)
    x = plan.input_cast("input_cast", img)
    x = plan.bn("Conv_BN", plan.conv("Conv", x, 32, 3, stride=2, same_tf=True, bn_stats=True), 1e-3, relu=2, **mob)
    for filters, stride, exp, bid, skip, rate in MOBILENETV2_BLOCKS:
        pre = "expanded_conv_%d_" % bid if bid else "expanded_conv_"
        inp = x
        if bid:
            x = plan.bn(pre + "expand_BN", plan.conv(pre + "expand", x, exp * x.C, 1, bn_stats=True), 1e-3, relu=2, **mob)
        x = plan.bn(pre + "depthwise_BN", plan.dwconv(pre + "depthwise", x, 3, stride=stride, dilation=rate), 1e-3, relu=2, **mob)
        x = plan.bn(pre + "project_BN", plan.conv(pre + "project", x, filters, 1, bn_stats=True), 1e-3, relu=0, **mob)
        if skip:
            x = plan.add(pre + "add", x, inp)
    b4 = plan.avgpool("image_pooling_pool", x, x.H)
    b4 = plan.bn("image_pooling_BN", plan.conv("image_pooling", b4, 256, 1, bn_stats=True), 1e-5, relu=1)
    b0 = plan.bn("aspp0_BN", plan.conv("aspp0", x, 256, 1, bn_stats=True), 1e-5, relu=1)
    # BilinearUpsampling of the 1x1 map is a broadcast under either bilinear convention: the integer-factor resize writes it
    # straight into its half of the Concatenate
    cat = plan.concat_resize("aspp_concat", [(b4, x.H), (b0, 1)])
    y = plan.bn("concat_projection_BN", plan.conv("concat_projection", cat, 256, 1, bn_stats=True), 1e-5, relu=1)
    y = plan.dropout("dropout", y, 0.1, DEEPLAB_DROPOUT_SALT)
    z = plan.conv(deeplab_logits_name(classes), y, classes, 1, bias=True)
    p_lo = plan.sigmoid_act("probs_lo", z)
    probs = plan.resize_ac("logits", p_lo, H, W)          # named like the other heads' output tensor; holds PROBABILITIES
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)
        plan.prob_loss(probs, target, loss[0], loss[1])
    else:
        plan.probs_out(probs)
    return probs


def _sepconv_bn(plan, x, filters, prefix, stride=1, rate=1, depth_activation=False, eps=1e-3):
    """model.py:110-147 ``SepConv_BN``: [ReLU] -> DepthwiseConv2D 3x3 (stride / atrous rate; explicit symmetric padding + 'valid'
    when strided) -> BN [-> ReLU] -> Conv2D 1x1 -> BN [-> ReLU]."""
    if not depth_activation:
        x = plan.relu(prefix + "_relu", x)
    x = plan.dwconv(prefix + "_depthwise", x, 3, stride=stride, dilation=rate, explicit_pad=stride != 1)
    x = plan.bn(prefix + "_depthwise_BN", x, eps, relu=1 if depth_activation else 0)
    x = plan.conv(prefix + "_pointwise", x, filters, 1, bn_stats=True)
    return plan.bn(prefix + "_pointwise_BN", x, eps, relu=1 if depth_activation else 0)


def _xception_block(plan, x, depth_list, prefix, skip_type, stride, rate=1, depth_activation=False, return_skip=False):
    """model.py:182-218 ``_xception_block``: three SepConv_BN (the last one strided), skip = the output of the second;
    shortcut 'conv' (1x1 strided convolution + BN, added), 'sum' (the input added) or 'none'."""
    inputs, r, skip = x, x, None
    for i in range(3):
        r = _sepconv_bn(plan, r, depth_list[i], prefix + "_separable_conv%d" % (i + 1), stride=stride if i == 2 else 1, rate=rate,
                        depth_activation=depth_activation)
        if i == 1:
            skip = r
    if skip_type == "conv":
        sc = plan.bn(prefix + "_shortcut_BN", plan.conv(prefix + "_shortcut", inputs, depth_list[-1], 1, stride=stride, bn_stats=True), 1e-3,
                     relu=0)
        out = plan.add(prefix + "_add", r, sc)
    elif skip_type == "sum":
        out = plan.add(prefix + "_add", r, inputs)
    else:
        out = r
    return (out, skip) if return_skip else out


def deeplab_xception(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=None, loss=(1.0, 1.0), with_loss=True, OS=16):
    """The reference's in-tree DeepLabV3+ over the modified Xception (``impl/deeplab/model.py:338-379`` entry / middle / exit flow,
    ``:436-469`` ASPP with three atrous SepConv branches, ``:471-491`` decoder with ``feature_projection0``): entry conv 3x3 s2 +
    conv 3x3, blocks [128]x3 s2 (conv skip), [256]x3 s2 (conv skip; its second SepConv is the decoder skip), [728]x3 (stride 2 at
    OS 16); 16 middle units [728]x3 (sum skip); exit blocks [728, 1024, 1024] (conv skip) and [1536, 1536, 2048] (no skip, activations
    inside); ASPP = image pooling + 1x1 + SepConv rates (6, 12, 18) [(12, 24, 36) at OS 8]; projection + Dropout(0.1); decoder:
    align-corners bilinear to 1/4, concat with the 48-channel projection of the skip, two SepConv 256; class convolution WITH the
    activation; align-corners bilinear upsampling of the probabilities."""
    if not 1 <= classes <= 32:
        raise ValueError("the HIP DeepLabV3 trains the 1-class sigmoid head and 2..32-class softmax heads")
    if OS not in (8, 16) or H != W:
        raise ValueError("DeepLabV3 / xception needs a square input and output stride 8 or 16")
    b3_stride, mid_rate, exit_rates, aspp_rates = (1, 2, (2, 4), (12, 24, 36)) if OS == 8 else (2, 1, (1, 2), (6, 12, 18))
    img = plan.input_u8("image", H, W, in_ch)
    x = plan.input_cast("input_cast", img)
    x = plan.bn("entry_flow_conv1_1_BN", plan.conv("entry_flow_conv1_1", x, 32, 3, stride=2, same_tf=True, bn_stats=True), 1e-3, relu=1)
    x = plan.bn("entry_flow_conv1_2_BN", plan.conv("entry_flow_conv1_2", x, 64, 3, pad=1, bn_stats=True), 1e-3, relu=1)
    x = _xception_block(plan, x, [128, 128, 128], "entry_flow_block1", "conv", 2)
    x, skip1 = _xception_block(plan, x, [256, 256, 256], "entry_flow_block2", "conv", 2, return_skip=True)
    x = _xception_block(plan, x, [728, 728, 728], "entry_flow_block3", "conv", b3_stride)
    for i in range(16):
        x = _xception_block(plan, x, [728, 728, 728], "middle_flow_unit_%d" % (i + 1), "sum", 1, rate=mid_rate)
    x = _xception_block(plan, x, [728, 1024, 1024], "exit_flow_block1", "conv", 1, rate=exit_rates[0])
    x = _xception_block(plan, x, [1536, 1536, 2048], "exit_flow_block2", "none", 1, rate=exit_rates[1], depth_activation=True)
    b4 = plan.avgpool("image_pooling_pool", x, x.H)
    b4 = plan.bn("image_pooling_BN", plan.conv("image_pooling", b4, 256, 1, bn_stats=True), 1e-5, relu=1)
    b0 = plan.bn("aspp0_BN", plan.conv("aspp0", x, 256, 1, bn_stats=True), 1e-5, relu=1)
    bs = [_sepconv_bn(plan, x, 256, "aspp%d" % (i + 1), rate=r, depth_activation=True, eps=1e-5) for i, r in enumerate(aspp_rates)]
    cat = plan.concat_resize("aspp_concat", [(b4, x.H), (b0, 1)] + [(b, 1) for b in bs])
    y = plan.bn("concat_projection_BN", plan.conv("concat_projection", cat, 256, 1, bn_stats=True), 1e-5, relu=1)
    y = plan.dropout("dropout", y, 0.1, DEEPLAB_DROPOUT_SALT)
    y = plan.resize_ac("decoder_upsample", y, -(-H // 4), -(-W // 4))      # model.py:480-481: ceil(input / 4) = skip1's size
    d = plan.bn("feature_projection0_BN", plan.conv("feature_projection0", skip1, 48, 1, bn_stats=True), 1e-5, relu=1)
    y = plan.concat_resize("decoder_concat", [(y, 1), (d, 1)])
    y = _sepconv_bn(plan, y, 256, "decoder_conv0", depth_activation=True, eps=1e-5)
    y = _sepconv_bn(plan, y, 256, "decoder_conv1", depth_activation=True, eps=1e-5)
    z = plan.conv(deeplab_logits_name(classes), y, classes, 1, bias=True)
    p_lo = plan.sigmoid_act("probs_lo", z)
    probs = plan.resize_ac("logits", p_lo, H, W)
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)
        plan.prob_loss(probs, target, loss[0], loss[1])
    else:
        plan.probs_out(probs)
    return probs


def deeplab(plan, backbone, *args, **kw):
    if backbone == "xception":
        return deeplab_xception(plan, backbone, *args, **kw)
    kw.pop("OS", None)
    return deeplab_mobilenetv2(plan, backbone, *args, **kw)


NETWORKS = {"DeepLabV3": deeplab, "Unet": unet_resnet, "Linknet": linknet_resnet, "FPN": fpn_resnet, "PSPNet": pspnet_resnet}
