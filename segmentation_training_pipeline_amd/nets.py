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

RESNET_UNITS = {"resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}
STAGE_FILTERS = (64, 128, 256, 512)
BN_EPS_ENCODER = 2e-5
BN_EPS_DECODER = 1e-3

ENCODER_PREFIXES = ("bn_data", "conv0", "bn0", "stage", "bn1")


def known_backbones():
    return sorted(RESNET_UNITS)


def unet_resnet(plan, backbone, H, W, in_ch=3, classes=1, decoder_filters=(256, 128, 64, 32, 16),
                loss=(1.0, 1.0), with_loss=True):
    """Declares inputs 'image' (uint8 NHWC) and 'mask' (uint8 NHW1); returns the logits tensor."""
    if H % 32 or W % 32:
        raise ValueError("U-Net input height/width must be divisible by 32")
    units = RESNET_UNITS[backbone]
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
                shortcut = plan.conv(pre + "sc", a, f, 1, stride=stride, pad=0)
            else:
                shortcut = x
            y = plan.conv(pre + "conv1", a, f, 3, stride=stride, pad=1, bn_stats=True)
            y = plan.bn(pre + "bn2", y, BN_EPS_ENCODER, relu=True)
            x = plan.conv(pre + "conv2", y, f, 3, stride=1, pad=1, residual=shortcut, bn_stats=True)
    x = plan.bn("bn1", x, BN_EPS_ENCODER, relu=True)
    skips = (taps[4], taps[3], taps[2], relu0, None)
    for i, f in enumerate(decoder_filters):
        pre = "decoder_stage%d_" % i
        x = plan.conv(pre + "conv1", x, f, 3, pad=1, src1=skips[i], upsample=True, bn_stats=True)
        x = plan.bn(pre + "bn1", x, BN_EPS_DECODER, relu=True)
        x = plan.conv(pre + "conv2", x, f, 3, pad=1, bn_stats=True)
        x = plan.bn(pre + "bn2", x, BN_EPS_DECODER, relu=True)
    logits = plan.conv("final_conv", x, classes, 3, pad=1, bias=True)
    if with_loss:
        target = plan.input_u8("mask", H, W, 1)
        plan.sigmoid_loss(logits, target, loss[0], loss[1])
    return logits
