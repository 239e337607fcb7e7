"""Oracle (test infrastructure): DeepLabV3+ over MobileNetV2, PyTorch-CPU fp32.

Unlike the segmentation_models graphs, this architecture is IN the reference tree:
``segmentation_pipeline/impl/deeplab/model.py`` (registered as ``DeepLabV3`` at ``segmentation.py:31-33``; the example
experiment ``examples/people/ds_1.yaml:1-2`` trains it).  Every step below cites the lines it restates.  The Keras layers
themselves (BatchNormalization / Conv2D / DepthwiseConv2D arithmetic, TF 'same' padding, ``resize_bilinear``) are still
un-vendored, so parity stays UNPINNED in the sense of oracle/__init__.py - but the graph is pinned by in-tree source.

MobileNetV2 branch only (``backbone_name='mobilenetv2'``, alpha = 1, OS = 8: model.py:381-431), ASPP with the image-pooling and
1x1 branches (model.py:438-456), projection + Dropout(0.1) (model.py:457-461), the class convolution WITH the activation and
the bilinear (align_corners) upsampling of its output (model.py:485-486).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import nets

# (filters, stride, expansion, block_id, skip_connection, rate) - model.py:395-431
BLOCKS = [(16, 1, 1, 0, False, 1), (24, 2, 6, 1, False, 1), (24, 1, 6, 2, True, 1), (32, 2, 6, 3, False, 1), (32, 1, 6, 4, True, 1),
          (32, 1, 6, 5, True, 1), (64, 1, 6, 6, False, 1), (64, 1, 6, 7, True, 2), (64, 1, 6, 8, True, 2), (64, 1, 6, 9, True, 2),
          (96, 1, 6, 10, False, 2), (96, 1, 6, 11, True, 2), (96, 1, 6, 12, True, 2), (160, 1, 6, 13, False, 2), (160, 1, 6, 14, True, 4),
          (160, 1, 6, 15, True, 4), (320, 1, 6, 16, False, 4)]
BN_MOBILENET = dict(eps=1e-3, momentum=0.999)       # model.py:388-389, 247-248, 258-259, 267-268
BN_ASPP = dict(eps=1e-5, momentum=0.99)              # model.py:443, 449, 459 (Keras default momentum)
DROPOUT_RATE = 0.1                                   # model.py:461
DROPOUT_SALT = 0x0D0D


def logits_name(classes):
    """model.py:494-497: 'logits_semantic' for the 21 PASCAL-VOC classes, 'custom_logits_semantic' otherwise."""
    return "logits_semantic" if classes == 21 else "custom_logits_semantic"


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def _bn(P, name, c):
    P[name + "/gamma"] = np.ones(c, np.float32)
    P[name + "/beta"] = np.zeros(c, np.float32)
    P[name + "/moving_mean"] = np.zeros(c, np.float32)
    P[name + "/moving_variance"] = np.ones(c, np.float32)


def init_deeplab_mobilenetv2(in_ch=3, classes=1, seed=42):
    """Keras layouts: Conv2D kernels HWIO, DepthwiseConv2D kernels (kh, kw, C, 1); layer names as in model.py."""
    rng = np.random.RandomState(seed)
    P = OrderedDict()
    P["Conv/kernel"] = _glorot(rng, (3, 3, in_ch, 32), 9 * in_ch, 9 * 32)
    _bn(P, "Conv_BN", 32)
    cin = 32
    for filters, stride, exp, bid, skip, rate in BLOCKS:
        pre = "expanded_conv_%d_" % bid if bid else "expanded_conv_"            # model.py:240-251
        c = cin
        if bid:
            P[pre + "expand/kernel"] = _glorot(rng, (1, 1, cin, exp * cin), cin, exp * cin)
            _bn(P, pre + "expand_BN", exp * cin)
            c = exp * cin
        P[pre + "depthwise/depthwise_kernel"] = _glorot(rng, (3, 3, c, 1), 9 * c, 9)
        _bn(P, pre + "depthwise_BN", c)
        P[pre + "project/kernel"] = _glorot(rng, (1, 1, c, filters), c, filters)
        _bn(P, pre + "project_BN", filters)
        cin = filters
    for name in ("image_pooling", "aspp0"):
        P[name + "/kernel"] = _glorot(rng, (1, 1, cin, 256), cin, 256)
        _bn(P, name + "_BN", 256)
    P["concat_projection/kernel"] = _glorot(rng, (1, 1, 512, 256), 512, 256)
    _bn(P, "concat_projection_BN", 256)
    P[logits_name(classes) + "/kernel"] = _glorot(rng, (1, 1, 256, classes), 256, classes)
    P[logits_name(classes) + "/bias"] = np.zeros(classes, np.float32)
    return P


def _same_pad(size, k, stride, rate):
    """TF 'same': output ceil(size/stride); total padding split with the remainder at the END (bottom/right heavy)."""
    keff = (k - 1) * rate + 1
    out = -(-size // stride)
    total = max((out - 1) * stride + keff - size, 0)
    return total // 2, total - total // 2


def _conv_same(P, x, name, stride=1):
    w = P[name + "/kernel"].permute(3, 2, 0, 1)
    k = w.shape[2]
    pt, pb = _same_pad(x.shape[2], k, stride, 1)
    pl, pr = _same_pad(x.shape[3], k, stride, 1)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, P.get(name + "/bias"), stride=stride)


def _dwconv_same(P, x, name, stride, rate):
    w = P[name + "/depthwise_kernel"].permute(2, 3, 0, 1)                      # (kh,kw,C,1) -> (C,1,kh,kw)
    pt, pb = _same_pad(x.shape[2], 3, stride, rate)
    pl, pr = _same_pad(x.shape[3], 3, stride, rate)
    return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, None, stride=stride, dilation=rate, groups=x.shape[1])


def _bn_act(ctx, x, name, cfg, act):
    """Keras BatchNormalization (training: biased batch variance; moving variance with the unbiased estimate) + activation."""
    P = ctx.P
    c = x.shape[1]
    if ctx.training:
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean.view(1, c, 1, 1)) ** 2).mean(dim=(0, 2, 3))
        n = x.numel() // c
        with torch.no_grad():
            m = cfg["momentum"]
            ctx.bn_updates[name + "/moving_mean"] = (P[name + "/moving_mean"] * m + mean * (1 - m)).detach().clone()
            ctx.bn_updates[name + "/moving_variance"] = (P[name + "/moving_variance"] * m + var * (n / max(n - 1, 1)) * (1 - m)).detach().clone()
    else:
        mean, var = P[name + "/moving_mean"], P[name + "/moving_variance"]
    scale = torch.rsqrt(var + cfg["eps"]) * P[name + "/gamma"]
    y = (x - mean.view(1, c, 1, 1)) * scale.view(1, c, 1, 1) + P[name + "/beta"].view(1, c, 1, 1)
    if act == "relu6":
        return torch.clamp(y, 0.0, 6.0)                                        # K.relu(x, max_value=6), model.py:216-217
    return F.relu(y) if act == "relu" else y


def init_deeplab_xception(in_ch=3, classes=1, seed=42):
    """Modified Xception + ASPP + decoder (model.py:338-379, 436-491); Keras layouts, layer names as in model.py."""
    rng = np.random.RandomState(seed)
    P = OrderedDict()

    def conv(name, k, cin, cout):
        P[name + "/kernel"] = _glorot(rng, (k, k, cin, cout), k * k * cin, k * k * cout)

    def sep(prefix, cin, cout):
        P[prefix + "_depthwise/depthwise_kernel"] = _glorot(rng, (3, 3, cin, 1), 9 * cin, 9)
        _bn(P, prefix + "_depthwise_BN", cin)
        conv(prefix + "_pointwise", 1, cin, cout)
        _bn(P, prefix + "_pointwise_BN", cout)

    def block(prefix, cin, depths, skip):
        c = cin
        for i, d in enumerate(depths):
            sep(prefix + "_separable_conv%d" % (i + 1), c, d)
            c = d
        if skip == "conv":
            conv(prefix + "_shortcut", 1, cin, depths[-1])
            _bn(P, prefix + "_shortcut_BN", depths[-1])
        return c

    conv("entry_flow_conv1_1", 3, in_ch, 32); _bn(P, "entry_flow_conv1_1_BN", 32)
    conv("entry_flow_conv1_2", 3, 32, 64); _bn(P, "entry_flow_conv1_2_BN", 64)
    c = block("entry_flow_block1", 64, [128, 128, 128], "conv")
    c = block("entry_flow_block2", c, [256, 256, 256], "conv")
    c = block("entry_flow_block3", c, [728, 728, 728], "conv")
    for i in range(16):
        c = block("middle_flow_unit_%d" % (i + 1), c, [728, 728, 728], "sum")
    c = block("exit_flow_block1", c, [728, 1024, 1024], "conv")
    c = block("exit_flow_block2", c, [1536, 1536, 2048], "none")
    for name in ("image_pooling", "aspp0"):
        conv(name, 1, c, 256); _bn(P, name + "_BN", 256)
    for i in (1, 2, 3):
        sep("aspp%d" % i, c, 256)
    conv("concat_projection", 1, 5 * 256, 256); _bn(P, "concat_projection_BN", 256)
    conv("feature_projection0", 1, 256, 48); _bn(P, "feature_projection0_BN", 48)
    sep("decoder_conv0", 256 + 48, 256)
    sep("decoder_conv1", 256, 256)
    conv(logits_name(classes), 1, 256, classes)
    P[logits_name(classes) + "/bias"] = np.zeros(classes, np.float32)
    return P


BN_XCEPTION = dict(eps=1e-3, momentum=0.99)          # model.py:137, 143 (epsilon=1e-3 default of SepConv_BN; Keras default momentum)


def _dwconv_explicit(P, x, name, stride, rate):
    """model.py:126-137: stride 1 -> 'same'; stride > 1 -> ZeroPadding2D((pad_beg, pad_end)) + 'valid'."""
    if stride == 1:
        return _dwconv_same(P, x, name, 1, rate)
    w = P[name + "/depthwise_kernel"].permute(2, 3, 0, 1)
    keff = 3 + 2 * (rate - 1)
    beg = (keff - 1) // 2
    end = keff - 1 - beg
    return F.conv2d(F.pad(x, (beg, end, beg, end)), w, None, stride=stride, dilation=rate, groups=x.shape[1])


def _sepconv_bn(ctx, x, filters, prefix, stride=1, rate=1, depth_activation=False, cfg=BN_XCEPTION):
    """model.py:110-147."""
    P = ctx.P
    if not depth_activation:
        x = F.relu(x)
    x = _dwconv_explicit(P, x, prefix + "_depthwise", stride, rate)
    x = _bn_act(ctx, x, prefix + "_depthwise_BN", cfg, "relu" if depth_activation else None)
    x = _conv_same(P, x, prefix + "_pointwise")
    return _bn_act(ctx, x, prefix + "_pointwise_BN", cfg, "relu" if depth_activation else None)


def _xception_block(ctx, inputs, depths, prefix, skip_type, stride, rate=1, depth_activation=False):
    """model.py:182-218; returns (outputs, skip = output of the second SepConv)."""
    P = ctx.P
    r, skip = inputs, None
    for i in range(3):
        r = _sepconv_bn(ctx, r, depths[i], prefix + "_separable_conv%d" % (i + 1), stride=stride if i == 2 else 1, rate=rate,
                        depth_activation=depth_activation)
        if i == 1:
            skip = r
    if skip_type == "conv":
        w = P[prefix + "_shortcut/kernel"].permute(3, 2, 0, 1)
        sc = F.conv2d(inputs, w, None, stride=stride)                             # _conv2d_same, kernel 1: no padding either way
        return r + _bn_act(ctx, sc, prefix + "_shortcut_BN", BN_XCEPTION, None), skip
    if skip_type == "sum":
        return r + inputs, skip
    return r, skip


def deeplab_xception_forward(P, x_nhwc, training=True, taps=None, step=1, OS=16):
    """model.py:338-379 (feature extractor), :436-469 (ASPP), :471-491 (decoder); returns (PROBABILITIES nhwc, bn_updates)."""
    ctx = nets._Ctx(P, training, taps)
    x = x_nhwc.permute(0, 3, 1, 2)
    H, W = x.shape[2], x.shape[3]
    b3_stride, mid_rate, exit_rates, aspp_rates = (1, 2, (2, 4), (12, 24, 36)) if OS == 8 else (2, 1, (1, 2), (6, 12, 18))
    x = _bn_act(ctx, _conv_same(P, x, "entry_flow_conv1_1", stride=2), "entry_flow_conv1_1_BN", BN_XCEPTION, "relu")
    x = _bn_act(ctx, _conv_same(P, x, "entry_flow_conv1_2"), "entry_flow_conv1_2_BN", BN_XCEPTION, "relu")
    x, _ = _xception_block(ctx, x, [128, 128, 128], "entry_flow_block1", "conv", 2)
    x, skip1 = _xception_block(ctx, x, [256, 256, 256], "entry_flow_block2", "conv", 2)
    ctx.tap("entry_flow_block2", x)
    x, _ = _xception_block(ctx, x, [728, 728, 728], "entry_flow_block3", "conv", b3_stride)
    for i in range(16):
        x, _ = _xception_block(ctx, x, [728, 728, 728], "middle_flow_unit_%d" % (i + 1), "sum", 1, rate=mid_rate)
    ctx.tap("middle_flow", x)
    x, _ = _xception_block(ctx, x, [728, 1024, 1024], "exit_flow_block1", "conv", 1, rate=exit_rates[0])
    x, _ = _xception_block(ctx, x, [1536, 1536, 2048], "exit_flow_block2", "none", 1, rate=exit_rates[1], depth_activation=True)
    ctx.tap("exit_flow", x)
    hf, wf = x.shape[2], x.shape[3]
    b4 = F.avg_pool2d(x, kernel_size=(hf, wf))
    b4 = _bn_act(ctx, _conv_same(P, b4, "image_pooling"), "image_pooling_BN", BN_ASPP, "relu")
    b4 = F.interpolate(b4, size=(hf, wf), mode="bilinear", align_corners=True)
    b0 = _bn_act(ctx, _conv_same(P, x, "aspp0"), "aspp0_BN", BN_ASPP, "relu")
    bs = [_sepconv_bn(ctx, x, 256, "aspp%d" % (i + 1), rate=r, depth_activation=True, cfg=BN_ASPP) for i, r in enumerate(aspp_rates)]
    x = torch.cat([b4, b0] + bs, dim=1)                                           # model.py:467
    x = _bn_act(ctx, _conv_same(P, x, "concat_projection"), "concat_projection_BN", BN_ASPP, "relu")
    ctx.tap("concat_projection", x)
    if training:
        n, c, hh, ww = x.shape
        keep = dropout_mask(step, DROPOUT_SALT, n * hh * ww * c, DROPOUT_RATE).reshape(n, hh, ww, c).transpose(0, 3, 1, 2)
        x = x * torch.from_numpy(keep.astype(np.float32)) / (1.0 - DROPOUT_RATE)
    x = F.interpolate(x, size=(-(-H // 4), -(-W // 4)), mode="bilinear", align_corners=True)         # model.py:480-481 (ceil)
    d = _bn_act(ctx, _conv_same(P, skip1, "feature_projection0"), "feature_projection0_BN", BN_ASPP, "relu")
    x = torch.cat([x, d], dim=1)
    x = _sepconv_bn(ctx, x, 256, "decoder_conv0", depth_activation=True, cfg=BN_ASPP)
    x = _sepconv_bn(ctx, x, 256, "decoder_conv1", depth_activation=True, cfg=BN_ASPP)
    ctx.tap("decoder", x)
    z = _conv_same(P, x, "logits_semantic" if "logits_semantic/kernel" in P else "custom_logits_semantic")
    p = torch.sigmoid(z) if z.shape[1] == 1 else torch.softmax(z, dim=1)          # model.py:485: the activation lives in this layer
    p = F.interpolate(p, size=(H, W), mode="bilinear", align_corners=True)
    return p.permute(0, 2, 3, 1).contiguous(), ctx.bn_updates


def dropout_mask(step, salt, count, rate):
    """The kernel's counter-based mask (stp_dropout): keep where hash(step * 0x85EBCA77 + salt, i) >> 8 >= rate * 2^24."""
    with np.errstate(over="ignore"):
        seed = np.uint32((step * 0x85EBCA77 + salt) & 0xffffffff)
        h = seed ^ (np.arange(count, dtype=np.uint32) * np.uint32(0x9E3779B1))
        h ^= h >> np.uint32(16); h = h * np.uint32(0x7feb352d)
        h ^= h >> np.uint32(15); h = h * np.uint32(0x846ca68b)
        h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)) >= np.uint32(int(round(rate * 16777216.0)))


def deeplab_forward(P, x_nhwc, training=True, taps=None, step=1):
    """Returns (PROBABILITIES nhwc at the input size, bn_updates).  ``step``: value of the device step counter (it is ticked
    before the forward, so the first training step draws the mask of step 1)."""
    ctx = nets._Ctx(P, training, taps)
    x = x_nhwc.permute(0, 3, 1, 2)                                             # raw pixels: the graph has no preprocessing
    H, W = x.shape[2], x.shape[3]
    x = _bn_act(ctx, _conv_same(P, x, "Conv", stride=2), "Conv_BN", BN_MOBILENET, "relu6")       # model.py:383-390
    ctx.tap("Conv", x)
    for filters, stride, exp, bid, skip, rate in BLOCKS:                        # model.py:236-279
        pre = "expanded_conv_%d_" % bid if bid else "expanded_conv_"
        inp = x
        if bid:
            x = _bn_act(ctx, _conv_same(P, x, pre + "expand"), pre + "expand_BN", BN_MOBILENET, "relu6")
        x = _bn_act(ctx, _dwconv_same(P, x, pre + "depthwise", stride, rate), pre + "depthwise_BN", BN_MOBILENET, "relu6")
        x = _bn_act(ctx, _conv_same(P, x, pre + "project"), pre + "project_BN", BN_MOBILENET, None)
        if skip:
            x = inp + x
        ctx.tap("block%d" % bid, x)
    h8, w8 = x.shape[2], x.shape[3]
    b4 = F.avg_pool2d(x, kernel_size=(h8, w8))                                  # model.py:439 (pool = ceil(input / OS))
    b4 = _bn_act(ctx, _conv_same(P, b4, "image_pooling"), "image_pooling_BN", BN_ASPP, "relu")
    b4 = F.interpolate(b4, size=(h8, w8), mode="bilinear", align_corners=True)  # model.py:445
    b0 = _bn_act(ctx, _conv_same(P, x, "aspp0"), "aspp0_BN", BN_ASPP, "relu")    # model.py:448-450
    x = torch.cat([b4, b0], dim=1)                                              # model.py:456
    x = _bn_act(ctx, _conv_same(P, x, "concat_projection"), "concat_projection_BN", BN_ASPP, "relu")
    ctx.tap("concat_projection", x)
    if training:                                                                # Dropout(0.1), model.py:461
        n, c, hh, ww = x.shape
        keep = dropout_mask(step, DROPOUT_SALT, n * hh * ww * c, DROPOUT_RATE).reshape(n, hh, ww, c).transpose(0, 3, 1, 2)
        x = x * torch.from_numpy(keep.astype(np.float32)) / (1.0 - DROPOUT_RATE)
    z = _conv_same(P, x, "logits_semantic" if "logits_semantic/kernel" in P else "custom_logits_semantic")                              # model.py:485 - the activation lives in this layer
    p = torch.sigmoid(z) if z.shape[1] == 1 else torch.softmax(z, dim=1)       # sigmoid: one class; softmax: 2+ classes
    p = F.interpolate(p, size=(H, W), mode="bilinear", align_corners=True)      # model.py:486
    return p.permute(0, 2, 3, 1).contiguous(), ctx.bn_updates
