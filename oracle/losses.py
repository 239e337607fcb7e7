"""Oracle (test infrastructure): Keras / musket_core losses and metrics, PyTorch-CPU fp32.

Names follow the registry at ``segmentation_pipeline/segmentation.py:15-22`` and the
loss suggestions at ``segmentation_pipeline/schemas/segmentation.raml:12-21``; the
composite grammar ``"binary_crossentropy+0.1*dice_loss"`` is documented at
reference ``README.md:210-214``.  Bodies restate Keras 2.2.4 (TF backend) and
musket_core.losses (un-vendored: PARITY UNPINNED).  All take probabilities
(post-activation), like Keras losses do.
"""
import re

import numpy as np
import torch

KERAS_EPSILON = np.float32(1e-7)


def binary_crossentropy(y_true, p):
    """Keras ``binary_crossentropy`` on probabilities, TF backend: clip to
    [eps, 1-eps], recover logits, sigmoid-CE-with-logits, mean over all axes."""
    eps = torch.tensor(KERAS_EPSILON)
    one = torch.tensor(np.float32(1.0))
    pc = torch.clamp(p, eps, one - eps)
    z = torch.log(pc / (one - pc))
    ce = torch.clamp(z, min=0) - z * y_true + torch.log1p(torch.exp(-torch.abs(z)))
    return ce.mean()


def categorical_crossentropy(y_true, p):
    """Keras ``categorical_crossentropy`` on probabilities (last axis = classes)."""
    eps = torch.tensor(KERAS_EPSILON)
    p = p / p.sum(dim=-1, keepdim=True)
    p = torch.clamp(p, eps, 1.0 - eps)
    return (-(y_true * torch.log(p)).sum(dim=-1)).mean()


def dice_coef(y_true, p, smooth=1.0):
    inter = (y_true * p).sum()
    return (2.0 * inter + smooth) / (y_true.sum() + p.sum() + smooth)


def dice_loss(y_true, p):
    """musket_core.losses.dice_coef_loss: 1 - soft dice over the flattened batch."""
    return 1.0 - dice_coef(y_true, p)


def iou_coef(y_true, p, smooth=1.0):
    inter = (y_true * p).sum()
    return (inter + smooth) / (y_true.sum() + p.sum() - inter + smooth)


def iou_loss(y_true, p):
    return 1.0 - iou_coef(y_true, p)


def jaccard_loss(y_true, p, smooth=100.0):
    """``jaccard_loss`` = musket_core.losses.jaccard_distance_loss (the widely used Keras snippet): sums over the LAST
    (class) axis, (1 - (|y p| + s) / (|y| + |p| - |y p| + s)) * s with s = 100, Keras then takes the mean."""
    inter = (y_true * p).abs().sum(dim=-1)
    tot = (y_true.abs() + p.abs()).sum(dim=-1)
    return ((1.0 - (inter + smooth) / (tot - inter + smooth)) * smooth).mean()


def focal_loss(y_true, p, gamma=2.0, alpha=0.25):
    """``focal_loss``: binary focal loss, gamma 2 / alpha 0.25, probabilities clipped to [eps, 1-eps] (Keras epsilon), the
    positive and the negative term each a mean over all elements (musket_core body unpinned; constants fixed HERE)."""
    eps = float(KERAS_EPSILON)
    pt1 = torch.clamp(torch.where(y_true == 1, p, torch.ones_like(p)), eps, 1.0 - eps)
    pt0 = torch.clamp(torch.where(y_true == 0, p, torch.zeros_like(p)), eps, 1.0 - eps)
    return -(alpha * (1.0 - pt1) ** gamma * torch.log(pt1)).mean() - ((1.0 - alpha) * pt0 ** gamma * torch.log(1.0 - pt0)).mean()


def lovasz_loss(y_true, p):
    """``lovasz_loss``: binary Lovasz hinge (Berman et al. 2018) per image, mean over images, on the logits recovered the Keras
    way from the probabilities (clip to [eps, 1 - eps], log(p / (1 - p))).  errors = 1 - logit * (2 y - 1), sorted descending;
    loss = relu(errors_sorted) . grad with grad = the first differences of the Jaccard index of the sorted ground truth
    (constant w.r.t. the logits).  musket_core's body is unpinned: this is the published algorithm, fixed HERE."""
    eps = float(KERAS_EPSILON)
    pc = torch.clamp(p, eps, 1.0 - eps)
    logits = torch.log(pc / (1.0 - pc))
    total = 0.0
    for n in range(y_true.shape[0]):
        lg, gt = logits[n].reshape(-1), y_true[n].reshape(-1)
        errors = 1.0 - lg * (2.0 * gt - 1.0)
        order = torch.argsort(errors.detach(), descending=True, stable=True)
        es, gs = errors[order], gt[order].to(torch.float64)
        gts = gs.sum()
        inter = gts - gs.cumsum(0)
        union = gts + (1.0 - gs).cumsum(0)
        jac = 1.0 - inter / union
        jac = torch.cat([jac[:1], jac[1:] - jac[:-1]])
        total = total + (torch.relu(es).to(torch.float64) * jac).sum()
    return (total / y_true.shape[0]).to(torch.float32)


def dice_metric(y_true, p):
    """``dice`` metric: soft-dice formula on predictions thresholded at 0.5."""
    return dice_coef(y_true, (p > 0.5).to(p.dtype))


def iot_metric(y_true, p):
    """``iot`` metric ("at 0.5 threshold", schemas/segmentation.raml:105): iou_coef on thresholded predictions."""
    return iou_coef(y_true, (p > 0.5).to(p.dtype))


def binary_accuracy(y_true, p):
    return ((p > 0.5).to(p.dtype) == y_true).to(p.dtype).mean()


LOSSES = {
    "binary_crossentropy": binary_crossentropy,
    "categorical_crossentropy": categorical_crossentropy,
    "dice_loss": dice_loss,
    "iou_loss": iou_loss,
    "jaccard_loss": jaccard_loss,
    "focal_loss": focal_loss,
    "lovasz_loss": lovasz_loss,
}

_TERM = re.compile(r"^\s*(?:([0-9.eE+-]+)\s*\*\s*)?([A-Za-z_][A-Za-z0-9_]*)\s*$")


def parse_loss(spec):
    """``"a+w*b"`` -> [(1.0,'a'), (w,'b')]  (reference README.md:210-214)."""
    out = []
    for term in spec.split("+"):
        m = _TERM.match(term)
        if not m:
            raise ValueError("cannot parse loss term %r" % term)
        out.append((float(m.group(1)) if m.group(1) else 1.0, m.group(2)))
    return out


def composite_loss(spec, y_true, p):
    total = 0.0
    for w, name in parse_loss(spec):
        total = total + w * LOSSES[name](y_true, p)
    return total
