"""Oracle (test infrastructure): one training step = forward + loss + backward + optimizer.

This is the CPU restatement of the per-step hot loop of ``cfg.fit()`` (reference
``README.md:125``; loop body in un-vendored musket_core/Keras ``fit_generator``, call
sites ``segmentation_pipeline/segmentation.py:35,54,160-168``).  PARITY UNPINNED.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import losses, nets, optim


def synthetic_batch(n, h, w, seed=1234, discs=3):
    """images uint8 uniform 0..255; masks = union of random discs."""
    rng = np.random.RandomState(seed)
    img = rng.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    mask = np.zeros((n, h, w, 1), np.uint8)
    for i in range(n):
        for _ in range(discs):
            cy, cx = rng.uniform(0, h), rng.uniform(0, w)
            r = rng.uniform(0.08, 0.22) * min(h, w)
            mask[i, :, :, 0] |= ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.uint8)
    return img, mask


class OracleTrainer:
    """Holds numpy params + Keras-semantics optimizer; steps with torch autograd."""

    def __init__(self, params, backbone="resnet34", loss="binary_crossentropy+1.0*dice_loss",
                 optimizer="adam", lr=1e-3, freeze_encoder=False, clipnorm=None, clipvalue=None,
                 decoder_filters=(256, 128, 64, 32, 16), opt_kwargs=None, architecture="Unet", activation="sigmoid",
                 net_kwargs=None, storage=None, grad_scale=None, accum64=False):
        self.P = OrderedDict((k, v.copy()) for k, v in params.items())
        self.backbone = backbone
        self.loss_spec = loss
        self.opt = optim.make(optimizer, lr=lr, **(opt_kwargs or {}))
        self.trainable = nets.trainable_names(self.P, freeze_encoder)
        self.clipnorm, self.clipvalue = clipnorm, clipvalue
        self.decoder_filters = tuple(decoder_filters)
        self.architecture = architecture
        self.net_kwargs = dict(net_kwargs or {})      # PSPNet: downsample_factor
        self.steps_done = 0              # mirrors the device step counter that seeds DeepLab's dropout mask
        self.activation = activation     # "sigmoid": y [N,H,W,1] in {0,1};  "softmax": y [N,H,W,1] class index -> one-hot
        # "bf16" / "fp16": the storage-quantised oracle (nets._Ctx) - U-Net / Linknet / FPN / PSPNet over the ResNet and VGG encoders.
        # grad_scale: the loss scale the device's stored gradients carry (None: 1 - the fp16 tests pass the build's 2^14)
        self.storage = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[storage]
        self.grad_scale = float(grad_scale or 1.0)
        self.accum64 = bool(accum64)         # float64 accumulation inside the convolutions: the second evaluation of the same rounding points
        if self.storage is not None and architecture == "DeepLabV3":
            raise ValueError("the storage-quantised oracle covers the U-Net / Linknet / FPN / PSPNet graphs")

    def _forward(self, P, x, training, taps):
        if self.architecture == "DeepLabV3":      # returns PROBABILITIES (the activation is inside the model, deeplab.py)
            from . import deeplab
            if self.backbone == "xception":
                return deeplab.deeplab_xception_forward(P, x, training=training, taps=taps, step=self.steps_done + 1, **self.net_kwargs)
            return deeplab.deeplab_forward(P, x, training=training, taps=taps, step=self.steps_done + 1)
        q = dict(storage=self.storage, grad_scale=self.grad_scale, accum64=self.accum64)
        if self.architecture == "Linknet":
            return nets.linknet_resnet_forward(P, x, self.backbone, training=training, taps=taps, **q)
        if self.architecture == "PSPNet":
            return nets.pspnet_resnet_forward(P, x, self.backbone, training=training, taps=taps, step=self.steps_done + 1, **q, **self.net_kwargs)
        if self.architecture == "FPN":
            return nets.fpn_resnet_forward(P, x, self.backbone, training=training, taps=taps, step=self.steps_done + 1, **q, **self.net_kwargs)
        return nets.unet_resnet_forward(P, x, self.backbone, training=training, taps=taps, decoder_filters=self.decoder_filters, **q)

    def forward(self, x_nhwc, training=False, taps=None):
        with torch.no_grad():
            P = nets.to_torch(self.P)
            logits, _ = self._forward(P, torch.from_numpy(x_nhwc.astype(np.float32)), training, taps)
        return logits.numpy()

    def step(self, x_nhwc, y_nhwc, taps=None, apply=True):
        """x: [N,H,W,3] float32 (raw 0..255), y: [N,H,W,1] float32 {0,1}.
        Returns dict(logits, loss, dice, dice_loss, bce, grads)."""
        P = nets.to_torch(self.P, self.trainable)
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32))
        y = torch.from_numpy(np.ascontiguousarray(y_nhwc, dtype=np.float32))
        logits, bn_updates = self._forward(P, x, True, taps)
        if self.architecture == "DeepLabV3":
            p = logits
            if self.activation == "softmax":
                y = torch.nn.functional.one_hot(y[..., 0].long(), logits.shape[-1]).to(torch.float32)
        elif self.activation == "softmax":
            p = torch.softmax(logits, dim=-1)
            y = torch.nn.functional.one_hot(y[..., 0].long(), logits.shape[-1]).to(torch.float32)
        else:
            p = torch.sigmoid(logits)
        loss = losses.composite_loss(self.loss_spec, y, p)
        loss.backward()
        grads = OrderedDict((k, P[k].grad.numpy().copy()) for k in self.trainable)
        out = {
            "logits": logits.detach().numpy().copy(),
            "loss": float(loss.detach()),
            "bce": float((losses.categorical_crossentropy if self.activation == "softmax" else losses.binary_crossentropy)(y, p.detach())),
            "dice_loss": float(losses.dice_loss(y, p.detach())),
            "dice": float(losses.dice_metric(y, p.detach())),
            "binary_accuracy": float(losses.binary_accuracy(y, p.detach())),
            "grads": grads,
        }
        self.steps_done += 1
        if apply:
            g = optim.clip_grads(grads, self.clipnorm, self.clipvalue)
            self.opt.step(self.P, g)
            for k, v in bn_updates.items():
                self.P[k] = v.numpy().astype(np.float32)
        return out
