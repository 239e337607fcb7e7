"""Host side of the on-device augmentation stage.

The reference feeds the network from imgaug worker processes (``augmentation:`` YAML list, catalogue
``segmentation_pipeline/schemas/augmenters.raml:43-133``, semantics README.md:247-268: a
``Sequential`` of augmenters applied jointly to image and mask, followed by a resize to ``shape``).
Here the host only SAMPLES the per-image parameters (a few floats) and folds all geometric
augmenters + the final resize into one 2x3 output->input matrix per image; the pixels are moved
by one HIP kernel (``stp_augment_u8``), so no CPU worker touches image data.

Supported augmenters (YAML name -> effect):
  Fliplr(p), Flipud(p), Affine{scale, translate_percent, rotate, shear}, Add(value), Multiply(mul),
  Sequential / Sometimes(p, then_list) containers.
Anything else raises ``ValueError`` naming the augmenter (no silent skipping).
"""
import math

import numpy as np


def _rng_range(rng, v, default):
    """imgaug-style stochastic parameter: scalar, (lo, hi) tuple/list or {'x':..,'y':..} handled by caller."""
    if v is None:
        return default
    if isinstance(v, (list, tuple)):
        return float(rng.uniform(v[0], v[1]))
    return float(v)


class SampleParams(object):
    __slots__ = ("fliplr", "flipud", "scale", "tx", "ty", "rotate", "shear", "add", "mul")

    def __init__(self):
        self.fliplr = self.flipud = False
        self.scale, self.tx, self.ty, self.rotate, self.shear = 1.0, 0.0, 0.0, 0.0, 0.0
        self.add, self.mul = 0, 1.0


def _apply(spec, rng, sp):
    """spec: list of {Name: args} (the YAML form).  Later geometric augmenters compose onto earlier ones
    in the order given; one Affine is supported per pipeline (the README's example shape)."""
    for item in spec or []:
        if isinstance(item, str):
            name, args = item, None
        else:
            (name, args), = item.items()
        if name in ("Fliplr", "Flipud"):
            p = float(args) if args is not None and not isinstance(args, dict) else float((args or {}).get("p", 1.0))
            if rng.uniform() < p:
                if name == "Fliplr":
                    sp.fliplr = not sp.fliplr
                else:
                    sp.flipud = not sp.flipud
        elif name == "Affine":
            a = args or {}
            sp.scale *= _rng_range(rng, a.get("scale"), 1.0)
            tp = a.get("translate_percent")
            if isinstance(tp, dict):
                sp.tx += _rng_range(rng, tp.get("x"), 0.0)
                sp.ty += _rng_range(rng, tp.get("y"), 0.0)
            elif tp is not None:
                sp.tx += _rng_range(rng, tp, 0.0)
                sp.ty += _rng_range(rng, tp, 0.0)
            sp.rotate += _rng_range(rng, a.get("rotate"), 0.0)
            sp.shear += _rng_range(rng, a.get("shear"), 0.0)
        elif name == "Add":
            v = args.get("value") if isinstance(args, dict) else args
            sp.add += int(round(_rng_range(rng, v, 0.0)))
        elif name == "Multiply":
            v = args.get("mul") if isinstance(args, dict) else args
            sp.mul *= _rng_range(rng, v, 1.0)
        elif name == "Sequential":
            _apply(args if isinstance(args, list) else (args or {}).get("children", []), rng, sp)
        elif name == "Sometimes":
            a = args or {}
            if rng.uniform() < float(a.get("p", 0.5)):
                _apply(a.get("then_list", []), rng, sp)
        else:
            raise ValueError("augmenter %r is not available in the HIP augmentation stage" % name)


def matrix(sp, h, w, out_hw):
    """2x3 OUTPUT->INPUT pixel map: inverse of (flip o affine-about-centre), then the final resize."""
    rot, sh = math.radians(sp.rotate), math.radians(sp.shear)
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    a = np.array([[sp.scale * math.cos(rot), -sp.scale * math.sin(rot + sh), sp.tx * w],
                  [sp.scale * math.sin(rot), sp.scale * math.cos(rot + sh), sp.ty * h],
                  [0.0, 0.0, 1.0]])
    t0 = np.array([[1.0, 0.0, -cx], [0.0, 1.0, -cy], [0.0, 0.0, 1.0]])
    t1 = np.array([[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1.0]])
    fwd = t1 @ a @ t0
    if sp.fliplr:
        fwd = np.array([[-1.0, 0.0, w - 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]) @ fwd
    if sp.flipud:
        fwd = np.array([[1.0, 0.0, 0.0], [0.0, -1.0, h - 1.0], [0.0, 0.0, 1.0]]) @ fwd
    inv = np.linalg.inv(fwd)
    oh, ow = out_hw
    if (oh, ow) != (h, w):
        inv = inv @ np.array([[w / ow, 0.0, 0.5 * w / ow - 0.5], [0.0, h / oh, 0.5 * h / oh - 0.5], [0.0, 0.0, 1.0]])
    return inv[:2]


def sample_batch(spec, rng, n, h, w, out_hw):
    """float32 [n,10] parameter records for ``stp_augment_u8`` (m00 m01 m02 m10 m11 m12 add mul 0 0)."""
    out = np.zeros((n, 10), np.float32)
    for i in range(n):
        sp = SampleParams()
        _apply(spec, rng, sp)
        out[i, :6] = matrix(sp, h, w, out_hw).reshape(-1)
        out[i, 6], out[i, 7] = sp.add, sp.mul
    return out


def identity_batch(n, h, w, out_hw):
    """Validation / inference transform: Resize to the network shape only (transformAugmentor,
    reference segmentation.py:39,224)."""
    return sample_batch([], np.random.RandomState(0), n, h, w, out_hw)


# The README example pipeline (README.md:251-262) plus colour jitter: the benchmark's augmentation
BENCH_SPEC = [{"Fliplr": 0.5}, {"Flipud": 0.5},
              {"Affine": {"scale": [0.8, 1.5], "translate_percent": {"x": [-0.2, 0.2], "y": [-0.2, 0.2]},
                          "rotate": [-16, 16], "shear": [-16, 16]}},
              {"Add": [-20, 20]}, {"Multiply": [0.8, 1.2]}]
