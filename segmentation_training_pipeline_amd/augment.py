"""Host side of the on-device augmentation stage.

The reference feeds the network from imgaug worker processes (``augmentation:`` YAML list, catalogue
``segmentation_pipeline/schemas/augmenters.raml:43-133``, semantics README.md:247-268: a
``Sequential`` of augmenters applied jointly to image and mask, followed by a resize to ``shape``).
Here the host only SAMPLES the per-image parameters (a 24-float record per device pass) and folds runs of geometric
augmenters (+ the final resize) into one 2x3 output->input matrix; the pixels are moved by HIP kernels
(``stp_augment_u8``, ``stp_filter_u8``), so no CPU worker touches image data.

LISTED ORDER (imgaug ``Sequential``): a device pass is [warp, point operations in the kernel's fixed order, filters].  The
sampler (``sample_staged``) walks the YAML list and opens a NEW pass whenever the next augmenter could not run after the
previous ones inside the current pass (a geometric augmenter after a point operation or filter, a point operation that the
kernel applies before one already used, a repeated point operation, a third filter); the trailing Resize to ``shape`` is
folded into the last pass only when that pass is pure geometry, otherwise it is a pass of its own.  The common pipelines
(geometry first, then colour, then blur, item size = network size) stay ONE pass; others cost one extra elementwise pass per
boundary at the item's resolution.  ``sample_batch`` / ``sample_batch_ex`` (bench.py, record-layout tests) keep the merged
single-pass form.

Supported augmenters (YAML name -> effect):
  geometry (composed into the matrix): Fliplr(p), Flipud(p), Rotate90, Affine{scale, translate_percent, rotate, shear},
    CropToFixedSize{width,height}, PadToFixedSize{width,height}, Pad{px}, CropAndPad{percent} (the last two keep the size,
    imgaug's default ``keep_size=True``);
  point operations (the kernel's fixed order, see include/stp_hip.h): Add, Multiply (value or range, ``per_channel``),
    AddElementwise, MultiplyElementwise, AdditiveGaussianNoise{scale, per_channel}, Dropout{p, per_channel},
    Grayscale{alpha}, Invert(p);
  containers: Sequential, Sometimes{p, then_list}, OneOf.
  neighbourhood filters (a second kernel, ``stp_filter_u8``, on the augmented batch at the network resolution, after the
    point operations; up to MAX_FILTERS per image, in the listed order): GaussianBlur{sigma}, AverageBlur{k},
    MedianBlur{k}, Sharpen{alpha, lightness}, Emboss{alpha, strength}, EdgeDetect{alpha}, DirectedEdgeDetect{alpha, direction}.
  displacement fields (``stp_field_piecewise`` / ``stp_field_elastic`` -> ``stp_augment_field_u8``: the pass's warp samples
    at p + D(p), so the field costs no extra resampling): PiecewiseAffine{scale, nb_rows, nb_cols},
    ElasticTransformation{alpha, sigma}.  One field per pass, after the pass's matrix geometry and before its point
    operations; listed-order sampling only (``sample_staged`` / ``sample_batch_staged``).
  BackgroundReplacer{path, rate, erosion, augmenters} (musket's augmenter for background-removal tasks, README.md:270-278,
    FAQ.md:24-38): ``augmenters`` run first on the item; then, with probability 1 - rate, the pixels outside the mask (eroded
    by ``erosion`` pixels) take a random image of the folder, resized to the item (``stp_background_replace_u8`` at the head of
    a new pass).  3-channel items, listed-order sampling only.
Unknown augmenter names raise ``ValueError`` (no silent skipping).
"""
import os
import math

import numpy as np

AUG_RECORD = 24
F_INVERT, F_NOISE_PC, F_DROP_PC, F_ADDE_PC, F_MULE_PC, F_ADDE, F_MULE = 1, 2, 4, 8, 16, 32, 64
IRWIN_HALL_STD = 147.8   # std of the sum of 4 uniform bytes (the kernel's Gaussian-noise generator)
FILTER_RECORD, FILTER_KMAX, MAX_FILTERS = 173, 13, 2
ELASTIC_RECORD, ELASTIC_RMAX = 69, 64


def _rng_range(rng, v, default):
    """imgaug-style stochastic parameter: scalar or (lo, hi)."""
    if v is None:
        return default
    if isinstance(v, (list, tuple)):
        return float(rng.uniform(v[0], v[1]))
    return float(v)


def _arg(args, key, default=None):
    """YAML gives either a scalar/list (positional form, README.md:266-268) or a mapping."""
    if isinstance(args, dict):
        return args.get(key, default)
    return args if args is not None else default


class SampleParams(object):
    """One image's sampled pipeline: the canvas transform so far plus the point-operation parameters."""

    def __init__(self, h, w):
        self.h, self.w = float(h), float(w)        # current canvas
        self.M = np.eye(3)                          # current-canvas pixel -> original input pixel
        self.add = np.zeros(3)
        self.mul = np.ones(3)
        self.flags = 0
        self.gray, self.noise, self.drop = 0.0, 0.0, 0.0
        self.adde, self.mule = None, None
        self.filters = []                           # [(K, mode, weights K*K float or None)]
        self.disp = None                            # ("piecewise", rows, cols, int32 [rows*cols*2]) | ("elastic", int32 [69])
        self.bg = None                              # (uint8 [bh,bw,3] background image, erosion): replaced BEFORE this pass's warp

    # ---- geometry: every op appends the map from the NEW canvas to the PREVIOUS one
    def push(self, cur_to_prev):
        self.M = self.M @ cur_to_prev

    def affine(self, scale, tx, ty, rotate, shear):
        rot, sh = math.radians(rotate), math.radians(shear)
        cx, cy = self.w / 2.0 - 0.5, self.h / 2.0 - 0.5
        a = np.array([[scale * math.cos(rot), -scale * math.sin(rot + sh), tx * self.w],
                      [scale * math.sin(rot), scale * math.cos(rot + sh), ty * self.h], [0.0, 0.0, 1.0]])
        t0 = np.array([[1.0, 0.0, -cx], [0.0, 1.0, -cy], [0.0, 0.0, 1.0]])
        t1 = np.array([[1.0, 0.0, cx], [0.0, 1.0, cy], [0.0, 0.0, 1.0]])
        self.push(np.linalg.inv(t1 @ a @ t0))

    def window(self, top, left, new_h, new_w):
        """Crop (positive offsets) / pad with the constant border (negative offsets) to a new canvas."""
        self.push(np.array([[1.0, 0.0, left], [0.0, 1.0, top], [0.0, 0.0, 1.0]]))
        self.h, self.w = float(new_h), float(new_w)

    def resize(self, new_h, new_w):
        sy, sx = self.h / new_h, self.w / new_w
        self.push(np.array([[sx, 0.0, 0.5 * sx - 0.5], [0.0, sy, 0.5 * sy - 0.5], [0.0, 0.0, 1.0]]))
        self.h, self.w = float(new_h), float(new_w)


# rank of an augmenter inside one device pass (the kernel's order: warp, add, mul, mul-elementwise, add-elementwise, noise,
# dropout, grayscale, invert; then the filter kernel)
R_GEO, R_DISP, R_ADD, R_MUL, R_MULE, R_ADDE, R_NOISE, R_DROP, R_GRAY, R_INVERT, R_FILTER = range(11)


class Pipeline(object):
    """The sampled pipeline of one image as a list of device passes.  ``strict``: honour the listed order by opening a new
    pass where one pass cannot (module docstring); otherwise everything merges into one pass (kernel order)."""

    def __init__(self, h, w, strict):
        self.stages, self.ranks, self.strict = [SampleParams(h, w)], [R_GEO], bool(strict)

    @property
    def cur(self):
        return self.stages[-1]

    def _new(self):
        prev = self.cur
        self.stages.append(SampleParams(prev.h, prev.w))
        self.ranks.append(R_GEO)
        return self.cur

    def geo(self):
        if self.strict and self.ranks[-1] > R_GEO:
            self._new()
        return self.cur

    def background(self):
        """The background replacement reads the item as the previous augmenters left it: it heads a pass of its own unless
        nothing has happened yet."""
        if not self.strict:
            raise ValueError("BackgroundReplacer needs the listed-order sampler (sample_staged)")
        c = self.cur
        if self.ranks[-1] > R_GEO or c.bg is not None or c.disp is not None or not np.array_equal(c.M, np.eye(3)):
            self._new()
        return self.cur

    def displace(self):
        """A displacement field follows the matrix geometry of its pass (the warp samples at p + D(p)); a second field, or one
        after point operations / filters, opens a new pass."""
        if not self.strict:
            raise ValueError("PiecewiseAffine / ElasticTransformation need the listed-order sampler (sample_staged)")
        if self.ranks[-1] >= R_DISP:
            self._new()
        self.ranks[-1] = R_DISP
        return self.cur

    def point(self, rank):
        if self.strict and self.ranks[-1] >= rank:
            self._new()
        self.ranks[-1] = max(self.ranks[-1], rank)
        return self.cur

    def filt(self):
        if self.strict and len(self.cur.filters) >= MAX_FILTERS:
            self._new()
        self.ranks[-1] = R_FILTER
        return self.cur


def _per_channel(rng, args):
    pc = _arg(args, "per_channel", False) if isinstance(args, dict) else False
    return bool(pc) if not isinstance(pc, float) else rng.uniform() < pc


def _children(args):
    return args if isinstance(args, list) else (args or {}).get("children", (args or {}).get("then_list", []))


def _apply(spec, rng, pl):
    """spec: list of {Name: args} (the YAML form), applied in the listed order like imgaug's Sequential; pl: Pipeline."""
    for item in spec or []:
        if isinstance(item, str):
            name, args = item, None
        else:
            (name, args), = item.items()
        if name in ("Fliplr", "Flipud", "Invert"):
            p = float(args) if args is not None and not isinstance(args, dict) else float((args or {}).get("p", 1.0))
            if rng.uniform() < p:
                if name == "Fliplr":
                    sp = pl.geo()
                    sp.push(np.array([[-1.0, 0.0, sp.w - 1.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))
                elif name == "Flipud":
                    sp = pl.geo()
                    sp.push(np.array([[1.0, 0.0, 0.0], [0.0, -1.0, sp.h - 1.0], [0.0, 0.0, 1.0]]))
                else:
                    pl.point(R_INVERT).flags ^= F_INVERT
        elif name == "Rotate90":
            # musket's quarter-turn augmenter as the reference YAMLs use it (examples/people/ds_1.yaml:6): a random number
            # of counter-clockwise 90 degree turns (np.rot90 convention); the canvas swaps its sides on odd counts
            if args is None or args is True or (isinstance(args, (int, float)) and rng.uniform() < float(args)):
                for _ in range(int(rng.randint(0, 4))):
                    sp = pl.geo()
                    sp.push(np.array([[0.0, -1.0, sp.w - 1.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]))
                    sp.h, sp.w = sp.w, sp.h
        elif name == "Affine":
            a = args or {}
            scale = _rng_range(rng, a.get("scale"), 1.0)
            tp = a.get("translate_percent")
            if isinstance(tp, dict):
                tx, ty = _rng_range(rng, tp.get("x"), 0.0), _rng_range(rng, tp.get("y"), 0.0)
            elif tp is not None:
                tx, ty = _rng_range(rng, tp, 0.0), _rng_range(rng, tp, 0.0)
            else:
                tx = ty = 0.0
            pl.geo().affine(scale, tx, ty, _rng_range(rng, a.get("rotate"), 0.0), _rng_range(rng, a.get("shear"), 0.0))
        elif name == "CropToFixedSize":
            sp = pl.geo()
            tw, th = int(_arg(args, "width")), int(_arg(args, "height"))
            nh, nw = min(sp.h, th), min(sp.w, tw)
            top, left = rng.randint(0, int(sp.h - nh) + 1), rng.randint(0, int(sp.w - nw) + 1)     # position='uniform'
            sp.window(top, left, nh, nw)
        elif name == "PadToFixedSize":
            sp = pl.geo()
            tw, th = int(_arg(args, "width")), int(_arg(args, "height"))
            nh, nw = max(sp.h, th), max(sp.w, tw)
            top, left = rng.randint(0, int(nh - sp.h) + 1), rng.randint(0, int(nw - sp.w) + 1)
            sp.window(-top, -left, nh, nw)
        elif name in ("Pad", "CropAndPad"):
            sp = pl.geo()
            h0, w0 = sp.h, sp.w
            if name == "Pad":
                px = _arg(args, "px", 0)
                side = lambda: (int(px) if not isinstance(px, (list, tuple)) else int(rng.randint(int(px[0]), int(px[1]) + 1)))
                if isinstance(px, (list, tuple)) and len(px) == 4:
                    t, r, b, l = (int(v) for v in px)
                else:
                    t, r, b, l = side(), side(), side(), side()
            else:
                pc = _arg(args, "percent", 0.0)
                side = lambda: _rng_range(rng, pc, 0.0)
                t, r, b, l = (int(round(side() * h0)), int(round(side() * w0)), int(round(side() * h0)), int(round(side() * w0)))
            nh, nw = max(1.0, h0 + t + b), max(1.0, w0 + l + r)                       # negative = crop
            sp.window(-t, -l, nh, nw)
            sp.resize(h0, w0)                                                         # keep_size=True (imgaug default)
        elif name in ("Add", "Multiply"):
            v = _arg(args, "value" if name == "Add" else "mul")
            pc = _per_channel(rng, args)
            vals = [_rng_range(rng, v, 0.0 if name == "Add" else 1.0) for _ in range(3 if pc else 1)] * (1 if pc else 3)
            if name == "Add":
                pl.point(R_ADD).add += np.round(vals)
            else:
                pl.point(R_MUL).mul *= vals
        elif name in ("AddElementwise", "MultiplyElementwise"):
            v = _arg(args, "value" if name == "AddElementwise" else "mul")
            lo, hi = (v[0], v[1]) if isinstance(v, (list, tuple)) else (v, v)
            pc = _per_channel(rng, args)
            if name == "AddElementwise":
                sp = pl.point(R_ADDE)
                sp.adde = (int(round(lo)), int(round(hi)))
                sp.flags |= F_ADDE | (F_ADDE_PC if pc else 0)
            else:
                sp = pl.point(R_MULE)
                sp.mule = (float(lo), float(hi))
                sp.flags |= F_MULE | (F_MULE_PC if pc else 0)
        elif name == "AdditiveGaussianNoise":
            sp = pl.point(R_NOISE)
            sp.noise = _rng_range(rng, _arg(args, "scale", 0.0), 0.0)
            if _per_channel(rng, args):
                sp.flags |= F_NOISE_PC
        elif name == "Dropout":
            sp = pl.point(R_DROP)
            sp.drop = _rng_range(rng, _arg(args, "p", 0.0), 0.0)
            if _per_channel(rng, args):
                sp.flags |= F_DROP_PC
        elif name == "Grayscale":
            pl.point(R_GRAY).gray = _rng_range(rng, _arg(args, "alpha", 1.0), 1.0)
        elif name in ("GaussianBlur", "AverageBlur", "MedianBlur", "Sharpen", "Emboss", "EdgeDetect", "DirectedEdgeDetect"):
            f = _filter(name, args, rng)
            if f is not None:
                if not pl.strict and len(pl.cur.filters) >= MAX_FILTERS:
                    raise ValueError("more than %d neighbourhood filters in one augmentation pass" % MAX_FILTERS)
                pl.filt().filters.append(f)
        elif name == "PiecewiseAffine":
            # imgaug 0.3.0 geometric.PiecewiseAffine: a nb_rows x nb_cols grid at linspace(0, h) x linspace(0, w); every point is
            # moved by normal(0, scale) * (h, w) and clipped into the image (schemas/augmenters.raml:126-129)
            a = args if isinstance(args, dict) else {"scale": args}
            scale = _rng_range(rng, a.get("scale"), 0.0)
            rows, cols = int(a.get("nb_rows", 4)), int(a.get("nb_cols", 4))
            if rows < 2 or cols < 2:
                raise ValueError("PiecewiseAffine needs nb_rows, nb_cols >= 2")
            if scale > 0:
                sp = pl.displace()
                ys, xs = np.meshgrid(np.linspace(0, sp.h, rows), np.linspace(0, sp.w, cols), indexing="ij")
                jit = rng.normal(0.0, scale, size=(rows, cols, 2))
                dy = np.clip(ys + jit[..., 0] * sp.h, 0, sp.h - 1) - ys
                dx = np.clip(xs + jit[..., 1] * sp.w, 0, sp.w - 1) - xs
                grid = np.stack([np.rint(dx * 64.0), np.rint(dy * 64.0)], axis=-1)
                sp.disp = ("piecewise", rows, cols, np.clip(grid, -32768, 32767).astype(np.int32).reshape(-1))
        elif name == "ElasticTransformation":
            # imgaug 0.3.0 geometric.ElasticTransformation(alpha, sigma): see stp_field_elastic (schemas/augmenters.raml:130-133;
            # `scale` there is read as sigma)
            a = args if isinstance(args, dict) else {"alpha": args}
            alpha = _rng_range(rng, a.get("alpha"), 0.0)
            sigma = _rng_range(rng, a.get("sigma", a.get("scale")), 0.0)
            if alpha > 0:
                radius = int(4.0 * sigma + 0.5) if sigma > 0 else 0          # scipy.ndimage.gaussian_filter, truncate = 4
                if radius > ELASTIC_RMAX:
                    raise ValueError("ElasticTransformation: sigma %.3g needs a blur radius above %d" % (sigma, ELASTIC_RMAX))
                k = np.arange(-radius, radius + 1, dtype=np.float64)
                g = np.exp(-0.5 * (k / sigma) ** 2) if radius else np.ones(1)
                q = np.rint(g / g.sum() * 32768.0).astype(np.int64)
                q[radius] += 32768 - int(q.sum())                            # the quantised kernel keeps the DC gain exactly
                rec = np.zeros(ELASTIC_RECORD, np.int32)
                rec[0], rec[1], rec[2] = int(rng.randint(0, 1 << 24)), min(int(round(alpha * 64.0)), 1 << 15), radius
                rec[4:4 + radius + 1] = q[radius:]
                pl.displace().disp = ("elastic", rec)
        elif name == "BackgroundReplacer":
            a = args if isinstance(args, dict) else {"path": args}
            sub = a.get("augmenters")
            if sub:
                _apply([{k: v} for k, v in sub.items()] if isinstance(sub, dict) else list(sub), rng, pl)
            files = background_files(a.get("path"))
            if rng.uniform() >= float(a.get("rate", 0.5)):                    # rate = fraction of original backgrounds preserved
                bg = load_background(files[int(rng.randint(0, len(files)))])
                er = a.get("erosion", 0)
                er = int(rng.randint(int(er[0]), int(er[1]) + 1)) if isinstance(er, (list, tuple)) else int(er or 0)
                if not 0 <= er <= 32:
                    raise ValueError("BackgroundReplacer: erosion %d is outside 0..32" % er)
                pl.background().bg = (bg, er)
        elif name == "Sequential":
            _apply(_children(args), rng, pl)
        elif name == "Sometimes":
            a = args or {}
            if rng.uniform() < float(a.get("p", 0.5)):
                _apply(a.get("then_list", []), rng, pl)
        elif name == "OneOf":
            ch = _children(args)
            if ch:
                _apply([ch[rng.randint(0, len(ch))]], rng, pl)
        else:
            raise ValueError("augmenter %r is not available in the HIP augmentation stage" % name)


def _odd_k(rng, k):
    k = int(round(_rng_range(rng, k, 3)))
    return k


def _filter(name, args, rng):
    """(K, mode, weights) of one neighbourhood augmenter; None when it samples to the identity."""
    if name == "GaussianBlur":
        sigma = _rng_range(rng, _arg(args, "sigma", 0.0), 0.0)
        if sigma < 1e-3:
            return None
        K = min(FILTER_KMAX, 2 * int(math.ceil(3.0 * sigma)) + 1)          # +-3 sigma, truncated at radius 6 (sigma > 2)
        x = np.arange(K) - K // 2
        g = np.exp(-0.5 * (x / sigma) ** 2)
        g /= g.sum()
        return K, 0, np.outer(g, g)
    if name in ("AverageBlur", "MedianBlur"):
        k = _odd_k(rng, _arg(args, "k", 3))
        if name == "MedianBlur" and k % 2 == 0:
            k += 1                                                            # imgaug: even sizes are incremented
        if k <= 1:
            return None
        k = min(k, FILTER_KMAX)
        if name == "MedianBlur":
            return k, 1, None
        K = k if k % 2 else k + 1                                             # an even box sits in an odd window (anchor like cv2.blur)
        wts = np.zeros((K, K))
        wts[:k, :k] = 1.0 / (k * k)
        return K, 0, wts
    alpha = _rng_range(rng, _arg(args, "alpha", 1.0) if isinstance(args, dict) else args, 1.0)
    ident = np.zeros((3, 3)); ident[1, 1] = 1.0
    if name == "Sharpen":
        l = _rng_range(rng, _arg(args, "lightness", 1.0) if isinstance(args, dict) else None, 1.0)
        eff = np.array([[-1, -1, -1], [-1, 8 + l, -1], [-1, -1, -1]], np.float64)
    elif name == "Emboss":
        st = _rng_range(rng, _arg(args, "strength", 1.0) if isinstance(args, dict) else None, 1.0)
        eff = np.array([[-1 - st, 0 - st, 0], [0 - st, 1, 0 + st], [0, 0 + st, 1 + st]], np.float64)
    elif name == "DirectedEdgeDetect":
        # imgaug 0.3.0 convolutional.DirectedEdgeDetect: every neighbour cell is weighted by (1 - angle to the sampled direction
        # / 180 deg)^4, the weights normalised to sum 1 and negated, the centre is 1 (schemas/augmenters.raml:118-121)
        direction = _rng_range(rng, _arg(args, "direction", (0.0, 1.0)) if isinstance(args, dict) else (0.0, 1.0), 0.0)
        rad = np.deg2rad(int(direction * 360) % 360)
        dvec = np.array([np.cos(rad - 0.5 * np.pi), np.sin(rad - 0.5 * np.pi)])
        eff = np.zeros((3, 3), np.float64)
        for cx in (-1, 0, 1):
            for cy in (-1, 0, 1):
                if (cx, cy) != (0, 0):
                    cell = np.array([cx, cy], np.float64)
                    ang = np.degrees(np.arccos(np.clip(np.dot(cell / np.linalg.norm(cell), dvec / np.linalg.norm(dvec)), -1.0, 1.0)))
                    eff[cy + 1, cx + 1] = (1.0 - ang / 180.0) ** 4
        eff = -eff / eff.sum()
        eff[1, 1] = 1.0
    else:
        eff = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.float64)
    if alpha <= 0:
        return None
    return 3, 0, (1.0 - alpha) * ident + alpha * eff


def filter_records(filters_per_image):
    """[[(K, mode, weights), ...] per image] -> None (no filter anywhere) or int32 [passes, n, 173] for stp_filter_u8."""
    passes = max((len(f) for f in filters_per_image), default=0)
    if passes == 0:
        return None
    out = np.zeros((passes, len(filters_per_image), FILTER_RECORD), np.int32)
    for i, fl in enumerate(filters_per_image):
        for ps, (K, mode, wts) in enumerate(fl):
            out[ps, i, 0], out[ps, i, 1] = K, mode
            if wts is not None:
                q = np.rint(np.asarray(wts, np.float64) * 16384.0).astype(np.int64).reshape(-1)
                q[(K * K) // 2 if q[(K * K) // 2] else int(np.argmax(np.abs(q)))] += int(round(float(np.sum(wts)) * 16384.0)) - int(q.sum())
                out[ps, i, 4:4 + K * K] = q                     # the quantised weights keep the filter's exact DC gain
    return out


def record(sp, out_hw, seed):
    """float32[24] record of ``stp_augment_u8`` (layout: include/stp_hip.h) for one sampled pipeline."""
    if out_hw is not None:
        oh, ow = out_hw
        if (sp.h, sp.w) != (float(oh), float(ow)):
            sp.resize(oh, ow)                  # the trailing Resize to the network shape
    r = np.zeros(AUG_RECORD, np.float32)
    r[0:6] = sp.M[:2].reshape(-1)
    r[6:9], r[9:12] = sp.add, sp.mul
    r[12] = sp.flags
    r[13] = int(round(min(max(sp.gray, 0.0), 1.0) * 256))
    r[14] = int(round(sp.noise * 65536.0 / IRWIN_HALL_STD))
    r[15] = int(round(min(max(sp.drop, 0.0), 1.0) * (1 << 24)))
    if sp.adde is not None:
        r[16], r[17] = sp.adde
    if sp.mule is not None:
        r[18], r[19] = sp.mule
    r[20] = seed & 0xffffff
    return r


def _seed(sp, rng):
    return int(rng.randint(0, 1 << 24)) if sp.flags or sp.noise or sp.drop else 0


def sample_batch_ex(spec, rng, n, h, w, out_hw):
    """(float32 [n,24] records for ``stp_augment_u8``, None or int32 [passes,n,173] records for ``stp_filter_u8``): the MERGED
    single-pass form (every augmenter folded into one pass in the kernel's order; bench.py and the record-layout tests)."""
    out = np.zeros((n, AUG_RECORD), np.float32)
    filt = []
    for i in range(n):
        pl = Pipeline(h, w, strict=False)
        _apply(spec, rng, pl)
        sp = pl.cur
        out[i] = record(sp, out_hw, _seed(sp, rng))
        filt.append(sp.filters)
    return out, filter_records(filt)


def sample_staged(spec, rng, h, w, out_hw):
    """One image, listed order: [(float32[24] record, filters [(K, mode, weights)], (out_h, out_w)) per device pass]."""
    pl = Pipeline(h, w, strict=True)
    _apply(spec, rng, pl)
    oh, ow = out_hw
    last = pl.cur
    if (last.h, last.w) != (float(oh), float(ow)) and pl.ranks[-1] > R_GEO:
        pl._new()                                  # the trailing Resize follows point operations / filters: its own pass
    passes = []
    for k, sp in enumerate(pl.stages):
        final = k == len(pl.stages) - 1
        rec = record(sp, (oh, ow) if final else None, _seed(sp, rng))
        hw = (int(round(sp.h)), int(round(sp.w)))
        t = (rec, sp.filters, hw)
        if sp.disp is not None or sp.bg is not None:
            t += (sp.disp,)
        if sp.bg is not None:
            t += (sp.bg,)
        passes.append(t)
    return passes


def _disp_key(p):
    """Structure of a pass beyond its canvas: images agree on it or run one by one (a background image is per image)."""
    if len(p) > 4:
        return object()
    return None if len(p) < 4 or p[3] is None else (p[3][0],) + (tuple(p[3][1:3]) if p[3][0] == "piecewise" else ())


_BG_FILES, _BG_IMAGES = {}, {}


def background_files(path):
    """Image files of a BackgroundReplacer folder (sorted; cached)."""
    if not path or not os.path.isdir(str(path)):
        raise ValueError("BackgroundReplacer: path %r is not a directory" % (path,))
    path = os.path.abspath(str(path))
    if path not in _BG_FILES:
        fs = sorted(f for f in os.listdir(path) if f.lower().endswith((".jpg", ".jpeg", ".png", ".bmp")))
        if not fs:
            raise ValueError("BackgroundReplacer: no images in %r" % path)
        _BG_FILES[path] = [os.path.join(path, f) for f in fs]
    return _BG_FILES[path]


def load_background(f, cache=256):
    """uint8 [h, w, 3] (decoded once; at most ``cache`` images stay resident)."""
    if f not in _BG_IMAGES:
        from PIL import Image
        if len(_BG_IMAGES) >= cache:
            _BG_IMAGES.pop(next(iter(_BG_IMAGES)))
        _BG_IMAGES[f] = np.array(Image.open(f).convert("RGB"), dtype=np.uint8, order="C")
    return _BG_IMAGES[f]


def resolve_paths(spec, base_dir):
    """BackgroundReplacer paths such as ``./bg`` (README.md:275) that do not exist relative to the working directory are
    taken relative to the experiment's directory.  Returns a new list."""
    out = []
    for item in spec or []:
        if isinstance(item, dict):
            (name, args), = item.items()
            if name == "BackgroundReplacer" and isinstance(args, dict):
                args = dict(args)
                pth = args.get("path")
                if pth and not os.path.isdir(str(pth)) and base_dir and os.path.isdir(os.path.join(base_dir, str(pth))):
                    args["path"] = os.path.join(base_dir, str(pth))
                sub = args.get("augmenters")
                if sub:
                    args["augmenters"] = resolve_paths([{k: v} for k, v in sub.items()] if isinstance(sub, dict) else sub, base_dir)
                item = {name: args}
            elif name in ("Sequential", "OneOf") and args:
                item = {name: resolve_paths(_children(args), base_dir)}
            elif name == "Sometimes" and isinstance(args, dict):
                item = {name: dict(args, then_list=resolve_paths(args.get("then_list", []), base_dir))}
        out.append(item)
    return out


def batch_disp(disps):
    """[displacement of one image] of equal kind -> the batch form: ("piecewise", rows, cols, int32 [n, rows*cols*2]) or
    ("elastic", int32 [n, 69])."""
    d0 = disps[0]
    return (d0[0], d0[1], d0[2], np.stack([d[3] for d in disps])) if d0[0] == "piecewise" else (d0[0], np.stack([d[1] for d in disps]))


def batch_of_one(p):
    """One image's pass -> the batch form ``DeviceFeeder._run_passes`` executes (n = 1)."""
    out = (p[0][None], filter_records([p[1]]), p[2])
    if len(p) > 3:
        out += (None if p[3] is None else batch_disp([p[3]]),)
    return out + tuple(p[4:5])


def sample_batch_staged(spec, rng, n, h, w, out_hw):
    """A batch of equally sized images in listed order -> (batch passes, per-image passes):
    ``batch passes`` = [(records float32 [n,24], filter records int32 [f,n,173] or None, (out_h, out_w)) per device pass] when
    every image has the same pass structure (same count, same canvas sizes: the batch kernels apply), else None and
    ``per-image passes`` = [sample_staged(...) per image] (Sometimes / OneOf changed the structure of some images)."""
    per = [sample_staged(spec, rng, h, w, out_hw) for _ in range(n)]
    k = len(per[0])
    if all(len(p) == k and all(p[j][2] == per[0][j][2] and _disp_key(p[j]) == _disp_key(per[0][j]) for j in range(k)) for p in per):
        out = []
        for j in range(k):
            bp = (np.stack([p[j][0] for p in per]).astype(np.float32), filter_records([p[j][1] for p in per]), per[0][j][2])
            out.append(bp if len(per[0][j]) < 4 or per[0][j][3] is None else bp + (batch_disp([p[j][3] for p in per]),))
        return out, None
    return None, per


def sample_batch(spec, rng, n, h, w, out_hw):
    """float32 [n,24] parameter records for ``stp_augment_u8`` (pipelines without neighbourhood filters)."""
    out, filt = sample_batch_ex(spec, rng, n, h, w, out_hw)
    if filt is not None:
        raise ValueError("this pipeline contains neighbourhood filters: use sample_batch_ex")
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
