#!/usr/bin/env python
"""Benchmark of the hot path: one training step on one MI355X (N ranks under torch.distributed.run).

DEFAULT (--config 1, what the driver runs) = BASELINE.json configs[1]: U-Net/ResNet-34, 512x512x3 -> 1 class, batch 16 per GPU, bf16
MFMA; configs[2] = the same with --gpus 8.  --config 3 = configs[3]: FPN/ResNet-50 1024x1024 3-class batch 4, fp16 MFMA;
--config 4 = configs[4]: PSPNet/ResNet-101 768x768 20-class batch 8, bf16, heavy augmentation.  Same JSON shape
for every config (roofline, roofline_hbm, step_traffic_gb, kernel_time_us, cpu_baseline, box_calibration).

One "step" = on-device augmentation of the resident uint8 batch + weight compute copies + forward + loss (sigmoid BCE + Dice, or
softmax CCE + Dice) + backward + (RCCL gradient all-reduce when N > 1) + Adam.  Raw images and masks are resident in HBM before the
timed region.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --config 4 --cpu-baseline short
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \
        bench.py --gpus 8 --steps 20 --warmup 5
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 512          # config 1 (the module-level names are config 1's: dice_delta_vs_oracle, tests)
BATCH = 16
LOSS = "binary_crossentropy+1.0*dice_loss"
LOSS_SOFTMAX = "categorical_crossentropy+1.0*dice_loss"
# algorithmic work per trained image (BASELINE.md 3 / SURVEY 8d): conv MACs only, 2 FLOP/MAC, 3x forward
FLOP_PER_IMAGE = 187.94e9
PEAK_BF16_TFLOPS = 2500.0      # dense bf16 / fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0

# BASELINE.json configs by index: network, input, batch per GPU, classes, the precision BASELINE.json names,
# augmentation pipeline ("S1": the README example + colour jitter = augment.BENCH_SPEC; "S4": heavy = Affine + blur + noise + CropAndPad)
CONFIGS = {
    1: dict(architecture="Unet", backbone="resnet34", size=512, batch=16, classes=1, dtype="bf16", augment="S1", pmc=""),
    3: dict(architecture="FPN", backbone="resnet50", size=1024, batch=4, classes=3, dtype="fp16", augment="S1", pmc="_config3"),
    4: dict(architecture="PSPNet", backbone="resnet101", size=768, batch=8, classes=20, dtype="bf16", augment="S4", pmc="_config4"),
}
_PMC = ""            # suffix of the counter files of the running config: profiles/*_pmc_traffic<_PMC>.json, *_pmc_sq<_PMC>.json


def augment_spec(kind):
    from segmentation_training_pipeline_amd import augment
    if kind == "S1":
        return augment.BENCH_SPEC
    # S4: "heavy augment (Affine + blur + noise + CropAndPad)" on top of the flips and the colour jitter of S1
    return [{"Fliplr": 0.5}, {"Flipud": 0.5}, {"CropAndPad": {"percent": [-0.1, 0.1]}},
            {"Affine": {"scale": [0.8, 1.5], "translate_percent": {"x": [-0.2, 0.2], "y": [-0.2, 0.2]}, "rotate": [-16, 16], "shear": [-16, 16]}},
            {"Add": [-20, 20]}, {"Multiply": [0.8, 1.2]}, {"AdditiveGaussianNoise": {"scale": [0, 12.75]}}, {"GaussianBlur": {"sigma": [0.5, 1.5]}}]


def synthetic_data(rank, batch, size, classes):
    """uniform uint8 images; 1 class: the union of 3 random discs per mask; softmax heads: a class-index map of
    ``classes`` random regions (nearest random seed point).  Seed 1234 + rank."""
    rng = np.random.RandomState(1234 + rank)
    img = rng.randint(0, 256, size=(batch, size, size, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    msk = np.zeros((batch, size, size), np.uint8)
    for i in range(batch):
        if classes == 1:
            for _ in range(3):
                cy, cx, r = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(0.08, 0.22) * size
                msk[i] |= ((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r).astype(np.uint8)
        else:
            best = np.full((size, size), np.inf)
            for c in range(classes):
                cy, cx = rng.uniform(0, size), rng.uniform(0, size)
                d = (yy - cy) ** 2 + (xx - cx) ** 2
                msk[i][d < best] = c
                best = np.minimum(best, d)
    return rng, img, msk


CONV_TILES = {1: "128, 128, 2, 2", 2: "64, 256, 1, 4", 3: "32, 256, 1, 4", 4: "16, 256, 1, 4", 5: "64, 64, 2, 2", 6: "128, 64, 4, 1",
              7: "64, 128, 2, 2"}


def wgrad_tile(cout):
    return "16, 256, 1, 4" if cout <= 16 else "32, 256, 1, 4" if cout <= 32 else "64, 128, 1, 4" if cout <= 64 else "128, 128, 2, 2"


def kernel_key(name, meta, dtype):
    t = "float" if dtype == "fp32" else "unsigned short"
    if dtype == "fp16":
        dtype = "bf16"          # same kernel set, same template arguments: the storage format is a build parameter
    if name == "stp_conv2d":
        tile = meta["tile"]
        if tile == 512:   # small-channel streaming kernel: the lean form (conv_sc_lean.hip, 16-bit storage: <input channels, 16-channel output
            # tiles, epilogue, ...>) or the generic one (conv_sc.hip: <storage, input channels, 16-channel output tiles>)
            if dtype == "bf16" and os.environ.get("STP_SC_LEAN", "1") != "0" and meta.get("sc"):
                return "conv_sc_lean_kernel<%d, %d" % tuple(meta["sc"])
            return "conv_sc_stream_kernel<%s, %d, %d>" % ((t,) + tuple(meta["sc"])) if meta.get("sc") else "conv_sc_stream_kernel<%s>" % t
        if tile == 640:   # wide-output small-channel data gradient (conv_sc.hip: stp_conv2d_scw)
            return "conv_scw_stream_kernel<%s>" % t
        if tile == 704:   # narrow-output small-channel forward (conv_sc.hip: stp_conv2d_scn)
            return "conv_scn_stream_kernel<%s>" % t
        if tile == 736:   # 64 -> 64 channels with the weights in registers (conv_sc.hip: stp_conv2d_s64)
            return "conv_s64_stream_kernel<%s>" % t
        if tile == 800:   # pointwise (1x1 / stride 1) pixel-streaming kernel (conv_pw.hip): <input channels, output channels, waves, tile pixels, ...>
            return "conv_pw_kernel<%d, %d" % tuple(meta["pw"])
        if tile == 768:   # the stem: persistent form (conv_sc_lean.hip) unless switched off
            return "conv_stem_lean_kernel" if os.environ.get("STP_STEM_LEAN", "1") != "0" else "conv_stem_kernel"
        if tile >= 1024:  # halo-resident 3x3 kernel (conv_halo.hip): <rows of 16 pixels, channels, waves over channels x pixel rows> (the profiler's name carries the epilogue variant as a 5th argument)
            return "%s<%s>" % ("conv_halo_p64_kernel" if tile == 1029 else          # (p64: one workgroup per CU walks the tiles of a 64 -> 64 layer, weights in registers)
                               "conv_halo_s2d_kernel" if meta.get("s2d") else "conv_halo_fold1_kernel" if meta.get("fold1") else
                               "conv_halo2_kernel" if meta.get("src2") else "conv_halo_kernel",      # (halo2: two sources / upsampled first source; s2d: space-to-depth data gradient of a stride-2 layer)
                               ("16, 128, 2, 4", "8, 128, 2, 4", "16, 64, 1, 8", "8, 64, 2, 4", "32, 64, 1, 8", "16, 64, 2, 4")[tile - 1024])
        if tile >= 256:  # buffer-DMA kernel, per-lane tap (small channel counts), 2 stages
            return "conv_igemm_ut_kernel<%s, %s, 2, false>" % (t, CONV_TILES[tile - 256])
        if tile >= 64:   # uniform-tap buffer-DMA kernel: tile = 32*STAGES + base tile
            return "conv_igemm_ut_kernel<%s, %s, %d, true>" % (t, CONV_TILES[tile % 32], tile // 32)
        c4 = "true" if (dtype == "bf16" and meta["layer"] == "conv0") else "false"
        return "conv_igemm_kernel<%s, %s, %s>" % (t, CONV_TILES[tile], c4)
    if name == "stp_wgrad_group_partial":      # grouped weight gradient (conv_wgrad.hip): one launch per group of layers
        if meta.get("taps9"):                  # all-taps tiles (round 4): <output channels per workgroup, ring stages>
            return "conv_wgrad_taps9_group_kernel<%d, 3>" % meta["bm"]
        return "conv_wgrad_row_group%s_kernel<%s, 3>" % ("_pbn" if meta.get("pbn") else "", {128: "128, 2, 2", 64: "64, 1, 4", 32: "32, 1, 4"}[meta["bm"]])
    if name == "stp_conv2d_wgrad":
        if meta.get("sc"):
            if dtype == "bf16" and meta["layer"] == "conv0":
                return "conv_stem_wgrad_lean_kernel"
            return "conv_sc_wgrad_lean_kernel" if (dtype == "bf16" and os.environ.get("STP_SC_LEAN", "1") != "0") else "conv_sc_wgrad_kernel<%s>" % t
        if meta.get("kernel_id") in (2, 3):   # row-of-taps kernel (conv_wgrad.hip): <output channels per workgroup, waves over channels x columns>
            return "conv_wgrad_row_kernel<%s, %s>" % ("128, 2, 2" if meta["kernel_id"] == 2 else "64, 1, 4", os.environ.get("STP_WGRAD_ROW_STAGES", "3"))
        if dtype == "bf16" and meta["layer"] == "conv0":
            return "conv_wgrad_kernel<%s, %s, true>" % (t, wgrad_tile(meta["cout"]))
        if dtype == "fp32" and meta["cout"] <= 32:
            return "conv_wgrad_kernel<%s, %s, false>" % (t, wgrad_tile(meta["cout"]))
        # (..., STAGES = 2, ROWU = true: every feature map of this workload is a power of two)
        return "conv_wgrad_dma_kernel<%s, %s, 2, true>" % (t, wgrad_tile(meta["cout"]))
    return name


_ESIZE = {0: 4, 1: 2, 2: 1, 3: 2}      # stp dtype codes (include/stp_hip.h): F32, BF16, U8, F16


def hbm_bytes(name, a):
    """ALGORITHMIC HBM bytes of one launch of a memory-bound C-ABI entry point (every operand tensor read or written exactly once;
    per-channel tables and partial sums ignored), from its argument list (include/stp_hip.h).  None = not a streaming pass."""
    if name == "stp_bn_apply":                       # (x, xdtype, y, ydtype, rows, C, Cy, ...)
        return a[4] * (a[5] * _ESIZE[a[1]] + a[6] * _ESIZE[a[3]])
    if name == "stp_bn_finalize_apply":              # (partial, tiles, x, y, dtype, rows, C, ...)
        return 2 * a[5] * a[6] * _ESIZE[a[4]]
    if name == "stp_bn_backward_fused":              # (x, g, dx, dtype, rows, C, mean, rstd, gamma, partial, tiles, dgamma, dbeta, accumulate_dx, ...)
        return (3 + int(bool(a[13]))) * a[4] * a[5] * _ESIZE[a[3]]
    if name == "stp_bn_backward_fused_add":          # (x, g, dx, dadd, dtype, rows, C, ..., accumulate_dx at 14)
        return (3 + int(bool(a[14]))) * a[5] * a[6] * _ESIZE[a[4]]
    if name == "stp_bn_backward":                    # (x, dy, dx, dtype, rows, C, ..., accumulate_dx at 13): sums pass + apply pass
        return (5 + int(bool(a[13]))) * a[4] * a[5] * _ESIZE[a[3]]
    if name == "stp_adam":                           # (param, grad, m, v, count, ...): 4 reads + 3 writes of fp32
        return 28 * a[4]
    if name == "stp_maxpool3x3s2":                   # (x, y, idx, N, H, W, C, dtype)
        n, h, w, c, es = a[3], a[4], a[5], a[6], _ESIZE[a[7]]
        return n * h * w * c * es + n * ((h + 1) // 2) * ((w + 1) // 2) * c * (es + 1)
    if name == "stp_bn_apply_maxpool3x3s2":          # (x, yb, y, idx, N, H, W, C, dtype, ...): one read + one write of the map, pooled map + indexes
        n, h, w, c, es = a[4], a[5], a[6], a[7], _ESIZE[a[8]]
        return 2 * n * h * w * c * es + n * (h // 2) * (w // 2) * c * (es + 1)
    if name == "stp_maxpool3x3s2_bwd":               # (idx, dy, dx, N, H, W, C, dtype, accumulate)
        n, h, w, c, es = a[3], a[4], a[5], a[6], _ESIZE[a[7]]
        return n * ((h + 1) // 2) * ((w + 1) // 2) * c * (es + 1) + n * h * w * c * es * (1 + int(bool(a[8])))
    return None


# Memory-bound kernel FAMILIES: the C-ABI entry points of one family launch the same device kernels (profiler names, prefix match),
# so the PMC traffic of the family = the bytes of those kernels per step / the launches of its pass kernels (`main`) per step.
HBM_FAMILIES = {
    "BatchNormalization backward": {"entries": ("stp_bn_backward_fused", "stp_bn_backward_fused_add", "stp_bn_backward"),
                                    "kernels": ("bn_bwd_apply_kernel", "bn_bwd_finalize_apply_kernel", "bn_bwd_finalize_tiles_kernel",
                                                "bn_bwd_partial_kernel", "bn_bwd_finalize_kernel"),
                                    "main": ("bn_bwd_apply_kernel", "bn_bwd_finalize_apply_kernel")},
    "BatchNormalization forward": {"entries": ("stp_bn_apply", "stp_bn_finalize_apply"),
                                   "kernels": ("bn_apply_v8_kernel", "bn_apply_u8_kernel", "bn_apply_kernel", "bn_finalize_apply_kernel"),
                                   "main": ("bn_apply_v8_kernel", "bn_apply_u8_kernel", "bn_apply_kernel", "bn_finalize_apply_kernel")},
    "max-pooling forward + backward": {"entries": ("stp_maxpool3x3s2", "stp_bn_apply_maxpool3x3s2", "stp_maxpool3x3s2_bwd"),
                                       "kernels": ("maxpool_fwd_kernel", "bn_apply_maxpool_kernel", "maxpool_bwd_kernel"),
                                       "main": ("maxpool_fwd_kernel", "bn_apply_maxpool_kernel", "maxpool_bwd_kernel")},
}


def per_kernel_profile(model, reps=3):
    """Eager instrumented passes: every launch of the step bracketed by HIP events on the stream the
    kernels run on (torch's current stream).  Returns {kernel: [launches, seconds, flops, algorithmic HBM bytes]} per step."""
    p = model.plan
    st = torch.cuda.current_stream()
    launches = [(l, "prep") for l in p.prep] + [(l, "fwd") for l in p.fwd] + [(l, "bwd") for l in p.bwd]
    acc = {}
    saved = [t.clone() for t in model._mutable_state()]
    times, recs = [], None          # per pass: the duration of every launch; the launches' records (same in every pass)
    for rep in range(reps + 1):
        evs = []
        for (fn, args, name, meta), _ in launches:
            if fn is None:
                continue  # fork/join markers of the side-stream weight-gradient chain: this pass is single-stream
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            rc = fn(*args, st.cuda_stream)
            e1.record(st)
            if rc != 0:
                raise RuntimeError("%s failed (%d)" % (name, rc))
            evs.append((name, meta, e0, e1, hbm_bytes(name, args)))
        torch.cuda.synchronize()
        if rep == 0:
            continue  # first pass warms caches / clocks
        times.append([e0.elapsed_time(e1) * 1e-3 for _, _, e0, e1, _ in evs])
        recs = [(name, meta, byt) for name, meta, _, _, byt in evs]
    # the MEDIAN over the passes per launch: one pass in which a launch waited tens of milliseconds for something outside the step (seen
    # once on the pool: 11 launches of one key 620 us each instead of 24) must not move a key to the top of the table
    for i, (name, meta, byt) in enumerate(recs):
        key = kernel_key(name, meta, model.dtype) if (meta and "flops" in meta) else name
        a = acc.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(np.median([t[i] for t in times]))
        a[2] += meta["flops"] if (meta and "flops" in meta) else 0.0
        a[3] += byt or 0
    for t, s in zip(model._mutable_state(), saved):
        t.copy_(s)
    return {k: [float(v[0]), v[1], v[2], float(v[3])] for k, v in acc.items()}


def flop_per_image(model):
    fwd = sum(m["flops"] for _, _, _, m in model.plan.fwd if m and "flops" in m)
    return 3.0 * fwd / model.batch


def csrc_fingerprint():
    """sha256 (first 16 hex digits) over the kernel sources of the build: the counter files under profiles/ record the fingerprint of
    the build they were collected on (scratch/pmc_aggregate*.py), so a bench line can say whether its ``traffic`` / ``mfma_util`` -
    which need separate profiler passes and are therefore read from those files - belong to the kernels that just ran."""
    import hashlib
    d = os.path.join(ROOT, "segmentation_training_pipeline_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _pmc_stale(path):
    try:
        with open(path) as f:
            return json.load(f).get("csrc_sha16") != csrc_fingerprint()
    except (OSError, ValueError):
        return True


def _pmc_files(pdir, suffix):
    """Counter files under profiles/: those collected on this build of csrc/ (matching fingerprint) first, then by name, newest first."""
    if not os.path.isdir(pdir):
        return []
    names = sorted((f for f in os.listdir(pdir) if f.endswith(suffix)), reverse=True)
    return sorted(names, key=lambda f: _pmc_stale(os.path.join(pdir, f)))      # (stable: False < True)


def _pmc_entries(kernels, kernel):
    """Counter records of ``kernel``: the exact name, or - the profiler's names of some kernels carry one more template argument than
    bench.py's keys (the epilogue variant of conv_halo_kernel) - every instance that extends it."""
    if kernel in kernels:
        return [kernels[kernel]]
    if kernel.endswith(">"):
        pre = kernel[:-1] + ","
        return [v for k, v in kernels.items() if k.startswith(pre)]
    return []


def pmc_traffic(kernel):
    """L2-miss bytes per launch of ``kernel`` (read + write) from the rocprofv3 PMC passes committed under profiles/
    (FETCH_SIZE and WRITE_SIZE need separate passes and a profiler run, so they are not collected live); None if
    that kernel was not in the measured build.  Several instances of one key: dispatch-weighted mean."""
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    names = _pmc_files(pdir, "_pmc_traffic%s.json" % _PMC)
    for name in names:      # the measurement of THIS build first, then the newest that knows this kernel
        path = os.path.join(pdir, name)
        try:
            with open(path) as f:
                ks = _pmc_entries(json.load(f)["kernels"], kernel)
        except (OSError, ValueError, KeyError):
            continue
        n = sum(k.get("dispatches", 1) for k in ks)
        if n:
            return int(sum((k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]) * k.get("dispatches", 1) for k in ks) / n), "profiles/" + name
    return None, None


def pmc_step_traffic():
    """Sum of the L2-miss bytes (FETCH_SIZE + WRITE_SIZE passes) over EVERY kernel of one step, from the counter file of this build
    (profiles/*_pmc_traffic.json; the pass runs `steps` eager steps, 3 unless the file says otherwise): (bytes per step, file) or
    (None, None)."""
    pdir = os.path.join(ROOT, "profiles")
    for name in _pmc_files(pdir, "_pmc_traffic%s.json" % _PMC):
        try:
            with open(os.path.join(pdir, name)) as f:
                d = json.load(f)
            steps = float(d.get("steps", 3))
            return sum(((k.get("fetch_bytes_per_launch") or 0) + (k.get("write_bytes_per_launch") or 0)) * k.get("dispatches", 0)
                       for k in d["kernels"].values()) / steps, "profiles/" + name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def pmc_family_traffic(family):
    """L2-miss bytes per launch of a memory-bound kernel family (HBM_FAMILIES): the bytes of its device kernels per step / the
    launches of its pass kernels per step, from the counter file of this build: (bytes per launch, file, kernels) or Nones."""
    pdir = os.path.join(ROOT, "profiles")
    fam = HBM_FAMILIES[family]
    for name in _pmc_files(pdir, "_pmc_traffic%s.json" % _PMC):
        try:
            with open(os.path.join(pdir, name)) as f:
                d = json.load(f)
            ks = {k: v for k, v in d["kernels"].items() if k.startswith(fam["kernels"])}
            n = sum(v.get("dispatches", 0) for k, v in ks.items() if k.startswith(fam["main"]))
            if not n:
                continue
            tot = sum(((v.get("fetch_bytes_per_launch") or 0) + (v.get("write_bytes_per_launch") or 0)) * v.get("dispatches", 0) for v in ks.values())
            return int(tot / n), "profiles/" + name, sorted(ks)
        except (OSError, ValueError, KeyError):
            continue
    return None, None, None


def pmc_mfma_util(kernel):
    """MFMA pipe utilisation of ``kernel`` from the committed SQ counter pass (profiles/*_pmc_sq.json, scratch/pmc_aggregate_sq.py):
    SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x kernel cycles); None when that kernel was not measured.  Several instances of one key:
    weighted by their total duration."""
    pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    names = _pmc_files(pdir, "_pmc_sq%s.json" % _PMC)
    for name in names:
        try:
            with open(os.path.join(pdir, name)) as f:
                ks = [k for k in _pmc_entries(json.load(f)["kernels"], kernel) if k.get("mfma_util") is not None]
        except (OSError, ValueError, KeyError):
            continue
        wsum = sum(k.get("dispatches", 1) * k.get("avg_duration_us", 1.0) for k in ks)
        if wsum:
            return round(sum(k["mfma_util"] * k.get("dispatches", 1) * k.get("avg_duration_us", 1.0) for k in ks) / wsum, 4), "profiles/" + name
    return None, None


def cpu_baseline(full_protocol=False, config=1):
    """The in-repo CPU oracle (a PORT: the reference's Keras-CPU fit() is not installable here) running the
    same step - CPU augmentation (oracle/augment.py: warp + point operations + neighbourhood filters of the config's pipeline) +
    forward + loss + backward + Adam - on a bounded sample of the workload.  Config 1 DEFAULT = BASELINE.md 4's protocol as written
    (batch 16, 3 warm-up + 10 timed steps, median: ~5 minutes of host time on the GPU box's cores - the driver's budget for the bench
    is 30 minutes); ``--cpu-baseline short`` runs 1 warm-up + 3 timed steps (~1.5 minutes) and says so in ``protocol``.  Configs 3 / 4:
    the config's own batch, 1 warm-up + 2 timed steps (full: 1 + 4)."""
    from oracle import augment as oaug
    from oracle import nets as onets
    from oracle import step as ostep
    from segmentation_training_pipeline_amd import augment
    cfg = CONFIGS[config]
    n, size, classes = cfg["batch"], cfg["size"], cfg["classes"]
    if config == 1:
        warm, reps = (3, 10) if full_protocol else (1, 3)
    else:
        warm, reps = (1, 4) if full_protocol else (1, 2)
    init = {"Unet": onets.init_unet_resnet, "FPN": onets.init_fpn_resnet, "PSPNet": onets.init_pspnet_resnet}[cfg["architecture"]]
    P = init(cfg["backbone"], classes=classes, seed=42)
    tr = ostep.OracleTrainer(P, backbone=cfg["backbone"], loss=LOSS if classes == 1 else LOSS_SOFTMAX, optimizer="adam", lr=1e-3,
                             architecture=cfg["architecture"], activation="sigmoid" if classes == 1 else "softmax")
    _, x8, y8 = synthetic_data(0, n, size, classes)
    rng = np.random.RandomState(1234)
    spec = augment_spec(cfg["augment"])

    def one_step():
        prm, filt = augment.sample_batch_ex(spec, rng, n, size, size, (size, size))
        xa, ya = oaug.warp_u8(x8, y8, prm, (size, size))
        for ps in range(0 if filt is None else filt.shape[0]):
            xa = oaug.filter_u8(xa, filt[ps])
        tr.step(xa.astype(np.float32), ya.reshape(n, size, size, 1).astype(np.float32))

    for _ in range(warm):
        one_step()
    times = []
    for _ in range(reps):
        t0 = time.time()
        one_step()
        times.append(time.time() - t0)
    med = float(np.median(times))
    full_name = "BASELINE.md 4" if config == 1 else "configs[%d]: 1 warm-up + 4 timed" % config
    return {"value": round(n / med, 3), "unit": "images/sec", "cores": int(torch.get_num_threads()), "kind": "port",
            "protocol": full_name if full_protocol else "short (batch %d, %d warm-up + %d timed, median%s)" % (
                n, warm, reps, "; BASELINE.md 4 asks 3 + 10" if config == 1 else ""),
            "protocol_detail": "batch %d, %d warm-up + %d timed steps, median" % (n, warm, reps),
            "sample": "oracle (numpy augmentation + PyTorch-CPU fp32) training step incl. the %s augmentation, %s/%s %dx%dx3 %d-class, "
                      "batch %d, median of %d timed steps after %d warm-up" % (cfg["augment"], cfg["architecture"], cfg["backbone"], size, size,
                                                                              classes, n, len(times), warm)}


class ClockSampler(object):
    """Samples the GPU's shader clock and socket power from the amdgpu hwmon files while a region runs (a host thread, 50 ms period; no
    subprocess, nothing on the device): the clocks a box HOLDS under the step's own load are what separates one box's step time from
    another's (`box_calibration.mfma_tflops` is the same 2.48 PFLOP/s on every box met: the idle-fabric MFMA clock does not differ).
    Everything is optional: unreadable files -> None."""

    def __init__(self, index=0):
        import glob
        self.files = {}
        cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        # the card of THIS HIP device: a box may show the sysfs nodes of every GPU of its host while one is visible to the process -
        # match the PCI address (domain:bus:device.function) of the device properties against the card's device link
        want = None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x." % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
        except Exception:
            want = None
        match = [c for c in cards if want and os.path.basename(os.path.realpath(os.path.join(c, "..", ".."))).startswith(want)]
        self.card = match[0] if match else None
        if match or len(cards) == 1:
            hw = match[0] if match else cards[0]
            for key, names in (("sclk_mhz", ("freq1_input",)), ("power_w", ("power1_average", "power1_input"))):
                for nm in names:
                    if os.path.exists(os.path.join(hw, nm)):
                        self.files[key] = os.path.join(hw, nm)
                        break
        self.samples = {k: [] for k in self.files}
        self._stop = None
        self._thread = None

    def __enter__(self):
        import threading
        if self.files:
            self._stop = threading.Event()

            def run():
                while not self._stop.is_set():
                    for k, f in self.files.items():
                        try:
                            with open(f) as fh:
                                self.samples[k].append(float(fh.read().strip()) * 1e-6)      # Hz -> MHz, microwatt -> W
                        except (OSError, ValueError):
                            pass
                    self._stop.wait(0.05)
            self._thread = threading.Thread(target=run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=1.0)
        return False

    def summary(self):
        out = {}
        for k, v in self.samples.items():
            if v:
                out[k] = round(float(np.mean(v)), 1)
                out[k + "_min"] = round(float(np.min(v)), 1)
                out[k + "_samples"] = len(v)
        return out or None


def box_calibration(dev, seconds=2.0):
    """What THIS box sustains on two fixed loads, measured before the timed region: boxes of the pool differ by +-2.5 %,
    more than a round's gain on the step, so a line is comparable with another line only next to these two numbers.
    ``mfma_tflops``: every SIMD issuing back-to-back v_mfma_f32_32x32x16 on register operands (csrc/calib.hip) for ~``seconds`` -
    the MFMA rate at the clocks the box holds under full matrix load (nominal 2516.6 at 2.4 GHz; ``sclk_mhz_from_mfma`` = that rate
    expressed as a clock); ``copy_gbs``: read + write bytes per second of a 1 GiB device copy with 16-byte accesses."""
    from segmentation_training_pipeline_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream()
    blocks, iters = 256 * 4, 4096
    out = torch.zeros(blocks * 512, dtype=torch.float32, device=dev)
    fl = int(lib.stp_calib_mfma_flops(blocks, iters))

    def mfma():
        _lib.check(lib.stp_calib_mfma(out.data_ptr(), blocks, iters, st.cuda_stream), "stp_calib_mfma")

    def timed_launches(fn, budget):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); fn(); e1.record(st)
        torch.cuda.synchronize()
        reps = max(3, int(budget / max(e0.elapsed_time(e1) * 1e-3, 1e-6)))
        evs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); fn(); b.record(st)
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) * 1e-3 for a, b in evs)
        return reps, ts
    reps, ts = timed_launches(mfma, seconds)
    last = ts[len(ts) // 2]                       # the median launch of the sustained region
    nbytes = 1 << 30
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.zero_()

    def copy():
        _lib.check(lib.stp_calib_copy(dst.data_ptr(), src.data_ptr(), nbytes, st.cuda_stream), "stp_calib_copy")
    creps, cts = timed_launches(copy, 0.5)
    cmed = cts[len(cts) // 2]
    tf = fl / last / 1e12
    return {"mfma_tflops": round(tf, 1), "mfma_tflops_first_launch": round(fl / ts[-1] / 1e12, 1), "mfma_launches": reps,
            "mfma_seconds": round(sum(ts), 2), "sclk_mhz_from_mfma": int(round(tf / 2516.6 * 2400.0)),
            "copy_gbs": round(2.0 * nbytes / cmed / 1e9, 1), "copy_launches": creps,
            "how": "csrc/calib.hip: 1024 workgroups x 8 waves x 65536 v_mfma_f32_32x32x16 per launch, median launch of ~%.0f s; 1 GiB device "
                   "copy (read + write bytes), median of %d" % (seconds, creps)}


def dice_delta_vs_oracle(device):
    """BASELINE.json's metric names "Dice delta vs ref": one training step of the headline network at its real resolution (U-Net/ResNet34,
    512 x 512, batch 2 - what one oracle step on the build container's CPU could pin) from the oracle's initial weights on the
    oracle's synthetic batch, against the Dice value of the fp32 CPU oracle's committed step (tests/golden/unet_resnet34_512_bs2.npz,
    tests/golden/make_golden.py --fullsize-only) - in the benchmarked precision AND in fp32 mode.  Part of the CPU-baseline /
    checker leg: the oracle supplies the initial weights and the batch, nothing here is timed."""
    from oracle import nets as onets
    from oracle import step as ostep
    from segmentation_training_pipeline_amd.backend import HipSegModel
    g = np.load(os.path.join(ROOT, "tests", "golden", "unet_resnet34_512_bs2.npz"))
    size, n = int(g["size"]), int(g["n"])
    x, y = ostep.synthetic_batch(n, size, size, seed=int(g["data_seed"]))
    P = onets.init_unet_resnet("resnet34", seed=int(g["seed"]))
    want_dice_loss, want_dice = float(g["scalars1"][2]), float(g["scalars1"][3])
    out = {"reference": "in-repo fp32 CPU oracle (parity unpinned by the reference: SURVEY 8c), U-Net/ResNet34 512x512 batch 2, first step",
           "oracle_dice": round(want_dice, 7), "north_star_bar": 1e-5}
    for dt in ("bf16", "fp32"):
        m = HipSegModel("Unet", "resnet34", (size, size, 3), 1, "sigmoid", batch=n, dtype=dt, loss=LOSS, optimizer="Adam", lr=1e-3,
                        use_graph=False, device=device)
        m.set_weights(P)
        met = m.train_on_batch(x, y)
        out[dt] = {"dice_delta": float("%.3g" % abs(met["dice"] - want_dice)), "dice_loss_delta": float("%.3g" % abs(met["dice_loss"] - want_dice_loss))}
        del m
    return out


def main():
    # stdout carries exactly ONE line (the JSON): libraries that print to the C-level stdout (RCCL's version banner, MIOpen) are
    # moved to stderr for the lifetime of the process, the result is written to the original descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                    help="BASELINE.json configs index: 1 = U-Net/ResNet34 512x512 bs16 bf16 (the headline, default; configs[2] with --gpus 8); "
                         "3 = FPN/ResNet50 1024x1024 3-class bs4 fp16; 4 = PSPNet/ResNet101 768x768 20-class bs8 bf16, heavy augmentation")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"],
                    help="default: the precision BASELINE.json names for the config (bf16 / fp16 / bf16).  bf16 = libstp_hip.so; fp16 = the "
                         "IEEE-half build of the same kernels (libstp_hip_f16.so, loss scale 2^14); fp32 = the parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="full", choices=["full", "short"],
                    help="config 1: full = BASELINE.md 4 as written (batch 16, 3 warm-up + 10 timed steps: ~5 min of host time); short = 1 + 3.  "
                         "configs 3 / 4: full = 1 + 4 steps of the config's batch, short = 1 + 2")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-calibration", action="store_true", help="skip the ~2.5 s box calibration (MFMA rate + copy bandwidth) before the timed region")
    ap.add_argument("--architecture", default=None, choices=["Unet", "Linknet", "FPN"],
                    help="config 1 only: Linknet / FPN over ResNet34 on the same shapes (SURVEY 8f N1 workloads, not the headline metric)")
    ap.add_argument("--eager", action="store_true", help="no hipGraph (for rocprofv3 kernel traces of the launches themselves)")
    ap.add_argument("--no-feed", action="store_true", help="skip the region fed from pinned host memory (profiler traces of the resident step only)")
    ap.add_argument("--sustain", type=float, default=8.0,
                    help="seconds of extra back-to-back steps after the counted ones, reported as `sustained` (0 = skip; 1-GPU runs only)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.architecture and args.config != 1:
        raise SystemExit("--architecture applies to --config 1")
    arch = args.architecture or cfg["architecture"]
    args.dtype = args.dtype or cfg["dtype"]
    H = W = cfg["size"]
    BATCH, classes = cfg["batch"], cfg["classes"]
    global _PMC
    _PMC = cfg["pmc"] if arch == cfg["architecture"] else "_" + arch.lower()

    from segmentation_training_pipeline_amd import augment, distributed, ops
    from segmentation_training_pipeline_amd.backend import HipSegModel

    rank, local_rank, world = distributed.env_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...` (WORLD_SIZE=%d)"
                         % (args.gpus, args.gpus, world))
    # the GPU of this rank: its own device; wrapped around when the node exposes fewer devices than ranks (the 2-rank dry run on
    # one GPU under STP_DIST_BACKEND=gloo, tests/test_dp_gpu.py) - on an 8-GPU node rank i drives GPU i
    dev_index = distributed.device_index(local_rank)
    torch.cuda.set_device(dev_index)
    force_dp = os.environ.get("STP_FORCE_DP") == "1"      # single-GPU exercise of the RCCL path
    # backend: RCCL ("nccl") unless STP_DIST_BACKEND names another one (gloo for several ranks on one GPU: RCCL refuses duplicate devices)
    distributed.init(os.environ.get("STP_DIST_BACKEND") or "nccl", force=force_dp)
    dev = torch.device("cuda", dev_index)

    # what this box sustains on two fixed loads (rank 0, before anything else runs on the device)
    calib = box_calibration(dev) if (rank == 0 and not args.no_calibration) else None
    if world > 1:
        dist.barrier()

    model = HipSegModel(arch, cfg["backbone"], (H, W, 3), classes, "sigmoid" if classes == 1 else "softmax", batch=BATCH, dtype=args.dtype,
                        loss=LOSS if classes == 1 else LOSS_SOFTMAX, optimizer="Adam", lr=1e-3, use_graph=not args.eager, device=str(dev))
    if world > 1 or force_dp:
        ov = os.environ.get("STP_DP_OVERLAP", "auto")
        model.set_data_parallel(distributed.make_reducer(force=force_dp), overlap={"0": False, "1": True, "auto": True}.get(ov, "buckets"))

    # synthetic data, seed 1234 + rank
    rng, img, msk = synthetic_data(rank, BATCH, H, classes)
    raw_img, raw_msk = torch.from_numpy(img).to(dev), torch.from_numpy(msk).to(dev)
    total = args.steps + args.warmup
    # the merged single-pass form of the config's pipeline: one stp_augment_u8 launch (warp + point operations, image and mask) and, for
    # S4, one stp_filter_u8 launch per neighbourhood filter (the Gaussian blur) through two ping-pong buffers
    spec = augment_spec(cfg["augment"])
    sampled = [augment.sample_batch_ex(spec, rng, BATCH, H, W, (H, W)) for _ in range(total)]
    prm = torch.from_numpy(np.stack([p for p, _ in sampled])).to(dev)
    npass = max((0 if f is None else f.shape[0]) for _, f in sampled)
    filt = None
    if npass:
        fr = np.zeros((total, npass, BATCH, augment.FILTER_RECORD), np.int32)       # (K = 0: a copy pass)
        for i, (_, f) in enumerate(sampled):
            if f is not None:
                fr[i, :f.shape[0]] = f
        filt = torch.from_numpy(fr).to(dev)
        fbuf = [torch.empty_like(raw_img) for _ in range(2)]
    in_img, in_msk = model.plan.inputs["image"].buf, model.plan.inputs["mask"].buf

    # The schedules of pipeline.Trainer.run_epoch_sums.  Default: augmentation, then the step, on one stream.  STP_FEED_OVERLAP=1 (opt-in,
    # measured slower: profiles/r04j_schedule_ab.txt): the augmentation kernel of the NEXT step is issued on an auxiliary stream once this
    # step's forward + backward has read the input buffers, and runs next to the optimizer.  Either way one step = one augmentation + one
    # forward / backward / optimizer.
    overlap = os.environ.get("STP_FEED_OVERLAP", "0") == "1"
    aux = torch.cuda.Stream(device=dev)

    def augment_into_plan(src_img, src_msk, i):
        if filt is None:
            ops.augment_u8(src_img, src_msk, in_img, in_msk, prm[i % total], BATCH, H, W, H, W, 3)
            return
        ops.augment_u8(src_img, src_msk, fbuf[0], in_msk, prm[i % total], BATCH, H, W, H, W, 3)
        for ps in range(npass):
            ops.filter_u8(fbuf[ps & 1], in_img if ps == npass - 1 else fbuf[1 - (ps & 1)], filt[i % total, ps], BATCH, H, W, 3)

    def step(i):
        if not overlap:
            augment_into_plan(raw_img, raw_msk, i)
            model.train_on_batch(None, None, fetch=False)
            return
        main = torch.cuda.current_stream()
        model.forward_backward()                     # (on the batch the previous step - or the priming call below - augmented)
        aux.wait_stream(main)
        with torch.cuda.stream(aux):
            augment_into_plan(raw_img, raw_msk, i + 1)
        model.apply_gradients()
        main.wait_stream(aux)

    dp_schedule = None
    if (world > 1 or force_dp) and os.environ.get("STP_DP_OVERLAP", "auto") == "auto":
        # overlapped vs serialised all-reduce: measured here, before the warm-up, on an augmented batch (10 steps each way; the state
        # is restored).  STP_DP_OVERLAP=0|1 skips the measurement.
        augment_into_plan(raw_img, raw_msk, 0)
        dp_schedule = model.calibrate_dp_schedule()
    if overlap:
        augment_into_plan(raw_img, raw_msk, 0)

    def timed(fn, first, count):
        """EXACTLY ``count`` steps bracketed by barrier + synchronize on both sides; MAX over ranks (seconds)."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(first, first + count):
            fn(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for i in range(args.warmup):
        step(i)
    elapsed = timed(step, args.warmup, args.steps)          # the contract's region: raw uint8 batch RESIDENT in HBM

    # ---- the network alone (no augmentation launch): the part of the step the launch tables / floor tables describe.  Configs 3 / 4 only
    # (config 1 keeps its established region list); the batch in the plan's input buffers is the last augmented one
    elapsed_model = None
    if args.config != 1 and not overlap:
        elapsed_model = timed(lambda i: model.train_on_batch(None, None, fetch=False), 0, args.steps)

    # ---- the same step FED from pinned host memory (north_star: "fed by pinned hipMemcpyAsync"): NHOST distinct raw batches in pinned
    # memory, THREE device staging buffers, a copy stream.  The host stays at most two steps ahead of the GPU (it waits for step i - 2
    # before it enqueues step i - what a data loader's bounded queue does anyway), so when it issues the H2D copy of batch i + 1 it KNOWS
    # the staging buffer (last read by the augmentation of step i - 2) is free, and when it enqueues the augmentation of step i the copy
    # of batch i - issued a whole step earlier - has landed: no GPU-side cross-stream wait is needed in steady state (an event wait is
    # enqueued only if the host-side query says the copy is still running).  Measured (profiles/r04n_feed_experiments.txt): with two
    # buffers and event waits both ways the fed step cost +0.42 ms although the copies themselves (228 + 81 us on the SDMA engine) cost
    # nothing (+0.03 ms without the waits): a cross-stream dependency next to a hipGraph launch costs ~0.1 ms on this runtime.
    # Reported next to the resident number (`value` stays the resident one, as the bench contract prescribes).
    NHOST, NBUF = 4, 3
    h_img = [torch.from_numpy(np.roll(img, k, axis=0).copy()).pin_memory() for k in range(NHOST)]
    h_msk = [torch.from_numpy(np.roll(msk, k, axis=0).copy()).pin_memory() for k in range(NHOST)]
    d_img = [torch.empty_like(raw_img) for _ in range(NBUF)]
    d_msk = [torch.empty_like(raw_msk) for _ in range(NBUF)]
    copy_stream = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(NBUF)]
    step_done = [torch.cuda.Event() for _ in range(NBUF)]

    def stage(i):
        b = i % NBUF
        with torch.cuda.stream(copy_stream):
            d_img[b].copy_(h_img[i % NHOST], non_blocking=True)
            d_msk[b].copy_(h_msk[i % NHOST], non_blocking=True)
            ready[b].record(copy_stream)

    def feed_into_plan(i, stream):
        b = i % NBUF
        if not ready[b].query():
            stream.wait_event(ready[b])                   # (not in steady state: the copy was issued a step ago)
        augment_into_plan(d_img[b], d_msk[b], i)

    def step_fed(i):
        main = torch.cuda.current_stream()
        step_done[(i - 2) % NBUF].synchronize()           # host throttle: step i - 2 has finished => staging buffer (i + 1) % 3 is free
        stage(i + 1)
        if not overlap:
            feed_into_plan(i, main)
            model.train_on_batch(None, None, fetch=False)
        else:
            model.forward_backward()
            aux.wait_stream(main)
            with torch.cuda.stream(aux):
                feed_into_plan(i + 1, aux)
            model.apply_gradients()
            main.wait_stream(aux)
        step_done[i % NBUF].record(main)

    torch.cuda.synchronize()
    for b in range(NBUF):
        step_done[b].record(torch.cuda.current_stream())
    stage(0)
    if overlap:
        stage(1)
        feed_into_plan(0, torch.cuda.current_stream())
    elapsed_fed = None
    if not args.no_feed:
        for i in range(2):
            step_fed(i)
        elapsed_fed = timed(step_fed, 2, args.steps)
    torch.cuda.synchronize()

    # ---- a LONG region of the resident step (world 1 only): the 20 counted steps last 0.15 s, which a 5 s utilisation sampler
    # misses entirely; ~8 s of back-to-back replays give an independent observer something to see, and a drift-free average
    sustained = None
    if world == 1 and args.sustain > 0:
        n_sus = max(200, int(args.sustain / max(elapsed / args.steps, 1e-4)))
        with ClockSampler(dev_index) as clocks:
            sustained = (n_sus, timed(step, 0, n_sus))
        sustained_clocks = clocks.summary()
    metrics = model.metrics()
    images_per_sec = world * BATCH * args.steps / elapsed

    net_name = "%s/%s" % ("U-Net" if arch == "Unet" else arch, cfg["backbone"].replace("resnet", "ResNet"))
    which = ("BASELINE.json configs[%d]" % args.config) if arch == cfg["architecture"] else "SURVEY 8f N1 workload, not the headline metric"
    loss_name = "BCE+Dice" if classes == 1 else "softmax CCE+Dice"
    out = {
        "metric": "images/sec %s %dx%d bs%d training step" % (net_name, H, W, BATCH), "value": round(images_per_sec, 2), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s %dx%dx3 %d-class, batch %d per GPU, %s, Adam, on-device augment %s (%s%s)"
                               % (net_name, H, W, classes, BATCH, loss_name, cfg["augment"], which,
                                  "; configs[2] data-parallel" if (world > 1 and args.config == 1) else ""),
                   "baseline_config_index": args.config,
                   "global_batch": BATCH * world, "parallelism": "dp%d" % world,
                   "hipgraph": not args.eager, "loss_after_run": round(metrics["loss"], 5), "dp_schedule": dp_schedule},
        # algorithmic FLOP per trained image = 3 x the forward conv FLOP the plan recorded (187.94 GFLOP for the U-Net)
        "step_mfma_frac": round(images_per_sec / world * flop_per_image(model) / (PEAK_BF16_TFLOPS * 1e12), 4),
        # the same K steps with every batch copied from pinned host memory (double-buffered hipMemcpyAsync on a copy stream)
        "ms_per_step_resident": round(1e3 * elapsed / args.steps, 3),
        "ms_per_step_without_augmentation": round(1e3 * elapsed_model / args.steps, 3) if elapsed_model else None,
        "ms_per_step_with_feed": round(1e3 * elapsed_fed / args.steps, 3) if elapsed_fed else None,
        "value_with_feed": round(world * BATCH * args.steps / elapsed_fed, 2) if elapsed_fed else None,
    }
    if sustained is not None:
        out["sustained"] = {"steps": sustained[0], "seconds": round(sustained[1], 3), "ms_per_step": round(1e3 * sustained[1] / sustained[0], 3),
                            "images_per_sec": round(BATCH * sustained[0] / sustained[1], 2),
                            # shader clock / socket power the box held over this region (amdgpu hwmon, 50 ms samples; None: not readable)
                            "clocks": sustained_clocks}
    if rank == 0 and not args.no_kernel_profile:
        prof = per_kernel_profile(model)
        tot = sum(v[1] for v in prof.values())
        gemm = {k: v for k, v in prof.items() if v[2] > 0}

        def roofline_of(key):
            n_l, sec, fl, _ = gemm[key]
            ach = fl / sec / 1e12
            traffic, traffic_src = pmc_traffic(key)
            mu, mu_src = pmc_mfma_util(key)
            return {"kernel": key, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src, "mfma_util": mu, "mfma_util_source": mu_src,
                    # the counter files come from separate profiler passes: true = collected on ANOTHER build of csrc/
                    "counters_stale": bool((traffic_src and _pmc_stale(os.path.join(ROOT, traffic_src))) or
                                           (mu_src and _pmc_stale(os.path.join(ROOT, mu_src)))),
                    "launches_per_step": round(n_l, 1), "avg_launch_us": round(1e6 * sec / n_l, 2),
                    "share_of_step_kernel_time": round(sec / tot, 3)}
        order = sorted(gemm, key=lambda k: -gemm[k][1])
        out["roofline"] = roofline_of(order[0])                      # the GEMM kernel with the largest total time per step
        out["roofline_next"] = [roofline_of(k) for k in order[1:4]]  # and the three after it (same fields)

        # the memory-bound half of the step: the non-GEMM kernel family (HBM_FAMILIES: entry points that launch the same device kernels)
        # with the largest time per step; achieved = algorithmic bytes (hbm_bytes: every operand tensor once) / HIP-event time
        def roofline_hbm_of(family):
            ent = [prof[e] for e in HBM_FAMILIES[family]["entries"] if e in prof]
            n_l, sec, byt = sum(v[0] for v in ent), sum(v[1] for v in ent), sum(v[3] for v in ent)
            ach = byt / sec / 1e9
            traffic, traffic_src, fam = pmc_family_traffic(family)
            return {"kernel": family, "entry_points": [e for e in HBM_FAMILIES[family]["entries"] if e in prof], "bound": "hbm",
                    "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                    "algorithmic_bytes_per_launch": int(byt / n_l), "traffic": traffic, "traffic_unit": "bytes/launch (L2-miss: FETCH_SIZE + WRITE_SIZE passes)",
                    "traffic_kernels": fam, "traffic_source": traffic_src,
                    "counters_stale": bool(traffic_src and _pmc_stale(os.path.join(ROOT, traffic_src))),
                    "launches_per_step": round(n_l, 1), "avg_launch_us": round(1e6 * sec / n_l, 2),
                    "share_of_step_kernel_time": round(sec / tot, 3)}
        fams = [f for f in HBM_FAMILIES if any(e in prof for e in HBM_FAMILIES[f]["entries"])]
        fams.sort(key=lambda f: -sum(prof[e][1] for e in HBM_FAMILIES[f]["entries"] if e in prof))
        if fams:
            out["roofline_hbm"] = roofline_hbm_of(fams[0])
            out["roofline_hbm_next"] = [roofline_hbm_of(f) for f in fams[1:3]]
        st_bytes, st_src = pmc_step_traffic()
        if st_bytes:
            out["step_traffic_gb"] = round(st_bytes / 1e9, 2)        # sum of L2-miss bytes of every kernel of one step (PMC passes)
            out["step_traffic_source"] = st_src
            out["step_traffic_stale"] = bool(_pmc_stale(os.path.join(ROOT, st_src)))
            out["step_hbm_floor_ms"] = round(st_bytes / (PEAK_HBM_GBS * 1e9) * 1e3, 3)       # that traffic at the 8 TB/s peak
            out["step_hbm_floor_ms_at_6300"] = round(st_bytes / 6.3e12 * 1e3, 3)             # ... at the rate a device copy reaches
            # the whole step against the HBM roofline: L2-miss bytes of one step / the measured step time / 8 TB/s (next to step_mfma_frac:
            # PSPNet/ResNet101 and FPN/ResNet50 are HBM-bound as whole steps, the U-Net is bound by neither)
            out["step_hbm_frac"] = round(st_bytes / (elapsed / args.steps) / (PEAK_HBM_GBS * 1e9), 4)
        top = sorted(prof.items(), key=lambda kv: -kv[1][1])[:16]
        out["kernel_time_us"] = {k: [round(v[0], 1), round(1e6 * v[1], 1), round(v[2] / v[1] / 1e12, 1) if v[2] else None] for k, v in top}
        out["kernel_time_total_us"] = round(1e6 * tot, 1)
        out["gemm_time_us"] = round(1e6 * sum(v[1] for v in gemm.values()), 1)
        out["non_gemm_time_us"] = round(1e6 * (tot - sum(v[1] for v in gemm.values())), 1)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.dtype == "bf16" and arch == "Unet" and args.config == 1:
            out["config"]["dice_delta_vs_oracle"] = dice_delta_vs_oracle(str(dev))
        if arch == cfg["architecture"]:
            out["cpu_baseline"] = cpu_baseline(full_protocol=args.cpu_baseline == "full", config=args.config)
    if calib is not None:
        out["box_calibration"] = calib
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
