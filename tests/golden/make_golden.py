"""Regenerates the committed golden fixtures.  Run from the repo root:

    python tests/golden/make_golden.py

* ``rle_golden.json`` - inputs/outputs of the REFERENCE's own
  ``segmentation_pipeline/impl/rle.py:10-35`` (``rle_encode`` / ``rle_decode``), imported from
  ``/root/reference`` with a stub for the absent ``skimage`` (only ``multi_rle_encode`` uses it).
  This is the only part of the reference that can execute in this container.
* ``unet_resnet18_64.npz`` / ``unet_resnet34_64.npz`` / ``linknet_resnet18_64.npz`` / ``fpn_resnet18_64.npz`` /
  ``pspnet_resnet18_96.npz`` - outputs of the in-repo oracle (PARITY
  UNPINNED: the reference's Keras path is not runnable) on a seeded synthetic batch; they pin
  the oracle against drift of itself / of the torch build.
* ``fpn_resnet50_128_fp16.npz`` / ``pspnet_resnet101_96_bf16.npz`` / ``linknet_resnet34_128_bf16.npz`` (``--storage-cases-only``) - the
  STORAGE-QUANTISED oracle (rounds where the kernels round) on BASELINE.json configs[3] / [4]'s graphs and Linknet at their benchmarked
  precision, at sizes the CPU finishes in seconds.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def make_rle():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("reference absent; keeping existing rle_golden.json")
        return
    sk = types.ModuleType("skimage")
    skm = types.ModuleType("skimage.morphology")
    skm.label = lambda a: a
    sk.morphology = skm
    sys.modules["skimage"] = sk
    sys.modules["skimage.morphology"] = skm
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_rle", os.path.join(ref, "segmentation_pipeline/impl/rle.py"))
    ref_rle = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_rle)
    rng = np.random.RandomState(7)
    cases = []
    shapes = [(6, 6), (5, 9), (16, 16), (1, 7), (7, 1), (32, 20)]
    for shp in shapes:
        for dens in (0.0, 0.15, 0.5, 1.0):
            m = (rng.uniform(size=shp) < dens).astype(np.uint8)
            enc = ref_rle.rle_encode(m)
            dec = ref_rle.rle_decode(enc, shp) if enc else np.zeros(shp, np.uint8).T
            cases.append({"shape": list(shp), "mask": m.tolist(), "rle": enc,
                          "decoded_shape": list(dec.shape), "decoded": dec.tolist()})
    m = np.zeros((6, 6), np.uint8)
    m[2:5, 1:3] = 1
    cases.append({"shape": [6, 6], "mask": m.tolist(), "rle": ref_rle.rle_encode(m),
                  "decoded_shape": [6, 6], "decoded": ref_rle.rle_decode(ref_rle.rle_encode(m), (6, 6)).tolist()})
    with open(os.path.join(HERE, "rle_golden.json"), "w") as f:
        json.dump({"source": "reference segmentation_pipeline/impl/rle.py:10-35", "cases": cases}, f)
    print("rle cases", len(cases))


def make_unet(backbone, size, n, fname, arch="Unet"):
    from oracle import nets, step
    P = {"Unet": nets.init_unet_resnet, "Linknet": nets.init_linknet_resnet, "FPN": nets.init_fpn_resnet,
         "PSPNet": nets.init_pspnet_resnet}[arch](backbone, seed=42)
    tr = step.OracleTrainer(P, backbone=backbone, loss="binary_crossentropy+1.0*dice_loss",
                            optimizer="adam", lr=1e-3, architecture=arch)
    x, y = step.synthetic_batch(n, size, size, seed=1234)
    xf, yf = x.astype(np.float32), y.astype(np.float32)
    o1 = tr.step(xf, yf)
    o2 = tr.step(xf, yf)
    names = list(o1["grads"].keys())
    np.savez_compressed(
        os.path.join(HERE, fname),
        x=x, y=y, seed=42,
        logits1=o1["logits"].astype(np.float32), logits2=o2["logits"].astype(np.float32),
        scalars1=np.array([o1[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], np.float64),
        scalars2=np.array([o2[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], np.float64),
        grad_names=np.array(names),
        grad_l2_step1=np.array([np.sqrt((o1["grads"][k].astype(np.float64) ** 2).sum()) for k in names]),
        param_sum_after2=np.array([tr.P[k].astype(np.float64).sum() for k in names]),
        param_abs_after2=np.array([np.abs(tr.P[k].astype(np.float64)).sum() for k in names]),
    )
    print(fname, "loss", o1["loss"], o2["loss"], "dice", o1["dice"])


def make_unet_fullsize(fname="unet_resnet34_512_bs2.npz", size=512, n=2, stride=4):
    """BASELINE.json configs[1]'s network at its real resolution (512 x 512; batch 2 keeps the CPU step in seconds): ONE training step
    of the oracle.  The inputs are regenerated from the seed by the test (step.synthetic_batch), the logits are kept on a
    stride-4 pixel grid (the full map is 2 MB), plus the scalars and every parameter's gradient norm."""
    from oracle import nets, step
    P = nets.init_unet_resnet("resnet34", seed=42)
    tr = step.OracleTrainer(P, backbone="resnet34", loss="binary_crossentropy+1.0*dice_loss", optimizer="adam", lr=1e-3)
    x, y = step.synthetic_batch(n, size, size, seed=1234)
    o1 = tr.step(x.astype(np.float32), y.astype(np.float32))
    names = list(o1["grads"].keys())
    lg = o1["logits"].astype(np.float32)
    np.savez_compressed(
        os.path.join(HERE, fname), seed=42, data_seed=1234, size=size, n=n, stride=stride,
        logits1_sampled=lg[:, ::stride, ::stride, :],
        logits1_sum=np.float64(lg.astype(np.float64).sum()), logits1_abs_sum=np.float64(np.abs(lg.astype(np.float64)).sum()),
        logits1_row_sums=lg.astype(np.float64).sum(axis=(2, 3)),
        scalars1=np.array([o1[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], np.float64),
        grad_names=np.array(names),
        grad_l2_step1=np.array([np.sqrt((o1["grads"][k].astype(np.float64) ** 2).sum()) for k in names]),
        param_sum_after1=np.array([tr.P[k].astype(np.float64).sum() for k in names]),
    )
    print(fname, "loss", o1["loss"], "dice", o1["dice"], "logit range", lg.min(), lg.max())


def make_unet_fullsize_storage(storage="bf16", size=512, n=2, stride=4):
    """The BENCHMARKED precision at the benchmarked shape: U-Net/ResNet34, 512 x 512 (batch 2), one forward + backward of the
    STORAGE-QUANTISED oracle (``OracleTrainer(storage="bf16")``: stored activations, stored gradients and weight compute copies are
    rounded where the kernels round them; statistics, accumulation and the loss stay fp32).  Same sampling as make_unet_fullsize, plus
    the gradients of a few layers in full (cosine bars) and every parameter's gradient norm."""
    from oracle import nets, step
    P = nets.init_unet_resnet("resnet34", seed=42)
    tr = step.OracleTrainer(P, backbone="resnet34", loss="binary_crossentropy+1.0*dice_loss", optimizer="adam", lr=1e-3, storage=storage)
    x, y = step.synthetic_batch(n, size, size, seed=1234)
    taps = {}
    o1 = tr.step(x.astype(np.float32), y.astype(np.float32), apply=False, taps=taps)
    names = list(o1["grads"].keys())
    lg = o1["logits"].astype(np.float32)
    # early activations on a stride-16 pixel grid: before rounding ties start to cascade the device must reproduce them EXACTLY
    early = {"tap_bn0": taps["relu0"], "tap_pooling0": taps["pooling0"], "tap_stage1_unit1_conv2": taps["stage1_unit1_out"]}
    early = {k: v.detach().float().numpy()[:, ::16, ::16, :].astype(np.float32) for k, v in early.items()}
    keep = ["final_conv/kernel", "decoder_stage4_conv2/kernel", "decoder_stage2_conv1/kernel", "decoder_stage0_bn1/gamma",
            "stage4_unit3_conv2/kernel", "stage3_unit1_bn2/beta", "stage2_unit2_conv1/kernel", "stage1_unit1_conv1/kernel", "conv0/kernel"]
    fname = "unet_resnet34_512_bs2_%s.npz" % storage
    np.savez_compressed(
        os.path.join(HERE, fname), seed=42, data_seed=1234, size=size, n=n, stride=stride, storage=storage,
        logits1_sampled=lg[:, ::stride, ::stride, :],
        logits1_abs_max=np.float64(np.abs(lg).max()),
        logits1_row_sums=lg.astype(np.float64).sum(axis=(2, 3)),
        scalars1=np.array([o1[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], np.float64),
        grad_names=np.array(names),
        grad_l2_step1=np.array([np.sqrt((o1["grads"][k].astype(np.float64) ** 2).sum()) for k in names]),
        full_grad_names=np.array(keep),
        **early,
        **{"grad_full_%d" % i: o1["grads"][k].astype(np.float32) for i, k in enumerate(keep)})
    print(fname, "loss", o1["loss"], "dice", o1["dice"], "logit range", lg.min(), lg.max())


# BASELINE.json configs[3] / [4] (and Linknet) AT THEIR BENCHMARKED PRECISION, at sizes the oracle finishes in seconds:
# (architecture, backbone, size, classes, storage, gradient scale of the stored gradients = the fp16 build's static loss scale)
STORAGE_CASES = [("FPN", "resnet50", 128, 3, "fp16", 16384.0), ("PSPNet", "resnet101", 96, 20, "bf16", 1.0), ("Linknet", "resnet34", 128, 1, "bf16", 1.0)]


def storage_case_inputs(size, classes, n=2):
    """Seeded batch of a storage case: S1 images; the 1-class mask of S1, or a blocky class-index map for the softmax heads."""
    from oracle import step
    x, y = step.synthetic_batch(n, size, size, seed=1234)
    if classes > 1:
        yy, xx = np.mgrid[0:size, 0:size]
        y = ((yy // 8 + 3 * (xx // 12)) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    return x, y


def storage_case_spec(classes):
    return ("sigmoid", "binary_crossentropy+1.0*dice_loss") if classes == 1 else ("softmax", "categorical_crossentropy+1.0*dice_loss")


def make_storage_case(arch, backbone, size, classes, storage, grad_scale):
    """One forward + backward of the STORAGE-QUANTISED oracle (``OracleTrainer(storage=...)``) for a non-U-Net graph, next to the fp32
    oracle's logits of the same step (the test shows the device closer to the former): logits, scalars, every gradient norm, a few
    gradients in full."""
    from oracle import nets, step
    init = {"Linknet": nets.init_linknet_resnet, "FPN": nets.init_fpn_resnet, "PSPNet": nets.init_pspnet_resnet}[arch]
    P = init(backbone, classes=classes, seed=42)
    x, y = storage_case_inputs(size, classes)
    act, spec = storage_case_spec(classes)
    kw = dict(backbone=backbone, loss=spec, optimizer="adam", lr=1e-3, architecture=arch, activation=act)
    o = step.OracleTrainer(P, storage=storage, grad_scale=grad_scale, **kw).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    o32 = step.OracleTrainer(P, **kw).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    names = list(o["grads"].keys())
    # gradients kept in full (cosine bars): the class convolution, the stem, and the small convolution kernels / BatchNormalization
    # vectors spread over the depth of the graph (every kernel above 40 K elements is represented by its norm only: fixture size)
    small = [k for k in names if o["grads"][k].size <= 40000 and (k.endswith("/kernel") or k.endswith("/gamma"))]
    keep = ["final_conv/kernel", "conv0/kernel"] + [small[i * (len(small) - 1) // 7] for i in range(8)]
    keep = list(dict.fromkeys(keep))
    fname = "%s_%s_%d_%s.npz" % (arch.lower(), backbone, size, storage)
    np.savez_compressed(
        os.path.join(HERE, fname), seed=42, arch=arch, backbone=backbone, size=size, classes=classes, storage=storage, grad_scale=grad_scale,
        x=x, y=y, logits1=o["logits"].astype(np.float32), logits1_fp32=o32["logits"].astype(np.float16),
        scalars1=np.array([o[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], np.float64),
        grad_names=np.array(names),
        grad_l2_step1=np.array([np.sqrt((o["grads"][k].astype(np.float64) ** 2).sum()) for k in names]),
        full_grad_names=np.array(keep),
        **{"grad_full_%d" % i: o["grads"][k].astype(np.float32) for i, k in enumerate(keep)})
    print(fname, "loss", o["loss"], "(fp32 oracle %.5f)" % o32["loss"], "logit range", float(np.abs(o["logits"]).max()))


if __name__ == "__main__":
    if "--storage-cases-only" in sys.argv:
        for c in STORAGE_CASES:
            make_storage_case(*c)
        sys.exit(0)
    if "--fullsize-only" in sys.argv:
        make_unet_fullsize()
        sys.exit(0)
    if "--fullsize-bf16-only" in sys.argv:
        make_unet_fullsize_storage("bf16")
        sys.exit(0)
    make_rle()
    make_unet("resnet18", 64, 2, "unet_resnet18_64.npz")
    make_unet("resnet34", 64, 2, "unet_resnet34_64.npz")
    make_unet("resnet18", 64, 2, "linknet_resnet18_64.npz", arch="Linknet")
    make_unet("resnet18", 64, 2, "fpn_resnet18_64.npz", arch="FPN")
    make_unet("resnet18", 96, 2, "pspnet_resnet18_96.npz", arch="PSPNet")
    make_unet_fullsize()
    make_unet_fullsize_storage("bf16")
    for c in STORAGE_CASES:
        make_storage_case(*c)
