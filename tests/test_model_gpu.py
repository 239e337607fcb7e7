"""GPU parity tests of the whole training step (U-Net over ResNet) against the CPU oracle and the
committed golden fixtures.  Bars (BASELINE.json north_star): logits within 1e-3, Dice within 1e-5
in the exact-fp32 mode; the bf16 mode (the benchmarked precision) is held to the documented
bf16 tolerances below.  Parity is vs the in-repo oracle: the reference's Keras path cannot run.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import nets as onets  # noqa: E402
from oracle import step as ostep  # noqa: E402

LOSS = "binary_crossentropy+1.0*dice_loss"
TAP_MAP = [("bn_data", "bn_data"), ("conv0", "conv0"), ("relu0", "bn0"), ("pooling0", "pooling0"),
           ("stage1_unit1_relu1", "stage1_unit1_bn1"), ("stage1_unit1_out", "stage1_unit1_conv2"),
           ("stage2_unit1_out", "stage2_unit1_conv2"), ("stage3_unit1_out", "stage3_unit1_conv2"),
           ("stage4_unit1_out", "stage4_unit1_conv2"), ("relu1", "bn1"),
           ("decoder_stage0_relu2", "decoder_stage0_bn2"), ("decoder_stage2_relu2", "decoder_stage2_bn2"),
           ("decoder_stage4_relu2", "decoder_stage4_bn2"),
           ("fpn_stage0_out", "fpn_stage0_segm2_bn"), ("fpn_stage1_out", "fpn_stage1_segm2_bn"),
           ("fpn_stage2_out", "fpn_stage2_segm2_bn"), ("fpn_stage3_out", "fpn_stage3_segm2_bn"),
           ("block1_out", "block1_conv2"), ("block2_out", "block2_conv2"), ("block3_out", "block3_conv3"),
           ("block4_out", "block4_conv3"), ("block5_out", "block5_conv3")]


def make(backbone, size, n, dtype, use_graph=False, **kw):
    from segmentation_training_pipeline_amd.backend import HipSegModel
    return HipSegModel(kw.pop("architecture", "Unet"), backbone, (size, size, 3), 1, "sigmoid", batch=n, dtype=dtype, loss=kw.pop("loss", LOSS),
                       optimizer=kw.pop("optimizer", "Adam"), lr=kw.pop("lr", 1e-3), use_graph=use_graph, **kw)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def first_bad_tap(model, taps, atol):
    """Localises a mismatch: walks the network in forward order."""
    for oname, pname in TAP_MAP:
        if oname not in taps:
            continue
        ref = taps[oname].detach().numpy()
        if oname.endswith("_out") and pname.endswith("conv2") and pname[:-1] + "3" in model.plan.tensors:
            pname = pname[:-1] + "3"                      # bottleneck units end in conv3
        got = model.activation(pname)[..., :ref.shape[-1]]
        err = np.abs(got - ref).max()
        if not err <= atol * max(1.0, np.abs(ref).max()):
            return "%s: max err %.3g (ref max %.3g)" % (pname, err, np.abs(ref).max())
    return None


@pytest.mark.parametrize("arch,backbone", [("Unet", "resnet18"), ("Unet", "resnet34"), ("Linknet", "resnet18"), ("Linknet", "resnet34"),
                                           ("Unet", "resnet50"), ("Linknet", "resnet50"),      # bottleneck encoders: 1x1 / 3x3 / 1x1
                                           ("FPN", "resnet18"), ("FPN", "resnet50")])         # BASELINE configs[3] family
def test_fp32_step_matches_oracle(arch, backbone):
    # (the 2048-channel bottleneck encoders get 128 px: at 64 px their last BatchNormalization sees 8 values per channel
    #  and the fp32 round-off of the two implementations, amplified by 1/sigma, exceeds the per-tap debugging tolerance)
    n, size = 2, (128 if backbone == "resnet50" else 64)
    P = {"Unet": onets.init_unet_resnet, "Linknet": onets.init_linknet_resnet, "FPN": onets.init_fpn_resnet}[arch](backbone, seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=LOSS, optimizer="sgd", lr=0.05, opt_kwargs={"momentum": 0.9}, architecture=arch)
    m = make(backbone, size, n, "fp32", optimizer="SGD", lr=0.05, opt_kwargs={"momentum": 0.9}, architecture=arch)
    assert sorted(m.get_weights()) == sorted(P)            # same parameter names and shapes as the oracle's Keras layout
    m.set_weights(P)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    # forward + backward, taps, THEN the optimizer: a BatchNormalization output that lives only inside its consumers' staging
    # is materialised on demand with the current gamma / beta (graph.Plan.tensor), so it must be read before they move
    m.load_batch(x, y)
    m.forward_backward()
    bad = first_bad_tap(m, taps, 2e-4)
    m.apply_gradients()
    met = m.metrics()
    assert bad is None, bad
    # north-star bar 1e-3 (stated for U-Net/ResNet34).  Linknet over the 2048-channel encoder - 3 x (conv, BN over as few as
    # 128 values) per decoder stage on top of 50 layers - lands at 1.3e-3 between two fp32 summation orders: 2e-3 there.
    latol = 2e-3 if (arch, backbone) in (("Linknet", "resnet50"), ("FPN", "resnet50")) else 1e-3
    np.testing.assert_allclose(m.logits(), o["logits"], atol=latol)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5                             # north-star bar
    # `dice` is the THRESHOLDED metric: a pixel whose probability is within the logit tolerance of 0.5 may fall on either
    # side, and one such pixel moves it by ~2/(sum y + sum t); allow two flips on top of the 1e-5 bar
    flips = 2 * 2.0 / (2.0 * float(y.sum()) + 1.0)
    assert abs(met["dice"] - o["dice"]) < 1e-5 + flips
    assert abs(met["loss"] - o["loss"]) < 1e-5 * max(1.0, abs(o["loss"]))
    assert abs(met["binary_crossentropy"] - o["bce"]) < 1e-5
    assert abs(met["binary_accuracy"] - o["binary_accuracy"]) < 1e-6 + 2.0 / y.size       # thresholded too: two flips
    # Gradients.  Two fp32 implementations of a ReLU network cannot agree element-wise on the sign of
    # pre-activations that are within rounding (~1e-6) of zero; with ~1e6 activations about one such
    # kink flips per step, and a single flip perturbs every upstream gradient by O(1e-3) relative L2
    # (measured: HIP's dx equals an fp64 recomputation from its own buffers to 5e-8).  So: layers after
    # the last ReLU-BN pair must agree tightly, all others in relative L2.
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        tight = k.startswith("final_conv")          # depends only on dL/dlogits and the last activation
        assert e <= (1e-4 if tight else (6e-2 if backbone == "resnet50" else 3e-2)), "grad %s: rel L2 %.3g" % (k, e)
    w = m.get_weights()
    watol = 3e-3 if backbone == "resnet50" else 2e-4     # three times as many ReLU layers to flip a kink in (lr 0.05 x 6e-2)
    for k in tr.P:   # lr (0.05) x gradient kink noise (<= 3e-2 relative L2) bounds the post-step difference
        np.testing.assert_allclose(w[k], tr.P[k], atol=watol, err_msg=k)
    # second step: re-synchronise the weights first (the kink noise above times lr would otherwise be
    # amplified by the next forward), then the forward must again agree to the 1e-3 bar and the
    # momentum update must carry over.
    m.set_weights(tr.P)
    o2 = tr.step(x.astype(np.float32), y.astype(np.float32))
    met2 = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o2["logits"], atol=latol)
    assert abs(met2["dice_loss"] - o2["dice_loss"]) < 1e-5
    w = m.get_weights()
    for k in tr.P:
        np.testing.assert_allclose(w[k], tr.P[k], atol=watol, err_msg=k)


@pytest.mark.parametrize("arch,backbone,size", [("Unet", "resnet34", 64), ("Linknet", "resnet18", 64), ("FPN", "resnet18", 64),
                                                ("PSPNet", "resnet18", 96)])
def test_fp32_adam_step_matches_golden_fixture(golden_dir, arch, backbone, size):
    g = np.load(os.path.join(golden_dir, "%s_%s_%d.npz" % (arch.lower(), backbone, size)))
    P = {"Unet": onets.init_unet_resnet, "Linknet": onets.init_linknet_resnet, "FPN": onets.init_fpn_resnet,
         "PSPNet": onets.init_pspnet_resnet}[arch](backbone, seed=int(g["seed"]))
    m = make(backbone, size, 2, "fp32", architecture=arch)
    m.set_weights(P)
    met = m.train_on_batch(g["x"], g["y"])
    np.testing.assert_allclose(m.logits(), g["logits1"], atol=1e-3)
    loss, bce, dice_loss, dice, acc = g["scalars1"]
    assert abs(met["dice_loss"] - dice_loss) < 1e-5 and abs(met["dice"] - dice) < 1e-5
    assert abs(met["loss"] - loss) < 2e-5
    names = [str(s) for s in g["grad_names"]]
    got = m.get_gradients()
    l2 = np.array([np.sqrt((got[k].astype(np.float64) ** 2).sum()) for k in names])
    np.testing.assert_allclose(l2, g["grad_l2_step1"], rtol=5e-2, atol=1e-6)   # ReLU-kink noise, see above
    # Adam's first update is lr*sign(g) wherever |g| >> eps: elements whose gradient is at rounding
    # level legitimately differ by up to 2*lr, and the next forward amplifies that, so the second
    # step is held to a statistical band only.
    met2 = m.train_on_batch(g["x"], g["y"])
    d = np.abs(m.logits() - g["logits2"])
    assert d.mean() < 4e-2 and np.corrcoef(m.logits().ravel(), g["logits2"].ravel())[0, 1] > 0.995
    assert abs(met2["loss"] - g["scalars2"][0]) < 2e-2


def test_fp32_fullsize_512_step_matches_golden_fixture(golden_dir):
    """BASELINE.json configs[1]'s network at its REAL resolution - U-Net/ResNet34, 512 x 512, batch 2 - against the oracle's
    committed outputs (tests/golden/unet_resnet34_512_bs2.npz, one CPU step in the build container): the tile selection, halo
    tiles and grouped weight gradients of the headline shapes at the north-star bars (logits 1e-3, Dice 1e-5), not at 64 px."""
    g = np.load(os.path.join(golden_dir, "unet_resnet34_512_bs2.npz"))
    size, n, stride = int(g["size"]), int(g["n"]), int(g["stride"])
    x, y = ostep.synthetic_batch(n, size, size, seed=int(g["data_seed"]))
    m = make("resnet34", size, n, "fp32")
    m.set_weights(onets.init_unet_resnet("resnet34", seed=int(g["seed"])))
    met = m.train_on_batch(x, y)
    lg = m.logits()
    np.testing.assert_allclose(lg[:, ::stride, ::stride, :], g["logits1_sampled"], atol=1e-3)
    np.testing.assert_allclose(lg.astype(np.float64).sum(axis=(2, 3)), g["logits1_row_sums"], atol=1e-3 * size)   # every pixel enters a row sum
    assert abs(np.abs(lg.astype(np.float64)).sum() - float(g["logits1_abs_sum"])) < 1e-4 * float(g["logits1_abs_sum"])
    loss, bce, dice_loss, dice, acc = g["scalars1"]
    assert abs(met["dice_loss"] - dice_loss) < 1e-5 and abs(met["dice"] - dice) < 1e-5
    assert abs(met["loss"] - loss) < 2e-5 and abs(met["binary_crossentropy"] - bce) < 2e-5
    names = [str(s) for s in g["grad_names"]]
    got = m.get_gradients()
    l2 = np.array([np.sqrt((got[k].astype(np.float64) ** 2).sum()) for k in names])
    np.testing.assert_allclose(l2, g["grad_l2_step1"], rtol=5e-2, atol=1e-6)   # ReLU-kink noise


def test_bf16_fullsize_512_step_matches_storage_quantised_golden(golden_dir, monkeypatch):
    """The BENCHMARKED precision at the benchmarked shape (U-Net/ResNet34, 512 x 512, bf16; batch 2): bench.py's kernels - halo
    tiles, grouped weight gradients, small-channel streaming kernels with their fused producer BatchNormalization - against the
    storage-quantised oracle's committed outputs (tests/golden/unet_resnet34_512_bs2_bf16.npz: ``OracleTrainer(storage="bf16")``,
    one forward + backward on the build container's CPU).  Bars in storage ulps, as in
    test_16bit_step_matches_the_storage_quantised_oracle: exact agreement where no rounding tie has cascaded yet, measured bars after."""
    monkeypatch.setenv("STP_UPCOLLAPSE", "0")      # (class-collapsed weight copies: a rounding point the oracle does not have)
    g = np.load(os.path.join(golden_dir, "unet_resnet34_512_bs2_bf16.npz"))
    size, n, stride = int(g["size"]), int(g["n"]), int(g["stride"])
    x, y = ostep.synthetic_batch(n, size, size, seed=int(g["data_seed"]))
    m = make("resnet34", size, n, "bf16")
    m.set_weights(onets.init_unet_resnet("resnet34", seed=int(g["seed"])))
    m.load_batch(x, y)
    m.forward_backward()
    met = m.metrics()
    lg = m.logits()
    # (1) before rounding ties cascade the device reproduces the oracle's stored activations EXACTLY: stem BatchNormalization + ReLU,
    # max-pooling, and (>= 95 % of the values, the rest one storage ulp away) the first residual unit
    for tap, name, exact_min, ulps in (("tap_bn0", "bn0", 0.999, 1.0), ("tap_pooling0", "pooling0", 0.999, 1.0),
                                       ("tap_stage1_unit1_conv2", "stage1_unit1_conv2", 0.95, 2.0)):
        want = g[tap]
        have = m.activation(name)[:, ::16, ::16, :want.shape[-1]]
        u = 2.0 ** (np.floor(np.log2(float(np.abs(want).max()))) - 7)
        d = np.abs(have - want)
        print("%-22s exact %.4f  max %.2f ulp" % (name, float((d == 0).mean()), d.max() / u))
        assert (d == 0).mean() >= exact_min and d.max() <= ulps * u, (name, float((d == 0).mean()), d.max() / u)
    # (2) logits.  Measured (MI355X, scratch/r04/dbg_bf16_512.py walks the taps): the two bf16 computations agree bit for bit through
    # the stem, 98.6 % through stage 1, then fp32 summation order decides rounding ties - each a one-ulp flip that the following
    # ~50 layers spread: mean 3.3 ulp of the logit range at 512 px (ResNet34 at 64 px: 1.0; ResNet18: 0.6), max 32 ulp, while BOTH
    # sit 6.6 ulp from the fp32 oracle.  Bars: mean <= 4.5 / max <= 48 ulp, and at least 1.6x closer to the oracle that shares the
    # rounding points than to the fp32 oracle's committed logits (tests/golden/unet_resnet34_512_bs2.npz).
    ref = g["logits1_sampled"]
    rng_ = float(g["logits1_abs_max"])
    ulp = 2.0 ** (np.floor(np.log2(rng_)) - 7)
    err = np.abs(lg[:, ::stride, ::stride, :] - ref)
    g32 = np.load(os.path.join(golden_dir, "unet_resnet34_512_bs2.npz"))
    err32 = np.abs(lg[:, ::stride, ::stride, :] - g32["logits1_sampled"])
    print("512 x 512 bf16 vs storage-quantised golden: range %.3f ulp %.4g max %.2f ulp mean %.3f ulp; vs the fp32 golden mean %.3f ulp"
          % (rng_, ulp, err.max() / ulp, err.mean() / ulp, err32.mean() / ulp))
    assert err.max() <= 48.0 * ulp and err.mean() <= 4.5 * ulp, (err.max() / ulp, err.mean() / ulp)
    assert err.mean() * 1.6 < err32.mean(), (err.mean() / ulp, err32.mean() / ulp)
    rs = np.abs(lg.astype(np.float64).sum(axis=(2, 3)) - g["logits1_row_sums"])
    assert rs.max() <= 4.5 * ulp * size, rs.max() / (ulp * size)             # every pixel enters a row sum: |mean error| per pixel <= 4.5 ulp
    loss, bce, dice_loss, dice, acc = g["scalars1"]
    print("loss %.5f (golden %.5f)  dice_loss %.5f (%.5f)" % (met["loss"], loss, met["dice_loss"], dice_loss))
    assert abs(met["loss"] - loss) < 5e-3 and abs(met["dice_loss"] - dice_loss) < 2e-3
    got = m.get_gradients()
    names = [str(s) for s in g["grad_names"]]
    l2 = np.array([np.sqrt((got[k].astype(np.float64) ** 2).sum()) for k in names])
    # norms within 15 % + 0.2 % of the largest norm (BatchNormalization gradients are differences of large cancelling sums: the few
    # parameters whose gradient norm is ~1e-3 of the largest move by up to 50 % between two bf16 computations)
    gmax = float(g["grad_l2_step1"].max())
    print("gradient norms: worst |difference| / (0.15 |ref| + 0.002 max) = %.3f" % float((np.abs(l2 - g["grad_l2_step1"]) / (0.15 * g["grad_l2_step1"] + 2e-3 * gmax)).max()))
    np.testing.assert_allclose(l2, g["grad_l2_step1"], rtol=0.15, atol=2e-3 * gmax)
    cos = {}
    for i, k in enumerate(str(s) for s in g["full_grad_names"]):
        a, b = got[k].ravel().astype(np.float64), g["grad_full_%d" % i].ravel().astype(np.float64)
        cos[k] = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
    print("gradient cosines:", {k: round(v, 5) for k, v in cos.items()})
    # measured: head 1.0000, decoder_stage4_conv2 0.9992, decoder_stage2_conv1 0.929, stage4 0.866, stage 1-3 and the stem 0.79-0.81 -
    # the 3.3-ulp logit drift decorrelates the gradient by ~1.5 % per BatchNormalization / ReLU pair going backwards (64 px /
    # ResNet18: 0.967 at worst; against the fp32 oracle the bf16 minimum is ~0.7: test_bf16_step_close_to_fp32_oracle)
    assert cos["final_conv/kernel"] > 0.9995 and cos["decoder_stage4_conv2/kernel"] > 0.995 and min(cos.values()) > 0.7, cos


def test_fp32_pspnet_resnet101_step_matches_oracle():
    """BASELINE.json configs[4]'s encoder (PSPNet over ResNet101: 23 bottleneck units in stage 3, cut at the 1/8 feature) against the
    oracle at 96 px, 20 classes, batch 2 - the parity case of the PSPNet/ResNet101 768 x 768 workload."""
    n, size, classes = 2, 96, 20
    P = onets.init_pspnet_resnet("resnet101", classes=classes, seed=42)
    x, _ = ostep.synthetic_batch(n, size, size, seed=8)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 8 + 3 * (xx // 12)) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = "categorical_crossentropy+1.0*dice_loss"
    tr = ostep.OracleTrainer(P, backbone="resnet101", loss=spec, optimizer="sgd", lr=0.02, architecture="PSPNet", activation="softmax")
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("PSPNet", "resnet101", (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02,
                    use_graph=False)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        # (twice the ReLU layers of ResNet50 to flip a kink in: the resnet50 bar of test_fp32_step_matches_oracle)
        assert e <= (1e-4 if k.startswith("final_conv") else 6e-2), "grad %s: rel L2 %.3g" % (k, e)


def test_fp32_backward_equals_fp64_recomputation_from_own_buffers():
    """Backward parity that does not rest on the 3e-2 whole-step bars (those are set by ReLU-kink flips BETWEEN two fp32 forward
    passes): dW and dX of two 3x3 layers - one in the decoder, one in the encoder - are recomputed in float64 numpy
    (oracle/np_ops.py) from the HIP path's OWN stored activations, stored dY and weights.  Same kinks on both sides, so what is
    left is fp32 accumulation order: <= 1e-6 relative L2.  dX is the masked gradient g = dgrad(dY) * [BN output > 0] that the fused
    BatchNormalization-backward epilogue stores (the mask re-derived from the stored ReLU output)."""
    from oracle import np_ops
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    m = make("resnet18", size, n, "fp32")
    m.set_weights(P)
    m.load_batch(x, y)
    m.forward_backward()
    torch.cuda.synchronize()
    ts = m.plan.tensors
    W, G = m.get_weights(), m.get_gradients()
    for conv, src in (("decoder_stage1_conv2", "decoder_stage1_bn1"), ("stage2_unit2_conv1", "stage2_unit2_bn1")):
        xin = m.activation(src).astype(np.float64)                       # BN + ReLU output as the HIP path computed it
        dy = ts[conv].grad.to(torch.float32).cpu().numpy().astype(np.float64)[..., :ts[conv].C]
        w = W[conv + "/kernel"].astype(np.float64)                       # Keras layout [kh, kw, in, out]
        dw_ref = np_ops.conv2d_wgrad(xin, dy, (3, 3), 1, 1)
        e = rel_l2(G[conv + "/kernel"], dw_ref)
        assert e <= 1e-6, "%s dW: rel L2 %.3g vs the fp64 recomputation" % (conv, e)
        assert ts[src].meta.get("uses") == 1 and ts[src].meta.get("bnb") is not None      # the fused form: dX buffer = masked gradient
        dx_ref = np_ops.conv2d_dgrad(dy, w, (ts[src].H, ts[src].W), 1, 1) * (xin > 0)
        dx = ts[src].grad.to(torch.float32).cpu().numpy()
        e = rel_l2(dx, dx_ref)
        assert e <= 1e-6, "%s dX: rel L2 %.3g vs the fp64 recomputation" % (conv, e)


def test_fp32_vgg16_unet_step_matches_oracle():
    """U-Net over keras.applications VGG16: biased 3x3 convolutions with the ReLU fused into the epilogue
    (gradient through stp_relu_bwd), 2x2 max-pooling, raw-pixel input without normalisation, five skip connections."""
    n, size = 2, 64
    P = onets.init_unet_resnet("vgg16", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone="vgg16", loss=LOSS, optimizer="sgd", lr=1e-3)
    m = make("vgg16", size, n, "fp32", optimizer="SGD", lr=1e-3)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    # forward + backward, taps, THEN the optimizer: a BatchNormalization output that lives only inside its consumers' staging
    # is materialised on demand with the current gamma / beta (graph.Plan.tensor), so it must be read before they move
    m.load_batch(x, y)
    m.forward_backward()
    bad = first_bad_tap(m, taps, 2e-4)
    m.apply_gradients()
    met = m.metrics()
    assert bad is None, bad
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 1e-5 * max(1.0, abs(o["loss"]))
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
    mb = make("vgg16", size, n, "bf16", use_graph=True, optimizer="Adam", lr=1e-4)     # bf16 + hipGraph: runs and learns
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(10):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0


def test_fp32_transpose_decoder_step_matches_oracle():
    """`decoder_block_type: transpose` (Conv2DTranspose 4x4 s2 'same' -> BN -> ReLU -> concat -> conv3x3): the transposed
    convolution runs as the zero-insertion gather with the flipped kernel, its gradients as a stride-2 convolution and a
    zero-insertion weight-gradient; checked against torch's conv_transpose2d in the oracle at the usual bars."""
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=42, decoder_block_type="transpose")
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="sgd", lr=0.05, opt_kwargs={"momentum": 0.9})
    m = make("resnet18", size, n, "fp32", optimizer="SGD", lr=0.05, opt_kwargs={"momentum": 0.9}, decoder_block_type="transpose")
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    for k, v in m.get_weights().items():                       # Keras (kh,kw,out,in) layout survives the round trip
        np.testing.assert_array_equal(v, P[k], err_msg=k)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    # forward + backward, taps, THEN the optimizer: a BatchNormalization output that lives only inside its consumers' staging
    # is materialised on demand with the current gamma / beta (graph.Plan.tensor), so it must be read before they move
    m.load_batch(x, y)
    m.forward_backward()
    bad = first_bad_tap(m, taps, 2e-4)
    m.apply_gradients()
    met = m.metrics()
    assert bad is None, bad
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 1e-5 * max(1.0, abs(o["loss"]))
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
    # bf16 + hipGraph: runs and learns
    mb = make("resnet18", size, n, "bf16", use_graph=True, decoder_block_type="transpose")
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(10):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0


def test_fp32_softmax_head_step_matches_oracle():
    """`classes: 3, activation: softmax, loss: categorical_crossentropy+dice_loss` through the whole
    step (3-channel head conv, channel softmax, class-index masks) at the binary head's bars."""
    n, size, classes = 2, 64, 3
    P = onets.init_unet_resnet("resnet18", classes=classes, seed=42)
    x, _ = ostep.synthetic_batch(n, size, size, seed=77)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 16 + xx // 24) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)      # label image
    spec = "categorical_crossentropy+1.0*dice_loss"
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=spec, optimizer="sgd", lr=0.05, activation="softmax")
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("Unet", "resnet18", (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD",
                    lr=0.05, use_graph=False)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["dice"] - o["dice"]) < 1e-5
    assert abs(met["categorical_crossentropy"] - o["bce"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
    # inference: softmax probabilities, rows sum to one
    m.set_weights(tr.P)
    pr = m.predict(x)
    ref = torch.softmax(torch.from_numpy(tr.forward(x.astype(np.float32))), dim=-1).numpy()
    assert pr.shape == (n, size, size, classes)
    np.testing.assert_allclose(pr.sum(axis=-1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pr, ref, atol=1e-3)


def test_fp32_deeplabv3_mobilenetv2_step_matches_oracle():
    """The reference's IN-TREE model (segmentation_pipeline/impl/deeplab/model.py, `architecture: DeepLabV3`, the example
    experiment examples/people/ds_1.yaml): MobileNetV2 inverted residuals with depthwise / atrous convolutions and ReLU6,
    ASPP image pooling, Dropout with the reproducible device mask, sigmoid inside the model and align-corners upsampling of
    the PROBABILITIES, loss on probabilities.  Two steps (the second draws a different dropout mask)."""
    from oracle import deeplab as odl
    n, size = 2, 64
    P = odl.init_deeplab_mobilenetv2(seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=21)
    tr = ostep.OracleTrainer(P, backbone="mobilenetv2", loss=LOSS, optimizer="sgd", lr=0.02, architecture="DeepLabV3")
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("DeepLabV3", "mobilenetv2", (size, size, 3), 1, "sigmoid", batch=n, dtype="fp32", loss=LOSS, optimizer="SGD", lr=0.02,
                    use_graph=False)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    for k, v in m.get_weights().items():
        np.testing.assert_array_equal(v, P[k], err_msg=k)                      # depthwise kernels keep Keras' (kh,kw,C,1) layout
    for step_no in range(2):
        taps = {}
        o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
        met = m.train_on_batch(x, y)
        if step_no == 0:
            for oname, pname in (("Conv", "Conv_BN"), ("block0", "expanded_conv_project_BN"), ("block2", "expanded_conv_2_add"),
                                 ("block9", "expanded_conv_9_add"), ("block16", "expanded_conv_16_project_BN")):
                ref = taps[oname].detach().numpy()
                np.testing.assert_allclose(m.activation(pname), ref, atol=3e-4 * max(1.0, np.abs(ref).max()), err_msg=pname)
        np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-4)          # probabilities: 1e-3 on logits ~ 2.5e-4 on p
        assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
        g = m.get_gradients()
        for k, ref in o["grads"].items():
            # a projection BN's beta feeds a 1x1 convolution + batch-statistics BN with no non-linearity in between: the loss
            # does not depend on it and both gradients are round-off around zero - hence the absolute floor
            e = np.linalg.norm(g[k].astype(np.float64) - ref) / (np.linalg.norm(ref.astype(np.float64)) + 1e-3)
            assert e <= (1e-4 if k.startswith("custom_logits") else 6e-2), "step %d grad %s: rel L2 %.3g" % (step_no, k, e)
        m.set_weights(tr.P)
    # inference: Dropout off, moving statistics
    pr = m.predict(x)
    ref = tr.forward(x.astype(np.float32))
    assert pr.shape == (n, size, size, 1)
    np.testing.assert_allclose(pr, ref, atol=2e-4)
    # bf16 + hipGraph: runs, draws a fresh mask per replay, learns
    mb = HipSegModel("DeepLabV3", "mobilenetv2", (size, size, 3), 1, "sigmoid", batch=n, dtype="bf16", loss=LOSS, optimizer="Adam", lr=1e-3)
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(15):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0 and int(mb.plan.step_state[0].item()) == 16


def test_fp32_pspnet_step_with_nested_pyramid_windows_matches_oracle():
    """PSPNet at a size whose pyramid windows nest like the 768 x 768 workload's (384 px: 1/8 feature 48 x 48, windows 48 / 24 / 16 / 8):
    the four average poolings run as ONE pass over the feature map and one over its gradient (stp_avgpool_pyramid, round 6), the loss
    reads the low-resolution logits (stp_softmax_cce_dice_up) - whole step against the oracle: level activations, logits, loss, gradients."""
    n, size, classes, backbone = 2, 384, 4, "resnet18"
    P = onets.init_pspnet_resnet(backbone, classes=classes, seed=42)
    x, _ = ostep.synthetic_batch(n, size, size, seed=8)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 48 + xx // 64) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = "categorical_crossentropy+1.0*dice_loss"
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.02, architecture="PSPNet", activation="softmax")
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("PSPNet", backbone, (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02, use_graph=False)
    names = [l[2] for l in m.plan.fwd + m.plan.bwd]
    assert names.count("stp_avgpool_pyramid") == 1 and names.count("stp_avgpool_pyramid_bwd") == 1 and "stp_avgpool" not in names
    assert names.count("stp_softmax_cce_dice_up") == 1
    m.set_weights(P)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    met = m.train_on_batch(x, y)
    for lvl in (1, 2, 3, 6):
        ref = taps["psp_level%d_out" % lvl].detach().numpy()
        np.testing.assert_allclose(m.activation("psp_level%d_bn" % lvl), ref, atol=5e-4 * max(1.0, np.abs(ref).max()), err_msg="level %d" % lvl)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        # (bn0/gamma: the sum over 147 K pixels of a quantity that cancels to ~1e-3 of its terms - 0.046-0.048 with every form of the plan,
        #  the separate pooling launches and the unfused loss included: scratch/r06/psp384_debug.py)
        assert e <= (1e-4 if k.startswith("final_conv") else 1e-1 if k == "bn0/gamma" else 3e-2), "grad %s: rel L2 %.3g" % (k, e)


@pytest.mark.parametrize("backbone,classes", [("resnet18", 1), ("resnet50", 5)])
def test_fp32_pspnet_step_matches_oracle(backbone, classes):
    """PSPNet (BASELINE.json configs[4] family): backbone cut at the 1/8 feature, pyramid pooling levels 1/2/3/6 through
    stp_avgpool -> 1x1 conv + BN + ReLU -> TF1 bilinear resize into the concatenated tensor, x8 resize of the logits.
    96 px = 12x12 feature map (divisible by 6); BatchNormalization at level 1 sees only `batch` values per channel."""
    n, size = 2, 96
    P = onets.init_pspnet_resnet(backbone, classes=classes, seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=8)
    act = "sigmoid" if classes == 1 else "softmax"
    if classes > 1:
        yy, xx = np.mgrid[0:size, 0:size]
        y = ((yy // 16 + xx // 24) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = LOSS if classes == 1 else "categorical_crossentropy+1.0*dice_loss"
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.02, architecture="PSPNet", activation=act)
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("PSPNet", backbone, (size, size, 3), classes, act, batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02,
                    use_graph=False)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    met = m.train_on_batch(x, y)
    for lvl in (1, 2, 3, 6):
        ref = taps["psp_level%d_out" % lvl].detach().numpy()
        np.testing.assert_allclose(m.activation("psp_level%d_bn" % lvl), ref, atol=5e-4 * max(1.0, np.abs(ref).max()), err_msg="level %d" % lvl)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
    mb = HipSegModel("PSPNet", backbone, (size, size, 3), classes, act, batch=n, dtype="bf16", loss=spec, optimizer="Adam", lr=1e-3)
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(10):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0                                   # bf16 + hipGraph: runs and learns


def test_fp32_fpn_resnet50_three_class_matches_oracle():
    """BASELINE.json configs[3] as a parity case (FPN / ResNet50, 3-class softmax), at a size the oracle finishes in seconds."""
    n, size, classes = 1, 128, 3
    P = onets.init_fpn_resnet("resnet50", classes=classes, seed=42)
    x, _ = ostep.synthetic_batch(n, size, size, seed=5)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 32 + xx // 48) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = "categorical_crossentropy+0.5*dice_loss"
    tr = ostep.OracleTrainer(P, backbone="resnet50", loss=spec, optimizer="sgd", lr=0.01, architecture="FPN", activation="softmax")
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("FPN", "resnet50", (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.01,
                    use_graph=False)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=2e-3)          # deep bottleneck encoder, batch of one: see above
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    pr = m.predict(x)
    assert pr.shape == (n, size, size, classes) and np.allclose(pr.sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("opt,lr", [("RMSprop", 1e-3), ("Nadam", 2e-3)])
def test_rmsprop_and_nadam_steps_match_oracle(opt, lr):
    """the remaining Keras optimizers through the whole step (fp32, hipGraph replay: Nadam's momentum
    schedule is device state).  Weights after 2 steps vs the oracle, re-synchronised between steps (ReLU-kink noise)."""
    P = onets.init_unet_resnet("resnet18", seed=3)
    x, y = ostep.synthetic_batch(2, 64, 64, seed=9)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer=opt.lower(), lr=lr)
    m = make("resnet18", 64, 2, "fp32", optimizer=opt, lr=lr, use_graph=True)
    m.set_weights(P)
    for step_no in range(2):
        o = tr.step(x.astype(np.float32), y.astype(np.float32))
        met = m.train_on_batch(x, y)
        assert abs(met["loss"] - o["loss"]) < 2e-5 * max(1.0, abs(o["loss"]))
        w = m.get_weights()
        # both optimizers normalise by sqrt(v): the first update is ~ lr*sign(g) (x 3.16 for RMSprop's rho = 0.9), so an
        # element whose gradient is at rounding level may move the other way; compare the bulk (99 %) tightly and bound
        # the rest by two full-size steps
        for k in tr.P:
            d = np.abs(w[k] - tr.P[k]).ravel()
            bad = int((d > 2e-5 + 0.05 * lr).sum())
            assert bad <= max(3, 0.02 * d.size) and d.max() <= 7.0 * lr + 1e-5, (k, bad, d.size, d.max())
        m.set_weights(tr.P)


@pytest.mark.parametrize("arch", ["Unet", "Linknet"])
def test_hipgraph_replay_equals_eager(arch):
    P = (onets.init_unet_resnet if arch == "Unet" else onets.init_linknet_resnet)("resnet18", seed=7)
    x, y = ostep.synthetic_batch(2, 64, 64, seed=5)
    outs = []
    for use_graph in (False, True):
        m = make("resnet18", 64, 2, "bf16", use_graph=use_graph, architecture=arch)
        m.set_weights(P)
        r = [m.train_on_batch(x, y) for _ in range(3)]
        outs.append((r, m.logits(), m.get_weights()))
    assert outs[0][0] == outs[1][0]                        # bitwise: kernels are deterministic
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for k in outs[0][2]:
        np.testing.assert_array_equal(outs[0][2][k], outs[1][2][k], err_msg=k)


def test_fixed_point_slot_sums_step_equals_finalize_step(monkeypatch):
    """STP_BN_SLOTS=1 (BatchNormalization sums accumulated in the conv epilogues' int64 slots, no finalize launches) is an opt-in
    schedule of the same arithmetic: fp32 steps agree with the default schedule to summation-order noise, and replay is bitwise."""
    P = onets.init_unet_resnet("resnet18", seed=7)
    x, y = ostep.synthetic_batch(2, 64, 64, seed=5)
    res = {}
    for slots, use_graph in ((0, False), (1, False), (1, True)):
        monkeypatch.setenv("STP_BN_SLOTS", str(slots))
        m = make("resnet18", 64, 2, "fp32", use_graph=use_graph)
        names = [l[2] for l in m.plan.fwd + m.plan.bwd]
        assert ("stp_bn_apply_slots" in names) == bool(slots) and ("stp_bn_backward_slots" in names) == bool(slots)
        m.set_weights(P)
        r1 = m.train_on_batch(x, y)
        first, w1 = m.logits(), m.get_weights()              # forward of step 1: same weights in every schedule
        r2 = m.train_on_batch(x, y)
        res[(slots, use_graph)] = ((r1, r2), m.logits(), m.get_weights(), first, w1)
    a, b, c = res[(0, False)], res[(1, False)], res[(1, True)]
    assert b[0] == c[0]
    np.testing.assert_array_equal(b[1], c[1])
    for k in b[2]:
        np.testing.assert_array_equal(b[2][k], c[2][k], err_msg=k)
    np.testing.assert_allclose(b[3], a[3], atol=1e-4 * max(1.0, np.abs(a[3]).max()))
    for k in ("loss", "dice"):
        assert abs(b[0][0][k] - a[0][0][k]) < 1e-5, k
    # the first Adam step moves every weight by lr * sign(g): only weights whose gradient is rounding noise may differ
    diff = sum(int((np.abs(b[4][k] - a[4][k]) > 1e-4).sum()) for k in a[4])
    assert diff < 0.02 * sum(v.size for v in a[4].values()), diff


def test_fixed_point_slot_sums_in_bf16_with_the_two_destination_data_gradient(monkeypatch):
    """STP_BN_SLOTS=1 in 16-bit storage on a U-Net/ResNet34 whose decoder_stage3_conv1 data gradient is the two-destination
    small-channel kernel (stp_conv2d_scw: 32 -> 64 summed 2x2 + 64 skip channels): that launch keeps FLOAT partial sums (the kernel
    has no slot form; with slots it used to fall through to the generic kernel, which refuses dst_sum2x2 - advisor finding, round 3);
    every other BatchNormalization runs on the slots.  The step runs, trains, and agrees with the default schedule."""
    P = onets.init_unet_resnet("resnet34", seed=7)
    x, y = ostep.synthetic_batch(2, 64, 64, seed=5)
    res = {}
    for slots in (0, 1):
        monkeypatch.setenv("STP_BN_SLOTS", str(slots))
        m = make("resnet34", 64, 2, "bf16")
        if slots:
            tiles = [(l[3] or {}).get("tile") for l in m.plan.bwd if l[2] == "stp_conv2d" and (l[3] or {}).get("layer") == "decoder_stage3_conv1"]
            assert tiles == [640], tiles                       # the two-destination kernel really takes the launch
            names = [l[2] for l in m.plan.bwd]
            assert "stp_bn_backward_slots" in names and "stp_bn_backward_fused" in names
        m.set_weights(P)
        r = [m.train_on_batch(x, y)["loss"] for _ in range(4)]
        res[slots] = (r, m.logits())
    assert np.isfinite(res[1][0]).all() and res[1][0][-1] < res[1][0][0]
    assert abs(res[1][0][0] - res[0][0][0]) < 2e-3, (res[0][0], res[1][0])     # same forward arithmetic up to summation order of the sums
    assert np.corrcoef(res[0][1].ravel(), res[1][1].ravel())[0, 1] > 0.8      # (four Adam steps apart in bf16: measured 0.90)


@pytest.mark.parametrize("use_graph", [False, True])
def test_overlapped_rccl_allreduce_equals_plain_step(use_graph):
    """Data-parallel path on one GPU (world size 1 over RCCL, GradReducer(force=True)): the backward cut into
    per-bucket segments with asynchronous all-reduces must leave exactly the weights of the plain step."""
    import socket
    import torch.distributed as dist
    from segmentation_training_pipeline_amd import distributed
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    x, y = ostep.synthetic_batch(2, 64, 64, seed=11)
    ref = make("resnet18", 64, 2, "bf16", use_graph=use_graph)
    ref.init_weights(seed=5)
    for _ in range(3):
        ref.train_on_batch(x, y)
    want = ref.get_weights()
    distributed.init("nccl", force=True)
    try:
        m = make("resnet18", 64, 2, "bf16", use_graph=use_graph)
        m.init_weights(seed=5)
        if use_graph:
            m.set_data_parallel(distributed.GradReducer(bucket_mb=8.0, force=True), overlap="buckets")   # 14.3M params -> 7 buckets
            segs = m._dp_segments()
            assert segs is not None and len(segs) >= 3 and sum(len(r) for _, r in segs) == 7
        else:
            m.set_data_parallel(distributed.GradReducer(force=True))                   # default: two phases
            segs = m._dp_segments()
            assert len(segs) == 2 and segs[0][1][0][1] == m.plan.G.numel() and segs[1][1][0][0] == 0
        for _ in range(3):
            m.train_on_batch(x, y)
        got = m.get_weights()
    finally:
        dist.destroy_process_group()
    for k in want:
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)


def test_bf16_step_close_to_fp32_oracle():
    """bf16 storage / bf16 MFMA inputs / fp32 accumulation, the benchmarked precision.  Every layer
    rounds inputs, weights and outputs to 8 significant bits (2^-9 relative RMS each), which over the
    ~40 conv+BN layers of the path compounds to ~2 % of the logit range (measured 1.8 % mean).  The
    per-kernel bf16 tests (test_ops_gpu.py) hold each kernel to one output rounding against the oracle
    evaluated on the same bf16-rounded inputs; this test bounds the end-to-end drift."""
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="adam", lr=1e-3)
    m = make("resnet18", size, n, "bf16")
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    ref = o["logits"]
    err = np.abs(m.logits() - ref)
    rng_ = np.abs(ref).max()
    assert err.mean() < 0.03 * rng_ and err.max() < 0.25 * rng_, (err.max(), err.mean(), rng_)
    assert np.corrcoef(m.logits().ravel(), ref.ravel())[0, 1] > 0.98
    assert abs(met["loss"] - o["loss"]) < 2e-2
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-2
    g = m.get_gradients()
    cos = []
    for k, r in o["grads"].items():
        if r.size > 64:
            a, b = g[k].ravel().astype(np.float64), r.ravel().astype(np.float64)
            cos.append(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    # bf16 drift of the forward (the net is evaluated at slightly perturbed activations) decorrelates
    # the gradient by ~1.5 % per BN/ReLU pair going backwards: 0.996 at the head, ~0.8 at the stem
    # (fp32 mode: 0.99999 everywhere - scratch/dbg_bf16.py).
    assert min(cos) > 0.65 and max(cos) > 0.99, (min(cos), max(cos))
    # training makes progress in bf16
    losses = [met["loss"]] + [m.train_on_batch(x, y)["loss"] for _ in range(8)]
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_16bit_step_matches_the_storage_quantised_oracle(dtype, monkeypatch):
    """The 16-bit training modes against the oracle that rounds WHERE THE KERNELS ROUND (oracle.nets._Ctx(storage=...): stored
    activations, stored gradients, weight compute copies; fp32 statistics / accumulation / loss): what is left between the two is
    fp32 summation order and the rare rounding tie that falls the other way - not the ~2 % drift of 8-bit storage itself, which
    both sides now share.  Bars in units of one storage ulp at the logit range."""
    # (the decoder's class-collapsed weight copies - sums of taps rounded once more - are a rounding point the oracle does not have)
    monkeypatch.setenv("STP_UPCOLLAPSE", "0")
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="adam", lr=1e-3, storage=dtype)
    o = tr.step(x.astype(np.float32), y.astype(np.float32), apply=False)
    m = make("resnet18", size, n, dtype)
    m.set_weights(P)
    met = m.train_on_batch(x, y)
    ref, got = o["logits"], m.logits()
    rng_ = float(np.abs(ref).max())
    ulp = 2.0 ** (np.floor(np.log2(rng_)) - (7 if dtype == "bf16" else 10))      # spacing of the format at the top of the logit range
    err = np.abs(got - ref)
    print("storage-quantised oracle [%s]: logit range %.3f ulp %.4g  max err %.4g (%.2f ulp)  mean err %.4g (%.3f ulp)  exact %.3f"
          % (dtype, rng_, ulp, err.max(), err.max() / ulp, err.mean(), err.mean() / ulp, float((err == 0).mean())))
    # measured (MI355X): bf16 max 4.4 / mean 0.60 ulp (vs the fp32 oracle: mean 2.9 ulp = 1.8 % of the range), fp16 max 11 / mean 1.6 ulp
    # (0.006 absolute: 3x below bf16's - the remaining differences are rounding ties that cascade through ~40 layers, not storage)
    bar_max, bar_mean = (8.0, 1.0) if dtype == "bf16" else (24.0, 3.0)
    assert err.max() <= bar_max * ulp and err.mean() <= bar_mean * ulp, (err.max() / ulp, err.mean() / ulp)
    o32 = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="adam", lr=1e-3).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    if dtype == "bf16":        # the device is at least 3x closer to the oracle that shares its rounding points than to the fp32 one
        assert err.mean() * 3.0 < np.abs(got - o32["logits"]).mean()
    assert abs(met["loss"] - o["loss"]) < 5e-3 and abs(met["dice_loss"] - o["dice_loss"]) < 2e-3
    g = m.get_gradients()
    cos = {}
    for k, r in o["grads"].items():
        if r.size > 64:
            a, b = g[k].ravel().astype(np.float64), r.ravel().astype(np.float64)
            cos[k] = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
    worst = min(cos, key=cos.get)
    print("gradient cosine: min %.5f (%s), head %.6f" % (cos[worst], worst, cos["final_conv/kernel"]))
    # measured: min 0.967 (bf16) / 0.980 (fp16) at a stage-2 BatchNormalization beta, head 0.99995 - against the fp32 oracle the bf16
    # minimum is ~0.7 (test_bf16_step_close_to_fp32_oracle): most of that decorrelation was the storage format, which both sides now share
    assert cos[worst] > 0.95 and cos["final_conv/kernel"] > 0.9995, (worst, cos[worst])


# (architecture, backbone, size, classes, storage) -> bars: logits max / mean error in storage ulps of the logit range, how many times
# closer to the storage-quantised oracle than to the fp32 one, |loss difference|, worst gradient cosine, class-convolution cosine.
# Measured on MI355X (profiles/r06a_16bit_parity_other_graphs.txt), device vs oracle | oracle (fp32 accumulation) vs oracle (fp64
# accumulation: the SAME rounding points evaluated a second, equally valid way = the noise floor of the 16-bit step):
#   PSPNet/ResNet101 bf16 : mean 1.13 ulp, max 6.8, cosine min 0.910 / median 0.953   |  1.09, 7.5, 0.946 / 0.962
#   FPN/ResNet50 fp16     : mean 7.73 ulp, max 65,  cosine min 0.853 / median 0.916   |  7.68, 50,  0.838 / 0.896
#   Linknet/ResNet34 bf16 : mean 4.20 ulp, max 44,  cosine min 0.339 / median 0.611   |  3.94, 43,  0.511 / 0.654
# i.e. the device sits AT the noise floor in all three; the absolute bars are ~1.6x the measurement, and the test also holds the device
# to 1.5x the floor it measures itself.  (Linknet's gradients decorrelate that fast in ANY bf16 evaluation: three BatchNormalizations
# per decoder stage, the first over 32 values per channel.)
STORAGE_CASE_BARS = {
    ("FPN", "resnet50", 128, 3, "fp16"): dict(max_ulp=110.0, mean_ulp=12.0, closer=1.5, loss=5e-3, cos_min=0.75, cos_head=0.999),
    ("PSPNet", "resnet101", 96, 20, "bf16"): dict(max_ulp=12.0, mean_ulp=2.0, closer=1.8, loss=5e-3, cos_min=0.85, cos_head=0.999),
    ("Linknet", "resnet34", 128, 1, "bf16"): dict(max_ulp=80.0, mean_ulp=7.0, closer=1.3, loss=5e-3, cos_min=0.2, cos_head=0.9995),
}


@pytest.mark.parametrize("case", sorted(STORAGE_CASE_BARS), ids=lambda c: "%s-%s-%d-%dcls-%s" % c)
def test_16bit_step_of_the_other_graphs_matches_the_storage_quantised_oracle(case, golden_dir):
    """BASELINE.json configs[3] (FPN / ResNet50, fp16, 3 classes) and configs[4] (PSPNet / ResNet101, bf16, 20 classes) - and Linknet /
    ResNet34 - AT THEIR BENCHMARKED PRECISION against the oracle that rounds where their kernels round
    (oracle.nets._Ctx(storage=...): stored convolution / BatchNormalization outputs, the top-down sums, pooled maps, resized slices, the
    tap channels of the class head, the resized logits, every stored gradient - at the fp16 build's loss scale - and the weight compute
    copies).  The 16-bit builds take graph paths the fp32 mode does not (tap-channel class head with padded gradient rows, lean /
    two-source kernels, the scatter form of the 1x1 / stride-2 shortcuts): this is their whole-step parity case.  The oracle runs LIVE
    here AND the device is held to the committed fixture of the same step (tests/golden/make_golden.py --storage-cases-only)."""
    arch, backbone, size, classes, dtype = case
    bars = STORAGE_CASE_BARS[case]
    g = np.load(os.path.join(golden_dir, "%s_%s_%d_%s.npz" % (arch.lower(), backbone, size, dtype)))
    x, y = g["x"], g["y"]
    gs = float(g["grad_scale"])
    act, spec = ("sigmoid", LOSS) if classes == 1 else ("softmax", "categorical_crossentropy+1.0*dice_loss")
    init = {"Linknet": onets.init_linknet_resnet, "FPN": onets.init_fpn_resnet, "PSPNet": onets.init_pspnet_resnet}[arch]
    P = init(backbone, classes=classes, seed=int(g["seed"]))
    kw = dict(backbone=backbone, loss=spec, optimizer="adam", lr=1e-3, architecture=arch, activation=act)
    o = ostep.OracleTrainer(P, storage=dtype, grad_scale=gs, **kw).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    # the noise floor: the same rounding points with float64 accumulation inside the convolutions
    o64 = ostep.OracleTrainer(P, storage=dtype, grad_scale=gs, accum64=True, **kw).step(x.astype(np.float32), y.astype(np.float32), apply=False)
    # the live oracle against its committed fixture: the same arithmetic on another host CPU (oneDNN picks its kernels by ISA), so fp32
    # summation order may decide a few rounding ties differently - bit-identical on the build container, held to the device's bars elsewhere
    drift = np.abs(o["logits"] - g["logits1"])
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel(arch, backbone, (size, size, 3), classes, act, batch=x.shape[0], dtype=dtype, loss=spec, optimizer="Adam", lr=1e-3, use_graph=False)
    assert m.loss_scale == gs
    m.set_weights(P)
    m.load_batch(x, y)
    m.forward_backward()
    met = m.metrics()
    ref, got = o["logits"], m.logits()
    rng_ = float(np.abs(ref).max())
    ulp = 2.0 ** (np.floor(np.log2(rng_)) - (7 if dtype == "bf16" else 10))
    err = np.abs(got - ref)
    err32 = np.abs(got - g["logits1_fp32"].astype(np.float32))
    print("storage-quantised oracle %s: logit range %.3f ulp %.4g  max err %.4g (%.2f ulp)  mean err %.4g (%.3f ulp)  exact %.3f;  vs the fp32 oracle mean %.3f ulp (%.2fx)"
          % (case, rng_, ulp, err.max(), err.max() / ulp, err.mean(), err.mean() / ulp, float((err == 0).mean()), err32.mean() / ulp,
             err32.mean() / max(err.mean(), 1e-30)))
    print("live oracle vs its committed fixture: max %.2f ulp, mean %.3f ulp, exact %.4f" % (drift.max() / ulp, drift.mean() / ulp, float((drift == 0).mean())))
    errg = np.abs(got - g["logits1"])
    print("device vs the committed fixture: max %.2f ulp, mean %.3f ulp" % (errg.max() / ulp, errg.mean() / ulp))
    print("loss %.5f (oracle %.5f)  dice_loss %.5f (%.5f)" % (met["loss"], o["loss"], met["dice_loss"], o["dice_loss"]))
    floor = np.abs(o64["logits"] - ref)
    got_g = m.get_gradients()

    def cosines(ga):
        out = {}
        for k, r in o["grads"].items():
            if r.size > 64:
                a, b = ga[k].ravel().astype(np.float64), r.ravel().astype(np.float64)
                out[k] = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
        return out
    cos, cos_floor = cosines(got_g), cosines(o64["grads"])
    worst = min(cos, key=cos.get)
    srt, srt_f = sorted(cos.values()), sorted(cos_floor.values())
    print("gradient cosine: min %.5f (%s), 5th percentile %.5f, median %.5f, head %.6f" % (cos[worst], worst, srt[len(srt) // 20], srt[len(srt) // 2],
                                                                                          cos["final_conv/kernel"]))
    print("NOISE FLOOR (oracle with fp64 accumulation vs the oracle): logits max %.2f ulp, mean %.3f ulp; gradient cosine min %.5f, 5th percentile "
          "%.5f, median %.5f" % (floor.max() / ulp, floor.mean() / ulp, srt_f[0], srt_f[len(srt_f) // 20], srt_f[len(srt_f) // 2]))
    # the device against the floor it is compared with: within 1.5x of the distance between two CPU evaluations of the same rounding points
    assert err.mean() <= 1.5 * floor.mean() + 0.25 * ulp, (err.mean() / ulp, floor.mean() / ulp)
    assert 1.0 - srt[len(srt) // 2] <= 1.5 * (1.0 - srt_f[len(srt_f) // 2]) + 0.01, (srt[len(srt) // 2], srt_f[len(srt_f) // 2])
    assert err.max() <= bars["max_ulp"] * ulp and err.mean() <= bars["mean_ulp"] * ulp, (err.max() / ulp, err.mean() / ulp)
    assert errg.max() <= bars["max_ulp"] * ulp and errg.mean() <= bars["mean_ulp"] * ulp, (errg.max() / ulp, errg.mean() / ulp)
    assert drift.mean() <= bars["mean_ulp"] * ulp, drift.mean() / ulp
    assert err.mean() * bars["closer"] <= err32.mean(), (err.mean() / ulp, err32.mean() / ulp)
    assert abs(met["loss"] - o["loss"]) < bars["loss"] and abs(met["dice_loss"] - o["dice_loss"]) < bars["loss"]
    assert cos[worst] > bars["cos_min"] and cos["final_conv/kernel"] > bars["cos_head"], (worst, cos[worst], cos["final_conv/kernel"])
    # the committed gradients (a subset in full) against the device: the fixture pins the backward as well
    cg = {}
    for i, k in enumerate(str(s_) for s_ in g["full_grad_names"]):
        a, b = got_g[k].ravel().astype(np.float64), g["grad_full_%d" % i].ravel().astype(np.float64)
        cg[k] = a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)
    print("gradient cosines vs the committed fixture:", {k: round(v, 5) for k, v in cg.items()})
    assert min(cg.values()) > bars["cos_min"] and cg["final_conv/kernel"] > bars["cos_head"], cg


def test_fp16_step_close_to_fp32_oracle():
    """IEEE-half storage / v_mfma_*_f16 / fp32 accumulation (BASELINE.json configs[3] "fp16 MFMA"): libstp_hip_f16.so, the same
    sources as the bf16 build with the storage-format helpers switched.  11 significant bits instead of 8, so the end-to-end
    drift is ~8x smaller than the bf16 mode's; the backward runs under the static loss scale 2^14 (the 1/(N*H*W) BCE gradient
    would otherwise fall below fp16's normal range), divided out by the optimizer's device scalar.  Also: the scale must not
    change the update (loss_scale 2^14 vs 2^6: same weights after a step up to fp16 rounding of the gradients)."""
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="adam", lr=1e-3)
    m = make("resnet18", size, n, "fp16")
    assert m.loss_scale == 16384.0 and m.plan.lib.stp_storage_dtype() == 3
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    ref = o["logits"]
    err = np.abs(m.logits() - ref)
    rng_ = np.abs(ref).max()
    assert err.mean() < 0.005 * rng_ and err.max() < 0.05 * rng_, (err.max(), err.mean(), rng_)
    assert np.corrcoef(m.logits().ravel(), ref.ravel())[0, 1] > 0.999
    assert abs(met["loss"] - o["loss"]) < 3e-3
    assert abs(met["dice_loss"] - o["dice_loss"]) < 2e-3
    g = m.get_gradients()
    cos = []
    for k, r in o["grads"].items():
        if r.size > 64:
            a, b = g[k].ravel().astype(np.float64), r.ravel().astype(np.float64)
            assert np.isfinite(a).all(), k
            cos.append(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30))
    assert min(cos) > 0.9 and max(cos) > 0.999, (min(cos), max(cos))
    w1 = m.get_weights()
    m2 = make("resnet18", size, n, "fp16", loss_scale=64.0)
    m2.set_weights(P)
    m2.train_on_batch(x, y)
    g2 = m2.get_gradients()
    for k in ("decoder_stage4_conv2/kernel", "stage1_unit1_conv1/kernel", "conv0/kernel"):
        if k in g:
            assert rel_l2(g2[k], g[k]) < 2e-2, (k, rel_l2(g2[k], g[k]))
    # training makes progress in fp16 (eager and hipGraph replay agree bit for bit)
    losses = [met["loss"]] + [m.train_on_batch(x, y)["loss"] for _ in range(8)]
    assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0]
    mg = make("resnet18", size, n, "fp16", use_graph=True)
    mg.set_weights(P)
    lg = [mg.train_on_batch(x, y)["loss"] for _ in range(9)]
    assert lg == losses, (lg, losses)
    with pytest.raises(Exception):
        make("resnet18", size, n, "fp16", loss="binary_crossentropy+0.5*lovasz_loss", loss_scale=128.0)


def test_fp16_non_finite_gradients_skip_the_step_instead_of_poisoning_the_state():
    """fp16 storage, loss scaling on: the overflow guard (stp_grad_global_scale in front of the optimizer) turns a step whose gradient
    arena holds a non-finite value into a no-op - weights, Adam's moments and the step counter untouched, skipped_steps counts it.
    The fp16 build saturates its stores and its min / max clamps swallow NaNs, so a non-finite gradient cannot be provoked through
    the inputs: one is written into the arena between the backward and the optimizer."""
    x, y = ostep.synthetic_batch(2, 64, 64, seed=5)
    m = make("resnet18", 64, 2, "fp16")
    m.init_weights(seed=9)
    w0 = m.get_weights()
    m.load_batch(x, y)
    m.forward_backward()
    assert np.isfinite(m.plan.G.cpu().numpy()).all()
    m.plan.G[12345] = float("inf")
    m.apply_gradients()
    assert m.skipped_steps == 1 and int(m.opt_state[0].item()) == 0
    w1 = m.get_weights()
    for k in m.plan.params:
        assert np.array_equal(w0[k], w1[k]), k
    assert float(m.m.abs().max().item()) == 0.0 and float(m.v.abs().max().item()) == 0.0
    met = m.train_on_batch(x, y)                      # the next step is a normal one
    assert np.isfinite(met["loss"]) and m.skipped_steps == 1 and int(m.opt_state[0].item()) == 1
    assert any(not np.array_equal(w0[k], v) for k, v in m.get_weights().items() if k in m.plan.params)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fp16_dynamic_loss_scale_halves_on_overflow_and_grows_back(monkeypatch, use_graph):
    """Dynamic loss re-scaling on top of the overflow guard (device record ``dls``, stp_scale_by_device + stp_grad_global_scale_dls), all
    inside the captured step: a skipped step halves the multiplier, ``interval`` clean steps double it, the optimizer sees the same
    gradient whatever the multiplier (weights after a step at multiplier 1/2 equal those of the static-scale run to fp16 rounding),
    and get_gradients() divides the multiplier out."""
    monkeypatch.setenv("STP_LOSS_SCALE_INTERVAL", "3")
    x, y = ostep.synthetic_batch(2, 64, 64, seed=5)
    m = make("resnet18", 64, 2, "fp16", use_graph=use_graph)
    assert m.dls is not None and m.dynamic_loss_scale == 16384.0
    names = [l[2] for l in m.plan.fwd + m.plan.opt]
    assert "stp_scale_by_device" in names and "stp_grad_global_scale_dls" in names
    m.init_weights(seed=9)
    m.load_batch(x, y)
    m.forward_backward()
    g1 = m.get_gradients()
    m.plan.G[777] = float("nan")                       # provoke the guard (the fp16 build saturates: no overflow through the inputs)
    m.apply_gradients()
    assert m.skipped_steps == 1 and m.dynamic_loss_scale == 8192.0
    m.forward_backward()                              # same weights, half the scale: the same gradients after the division
    g2 = m.get_gradients()
    for k in ("final_conv/kernel", "decoder_stage2_conv1/kernel", "stage1_unit1_conv1/kernel"):
        assert rel_l2(g2[k], g1[k]) < 2e-2, (k, rel_l2(g2[k], g1[k]))
    assert abs(np.abs(m.plan.G.cpu().numpy()).max() / np.abs(g2_max(g2)) - 8192.0) / 8192.0 < 1e-3
    m.apply_gradients()
    losses = [m.train_on_batch(x, y)["loss"] for _ in range(2)]          # clean steps 2 and 3 -> the multiplier doubles back
    assert m.dynamic_loss_scale == 16384.0 and m.skipped_steps == 1 and int(m.opt_state[0].item()) == 3
    for _ in range(3):
        losses.append(m.train_on_batch(x, y)["loss"])
    # the multiplier is capped at 1 (the static scale): the saturating fp16 stores would hide an overflow from the guard above it
    assert m.dynamic_loss_scale == 16384.0 and np.isfinite(losses).all() and losses[-1] < losses[0]
    # the optimizer divides by the multiplier the gradients were PRODUCED under (dls[4]), not the next pass's (dls[0])
    m.forward_backward()
    m.dls[0] = 0.25
    w0 = m.plan.P.clone()
    m.apply_gradients()
    assert torch.isfinite(m.plan.P).all() and float((m.plan.P - w0).abs().max()) < 5e-3      # an Adam step of lr 1e-3, not one of 4x the gradient
    # a growing schedule stays available for experiments
    monkeypatch.setenv("STP_LOSS_SCALE_MAX_MULT", "4")
    mg = make("resnet18", 64, 2, "fp16", use_graph=use_graph)
    mg.init_weights(seed=9)
    for _ in range(3):
        mg.train_on_batch(x, y)
    assert mg.dynamic_loss_scale == 32768.0
    monkeypatch.delenv("STP_LOSS_SCALE_MAX_MULT")
    # the static-scale schedule of round 3 is still there
    monkeypatch.setenv("STP_DYNAMIC_LOSS_SCALE", "0")
    ms = make("resnet18", 64, 2, "fp16", use_graph=use_graph)
    assert ms.dls is None and "stp_grad_global_scale" in [l[2] for l in ms.plan.opt]


def g2_max(g):
    return max(float(np.abs(v).max()) for v in g.values())


@pytest.mark.parametrize("spec", [LOSS, "focal_loss+dice_loss", "binary_crossentropy+0.5*iou_loss+0.02*jaccard_loss",
                                  "binary_crossentropy+0.5*lovasz_loss"])
def test_short_validation_batch_reruns_every_loss_family(spec):
    """Plan.rerun_loss (a validation batch padded by wrapping around is re-evaluated over its real samples only) for every loss
    launch the plan can hold - stp_sigmoid_bce_dice, stp_sigmoid_loss_ex, and stp_lovasz_hinge on top of either: the scalars of the
    first two samples of a padded batch of four equal those of a batch-2 model evaluating those two samples."""
    x, y = ostep.synthetic_batch(4, 64, 64, seed=21)
    x[2:], y[2:] = x[:2], y[:2]                      # the wrapped tail
    m4 = make("resnet18", 64, 4, "fp32", loss=spec)
    m2 = make("resnet18", 64, 2, "fp32", loss=spec)
    m4.init_weights(seed=3)
    m2.set_weights(m4.get_weights())
    out = []
    for m, xs, ys, n_real in ((m4, x, y, 2), (m2, x[:2], y[:2], 2)):
        ep = m.eval_plan()
        ep.inputs["image"].buf.copy_(torch.from_numpy(xs).reshape(ep.inputs["image"].buf.shape))
        ep.inputs["mask"].buf.copy_(torch.from_numpy(ys).to(torch.uint8).reshape(ep.inputs["mask"].buf.shape))
        ep.run(ep.prep); ep.run(ep.fwd)
        if n_real < m.batch:
            ep.rerun_loss(n_real)
        out.append(ep.loss_scalars.cpu().numpy().copy())
    np.testing.assert_allclose(out[0][:13], out[1][:13], rtol=2e-5, atol=2e-6)


def test_freeze_encoder_and_predict_and_checkpoint(tmp_path):
    n, size = 2, 64
    P = onets.init_unet_resnet("resnet18", seed=3)
    x, y = ostep.synthetic_batch(n, size, size, seed=9)
    m = make("resnet18", size, n, "fp32", freeze_encoder=True)
    m.set_weights(P)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="adam", lr=1e-3, freeze_encoder=True)
    for _ in range(2):
        m.train_on_batch(x, y)
        tr.step(x.astype(np.float32), y.astype(np.float32))
    w = m.get_weights()
    for k in P:
        if k.startswith("decoder_") or k.startswith("final_") or "moving" in k:
            continue
        np.testing.assert_array_equal(w[k], P[k], err_msg=k)       # frozen encoder parameters do not move
    assert np.abs(w["decoder_stage0_conv1/kernel"] - P["decoder_stage0_conv1/kernel"]).max() > 0
    # inference phase (moving statistics) vs the oracle's, on 3 images (chunking + padding of the last chunk)
    x3 = np.concatenate([x, x[:1]], axis=0)
    tr.P.update({k: w[k] for k in w})                              # same weights on both sides
    ref = torch.sigmoid(torch.from_numpy(tr.forward(x3.astype(np.float32), training=False))).numpy()
    np.testing.assert_allclose(m.predict(x3), ref, atol=2e-4)
    # checkpoint round trip
    path = str(tmp_path / "best-0.0.weights")
    m.save_weights(path)
    m2 = make("resnet18", size, n, "fp32")
    m2.load_weights(path)
    w2 = m2.get_weights()
    for k in w:
        np.testing.assert_array_equal(w[k], w2[k], err_msg=k)


@pytest.mark.parametrize("backbone,in_ch", [("resnet18", 5), ("vgg16", 4), ("resnet18", 1)])
def test_n_channel_inputs_match_the_oracle(backbone, in_ch):
    """``shape: [H, W, C]`` with C != 3 (reference segmentation.py:135-155 builds N-channel models; without encoder_weights the model
    is simply built on C channels): the raw image is padded to 4 / 8 channels behind the input BatchNormalization (ResNet) or
    cast (VGG), the stem convolution's master kernel keeps its real (k, k, C, 64) shape.  fp32 step against the oracle at the
    north-star bars; the bf16 mode trains."""
    from segmentation_training_pipeline_amd.backend import HipSegModel
    n, size = 2, 64
    P = onets.init_unet_resnet(backbone, in_ch=in_ch, seed=42)
    rng = np.random.RandomState(5)
    x = rng.randint(0, 256, size=(n, size, size, in_ch)).astype(np.uint8)
    _, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=LOSS, optimizer="sgd", lr=0.05, opt_kwargs={"momentum": 0.9})
    mk = lambda dt, g: HipSegModel("Unet", backbone, (size, size, in_ch), 1, "sigmoid", batch=n, dtype=dt, loss=LOSS, optimizer="SGD", lr=0.05,
                                   opt_kwargs={"momentum": 0.9}, use_graph=g)
    m = mk("fp32", False)
    assert sorted(m.get_weights()) == sorted(P)
    k0 = "conv0/kernel" if backbone.startswith("resnet") else "block1_conv1/kernel"
    assert m.get_weights()[k0].shape[2] == in_ch
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5
    g = m.get_gradients()
    assert rel_l2(g["final_conv/kernel"], o["grads"]["final_conv/kernel"]) < 1e-4
    assert rel_l2(g[k0], o["grads"][k0]) < 5e-2                                 # the stem's weight gradient, all C channels
    if backbone.startswith("resnet"):
        assert rel_l2(g["bn_data/beta"], o["grads"]["bn_data/beta"]) < 5e-2   # through the constant-1 channel of the padded input
    mb = mk("bf16", True)
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(5):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0, (l0, l1)


@pytest.mark.parametrize("arch,kw", [("PSPNet", {"downsample_factor": 4, "psp_conv_filters": 128}), ("PSPNet", {"downsample_factor": 16}),
                                     ("FPN", {"pyramid_block_filters": 128, "segmentation_block_filters": 64}),
                                     ("FPN", {"pyramid_block_filters": 256, "segmentation_block_filters": 128, "dropout": 0.3}),
                                     ("PSPNet", {"downsample_factor": 8, "dropout": 0.25}),
                                     ("FPN", {"pyramid_block_filters": 256, "segmentation_block_filters": 128, "interpolation": "nearest"}),
                                     ("PSPNet", {"downsample_factor": 8, "final_interpolation": "nearest"}),
                                     ("PSPNet", {"downsample_factor": 8, "psp_pooling_type": "max"})])
def test_non_default_decoder_options_match_the_oracle(arch, kw):
    """schemas/segmentation.raml:179-249: PSPNet ``downsample_factor`` 4 / 16 (feature = stage2 / stage4 unit1_relu1, final resize
    x4 / x16) and ``psp_conv_filters``; FPN ``pyramid_block_filters`` / ``segmentation_block_filters``.  fp32 step at the north-star
    bars, through the YAML-facing constructors (models.PSPNet / models.FPN keyword surface)."""
    from segmentation_training_pipeline_amd import models
    n = 2
    if arch == "PSPNet":
        f = kw["downsample_factor"]
        size = 6 * f * (2 if f == 4 else 1)                                   # feature map 12x12 (1/4) or 6x6 (1/16)
        P = onets.init_pspnet_resnet("resnet18", seed=42, conv_filters=kw.get("psp_conv_filters", 512), downsample_factor=f)
        okw = {"downsample_factor": f, "dropout": kw.get("dropout"), "final_interpolation": kw.get("final_interpolation", "bilinear"),
               "psp_pooling_type": kw.get("psp_pooling_type", "avg")}
        ctor = models.PSPNet
    else:
        size = 64
        P = onets.init_fpn_resnet("resnet18", seed=42, pyramid_filters=kw["pyramid_block_filters"], segmentation_filters=kw["segmentation_block_filters"])
        okw = {"dropout": kw.get("dropout"), "interpolation": kw.get("interpolation", "bilinear")}
        ctor = models.FPN
    x, y = ostep.synthetic_batch(n, size, size, seed=8)
    tr = ostep.OracleTrainer(P, backbone="resnet18", loss=LOSS, optimizer="sgd", lr=0.02, architecture=arch, net_kwargs=okw)
    sm = ctor("resnet18", input_shape=(size, size, 3), classes=1, activation="sigmoid", encoder_weights=None, **kw)
    sm.compile(optimizer="SGD", loss=LOSS, lr=0.02, batch=n, dtype="fp32", use_graph=False)
    m = sm.impl
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    assert rel_l2(g["final_conv/kernel"], o["grads"]["final_conv/kernel"]) < 1e-4
    for k, ref in o["grads"].items():
        assert rel_l2(g[k], ref) <= 3e-2, k
    if kw.get("dropout"):
        # SpatialDropout2D: the masked tensor has whole (sample, channel) maps at zero, about `dropout` of them, the kept ones scaled;
        # a second step draws another mask (device step counter), inference is the identity
        name = "fpn_dropout" if arch == "FPN" else "psp_dropout"
        a = m.activation(name)
        dead = (np.abs(a).reshape(n, -1, a.shape[-1]).max(axis=1) == 0)
        assert 0.05 < dead.mean() < 0.6
        assert np.isfinite(m.train_on_batch(x, y)["loss"])
        dead2 = (np.abs(m.activation(name)).reshape(n, -1, a.shape[-1]).max(axis=1) == 0)
        assert not np.array_equal(dead, dead2)
        p1, p2 = m.predict(x), m.predict(x)
        np.testing.assert_array_equal(p1, p2)


@pytest.mark.parametrize("OS", [16, 8])
def test_fp32_deeplabv3_xception_step_matches_oracle(OS):
    """The xception branch of the reference's in-tree DeepLabV3+ (impl/deeplab/model.py:338-379 entry / middle / exit flow with
    SepConv_BN and the three skip types, :453-469 atrous SepConv ASPP, :471-491 decoder with feature_projection0), output stride
    16 and 8: fp32 step against the oracle that follows model.py line by line (41.25 M parameters, the published size of
    DeepLabV3+ / Xception-65), taps along the network, then the bf16 + hipGraph mode trains."""
    from oracle import deeplab as odl
    from segmentation_training_pipeline_amd import models
    n, size = 2, 64
    P = odl.init_deeplab_xception(seed=42)
    assert abs(sum(v.size for k, v in P.items() if "moving" not in k) / 1e6 - 41.1) < 0.3
    x, y = ostep.synthetic_batch(n, size, size, seed=21)
    tr = ostep.OracleTrainer(P, backbone="xception", loss=LOSS, optimizer="sgd", lr=0.02, architecture="DeepLabV3", net_kwargs={"OS": OS})
    sm = models.Deeplabv3(encoder_weights=None, input_shape=(size, size, 3), classes=1, backbone_name="xception", OS=OS, activation="sigmoid")
    sm.compile(optimizer="SGD", loss=LOSS, lr=0.02, batch=n, dtype="fp32", use_graph=False)
    m = sm.impl
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    taps = {}
    o = tr.step(x.astype(np.float32), y.astype(np.float32), taps=taps)
    m.load_batch(x, y)
    m.forward_backward()
    for oname, pname in (("entry_flow_block2", "entry_flow_block2_add"), ("middle_flow", "middle_flow_unit_16_add"),
                         ("exit_flow", "exit_flow_block2_separable_conv3_pointwise_BN"), ("concat_projection", "concat_projection_BN"),
                         ("decoder", "decoder_conv1_pointwise_BN")):
        ref = taps[oname].detach().numpy()
        if pname == "concat_projection_BN":
            continue                                                             # (the Dropout that follows works in place on this buffer)
        np.testing.assert_allclose(m.activation(pname), ref, atol=1e-3 * max(1.0, np.abs(ref).max()), err_msg=pname)
    m.apply_gradients()
    met = m.metrics()
    np.testing.assert_allclose(m.logits(), o["logits"], atol=2.5e-4)              # probabilities: 1e-3 on logits ~ 2.5e-4 on p
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    assert rel_l2(g["custom_logits_semantic/kernel"], o["grads"]["custom_logits_semantic/kernel"]) < 1e-4
    for k in ("decoder_conv1_pointwise/kernel", "feature_projection0/kernel", "aspp2_depthwise/depthwise_kernel", "exit_flow_block1_shortcut/kernel",
              "middle_flow_unit_8_separable_conv2_pointwise/kernel", "entry_flow_block2_separable_conv2_depthwise/depthwise_kernel",
              "entry_flow_conv1_1/kernel"):
        e = np.linalg.norm(g[k].astype(np.float64) - o["grads"][k]) / (np.linalg.norm(o["grads"][k].astype(np.float64)) + 1e-3)
        assert e <= 8e-2, "grad %s: rel L2 %.3g" % (k, e)
    mb = models.Deeplabv3(encoder_weights=None, input_shape=(size, size, 3), classes=1, backbone_name="xception", OS=OS, activation="sigmoid")
    mb.compile(optimizer="Adam", loss=LOSS, lr=1e-3, batch=n, dtype="bf16")
    mb.impl.set_weights(P)
    l0 = mb.impl.train_on_batch(x, y)["loss"]
    for _ in range(12):
        l1 = mb.impl.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0, (l0, l1)


@pytest.mark.parametrize("arch", ["FPN", "PSPNet"])
def test_vgg16_under_fpn_and_pspnet_matches_the_oracle(arch):
    """README.md:587-589 lists the VGG encoders for every architecture: FPN over block5_pool + block5 / block4 / block3 skip layers,
    PSPNet over block4_conv3 (1/8).  fp32 step at the north-star bars."""
    from segmentation_training_pipeline_amd import models
    n, size = 2, (64 if arch == "FPN" else 96)
    P = (onets.init_fpn_resnet if arch == "FPN" else onets.init_pspnet_resnet)("vgg16", seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=8)
    tr = ostep.OracleTrainer(P, backbone="vgg16", loss=LOSS, optimizer="sgd", lr=0.002, architecture=arch)
    sm = (models.FPN if arch == "FPN" else models.PSPNet)("vgg16", input_shape=(size, size, 3), classes=1, activation="sigmoid", encoder_weights=None)
    sm.compile(optimizer="SGD", loss=LOSS, lr=0.002, batch=n, dtype="fp32", use_graph=False)
    m = sm.impl
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3 * max(1.0, np.abs(o["logits"]).max()))
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5
    g = m.get_gradients()
    assert rel_l2(g["final_conv/kernel"], o["grads"]["final_conv/kernel"]) < 1e-4
    for k, ref in o["grads"].items():
        assert rel_l2(g[k], ref) <= 3e-2, k


@pytest.mark.parametrize("arch,spec", [("Unet", "binary_crossentropy+0.5*iou_loss+0.02*jaccard_loss"), ("Unet", "focal_loss+dice_loss"),
                                       ("FPN", "jaccard_loss"), ("Linknet", "iou_loss+0.5*focal_loss"),
                                       ("Unet", "lovasz_loss"), ("PSPNet", "binary_crossentropy+0.5*lovasz_loss")])
def test_registry_losses_step_matches_oracle(arch, spec):
    """The other names of the loss registry (reference segmentation.py:15-22: iou_loss, jaccard_loss, focal_loss) in the
    composite grammar of README.md:210-214, through one full training step."""
    n, size, backbone = 2, 64, "resnet18"
    if arch == "PSPNet":
        size = 96
    P = {"Unet": onets.init_unet_resnet, "Linknet": onets.init_linknet_resnet, "FPN": onets.init_fpn_resnet,
         "PSPNet": onets.init_pspnet_resnet}[arch](backbone, seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=77)
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.05, opt_kwargs={"momentum": 0.9}, architecture=arch)
    m = make(backbone, size, n, "fp32", optimizer="SGD", lr=0.05, opt_kwargs={"momentum": 0.9}, architecture=arch, loss=spec)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    m.load_batch(x, y)
    m.forward_backward()
    met = m.metrics()
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    assert abs(met["loss"] - o["loss"]) < 2e-5 * max(1.0, abs(o["loss"]))
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
    with pytest.raises(ValueError):
        make(backbone, size, n, "fp32", loss="hinge_loss")
    if "lovasz" in spec:      # the sort + scan launches capture into the step's hipGraph; bf16 mode learns under this loss
        mb = make(backbone, size, n, "bf16", use_graph=True, architecture=arch, loss=spec)
        mb.set_weights(P)
        l0 = mb.train_on_batch(x, y)
        for _ in range(12):
            l1 = mb.train_on_batch(x, y)
        assert np.isfinite(l1["loss"]) and l1["loss"] < l0["loss"] and l1["lovasz_loss"] < l0["lovasz_loss"]


def test_registry_losses_are_binary_head_only():
    from segmentation_training_pipeline_amd.backend import parse_loss
    assert parse_loss("binary_crossentropy+0.1*dice_loss") == (1.0, 0.1)
    assert parse_loss("iou_loss+2*focal_loss") == (0.0, 0.0, 1.0, 0.0, 2.0, 0.0)
    with pytest.raises(ValueError):
        parse_loss("categorical_crossentropy+iou_loss", classes=3)
    with pytest.raises(ValueError):
        parse_loss("jaccard_loss", classes=1, architecture="DeepLabV3")


@pytest.mark.parametrize("backbone,classes", [("mobilenetv2", 3), ("xception", 5)])
def test_fp32_deeplabv3_multiclass_head_matches_oracle(backbone, classes):
    """The reference's DeepLabV3 with `classes: N, activation: softmax` (model.py:485-486: the class convolution carries the
    channel softmax, BilinearUpsampling resizes the PROBABILITIES), categorical_crossentropy + dice on them."""
    from oracle import deeplab as odl
    from segmentation_training_pipeline_amd.backend import HipSegModel
    n, size = 2, 64
    P = (odl.init_deeplab_mobilenetv2 if backbone == "mobilenetv2" else odl.init_deeplab_xception)(classes=classes, seed=42)
    x, _ = ostep.synthetic_batch(n, size, size, seed=21)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 16 + xx // 24) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = "categorical_crossentropy+0.5*dice_loss"
    kw = {"net_kwargs": {"OS": 16}} if backbone == "xception" else {}
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.02, architecture="DeepLabV3", activation="softmax", **kw)
    m = HipSegModel("DeepLabV3", backbone, (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02,
                    use_graph=False, **kw)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=2.5e-4)            # probabilities: 1e-3 on logits ~ 2.5e-4 on p
    np.testing.assert_allclose(m.logits().sum(axis=-1), 1.0, atol=1e-5)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    assert abs(met["categorical_crossentropy"] - o["bce"]) < 1e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = np.linalg.norm(g[k].astype(np.float64) - ref) / (np.linalg.norm(ref.astype(np.float64)) + 1e-3)
        assert e <= (1e-4 if k.startswith("custom_logits") else 8e-2), "grad %s: rel L2 %.3g" % (k, e)
    m.set_weights(tr.P)
    pr = m.predict(x)
    assert pr.shape == (n, size, size, classes)
    np.testing.assert_allclose(pr, tr.forward(x.astype(np.float32)), atol=5e-4)
    # bf16 + hipGraph: runs and learns
    mb = HipSegModel("DeepLabV3", backbone, (size, size, 3), classes, "softmax", batch=n, dtype="bf16", loss=spec, optimizer="Adam", lr=1e-3, **kw)
    mb.set_weights(P)
    l0 = mb.train_on_batch(x, y)["loss"]
    for _ in range(15):
        l1 = mb.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0


@pytest.mark.parametrize("backbone", ["mobilenetv2", "xception"])
def test_fp32_deeplabv3_non_divisible_input_and_voc_layer_name(backbone):
    """impl/deeplab/model.py:440,445,480-481 size its pooling / upsampling layers with ceil(input / OS) and ceil(input / 4), and
    :494-497 names the class convolution 'logits_semantic' when there are 21 classes: a 65 x 65 input (feature maps 33 / 17 / 9 / 5
    under TF 'same' padding) with 21 softmax classes, one fp32 step against the oracle."""
    from oracle import deeplab as odl
    from segmentation_training_pipeline_amd.backend import HipSegModel
    n, size, classes = 2, 65, 21
    P = (odl.init_deeplab_mobilenetv2 if backbone == "mobilenetv2" else odl.init_deeplab_xception)(classes=classes, seed=42)
    assert "logits_semantic/kernel" in P and "custom_logits_semantic/kernel" not in P
    x, _ = ostep.synthetic_batch(n, size, size, seed=21)
    yy, xx = np.mgrid[0:size, 0:size]
    y = ((yy // 7 + xx // 11) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
    spec = "categorical_crossentropy+0.5*dice_loss"
    kw = {"net_kwargs": {"OS": 16}} if backbone == "xception" else {}
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.02, architecture="DeepLabV3", activation="softmax", **kw)
    m = HipSegModel("DeepLabV3", backbone, (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02,
                    use_graph=False, **kw)
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    assert m.logits().shape == (n, size, size, classes)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=2.5e-4)
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 2e-5
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = np.linalg.norm(g[k].astype(np.float64) - ref) / (np.linalg.norm(ref.astype(np.float64)) + 1e-3)
        assert e <= (1e-4 if k.startswith("logits_semantic") else 8e-2), "grad %s: rel L2 %.3g" % (k, e)


@pytest.mark.parametrize("backbone,block", [("resnet18", "transpose"), ("vgg16", "upsampling"), ("vgg16", "transpose")])
def test_linknet_transpose_blocks_and_vgg_encoders_match_the_oracle(backbone, block):
    """Linknet's `decoder_block_type: transpose` (1x1 -> Conv2DTranspose 4x4 s2 -> 1x1, schemas/segmentation.raml:166-169) and
    the VGG encoders under Linknet (README.md:587-589; block5_pool + four skips) through the YAML-facing constructor."""
    from segmentation_training_pipeline_amd import models
    n, size = 2, 64
    lr = 0.002 if backbone == "vgg16" else 0.05
    P = onets.init_linknet_resnet(backbone, seed=42, decoder_block_type=block)
    x, y = ostep.synthetic_batch(n, size, size, seed=8)
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=LOSS, optimizer="sgd", lr=lr, architecture="Linknet")
    sm = models.Linknet(backbone, input_shape=(size, size, 3), classes=1, activation="sigmoid", encoder_weights=None, decoder_block_type=block)
    sm.compile(optimizer="SGD", loss=LOSS, lr=lr, batch=n, dtype="fp32", use_graph=False)
    m = sm.impl
    assert sorted(m.get_weights()) == sorted(P)
    m.set_weights(P)
    for k, v in m.get_weights().items():
        np.testing.assert_array_equal(v, P[k], err_msg=k)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    met = m.train_on_batch(x, y)
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3 * max(1.0, np.abs(o["logits"]).max()))
    assert abs(met["dice_loss"] - o["dice_loss"]) < 1e-5 and abs(met["loss"] - o["loss"]) < 1e-5 * max(1.0, abs(o["loss"]))
    g = m.get_gradients()
    assert rel_l2(g["final_conv/kernel"], o["grads"]["final_conv/kernel"]) < 1e-4
    for k, ref in o["grads"].items():     # (ReLU-kink noise, see test_fp32_step_matches_oracle; 16-channel stages: one flip weighs more)
        assert rel_l2(g[k], ref) <= 6e-2, k
    w = m.get_weights()
    for k in tr.P:
        np.testing.assert_allclose(w[k], tr.P[k], atol=2e-4, err_msg=k)
    # bf16 + hipGraph: runs and learns
    sb = models.Linknet(backbone, input_shape=(size, size, 3), classes=1, activation="sigmoid", encoder_weights=None, decoder_block_type=block)
    sb.compile(optimizer="Adam", loss=LOSS, lr=1e-4 if backbone == "vgg16" else 1e-3, batch=n, dtype="bf16")
    sb.impl.set_weights(P)
    l0 = sb.impl.train_on_batch(x, y)["loss"]
    for _ in range(10):
        l1 = sb.impl.train_on_batch(x, y)["loss"]
    assert np.isfinite(l1) and l1 < l0


def test_split_upsample_data_gradient_matches_the_oracle(monkeypatch):
    """STP_UPCOLLAPSE_BWD=1 (opt-in): the data gradient of conv3x3(concat(UpSampling2D(2)(x), skip)) as two launches - the skip's
    3x3 data gradient and a 4x4 / stride-2 convolution of dY with the summed taps that lands on the low-resolution x directly
    (stp_weight_prepare_upcollapse_bwd_batched) - against the oracle's autograd, with the fused BatchNormalization backward."""
    monkeypatch.setenv("STP_UPCOLLAPSE_BWD", "1")
    n, size, backbone = 2, 64, "resnet18"
    P = onets.init_unet_resnet(backbone, seed=42)
    x, y = ostep.synthetic_batch(n, size, size, seed=1234)
    tr = ostep.OracleTrainer(P, backbone=backbone, loss=LOSS, optimizer="sgd", lr=0.05)
    m = make(backbone, size, n, "fp32", optimizer="SGD", lr=0.05)
    assert any(rec[2] == "stp_weight_prepare_upcollapse_bwd_batched" for rec in m.plan.prep)
    assert not any(rec[2].startswith("stp_upsample2x_bwd") for rec in m.plan.bwd if rec[0] is not None)
    m.set_weights(P)
    o = tr.step(x.astype(np.float32), y.astype(np.float32))
    m.load_batch(x, y)
    m.forward_backward()
    np.testing.assert_allclose(m.logits(), o["logits"], atol=1e-3)
    g = m.get_gradients()
    for k, ref in o["grads"].items():
        e = rel_l2(g[k], ref)
        assert e <= (1e-4 if k.startswith("final_conv") else 3e-2), "grad %s: rel L2 %.3g" % (k, e)
