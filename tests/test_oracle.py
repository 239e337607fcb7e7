"""CPU tests: the oracle against its golden vectors and against naive numpy restatements."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import losses, nets, np_ops, optim, step


def test_conv_param_count_matches_published_unet_resnet34():
    # ~24.4 M parameters is the publicly quoted size of Unet('resnet34').
    P = nets.init_unet_resnet("resnet34")
    assert nets.conv_param_count(P) == 24421456


@pytest.mark.parametrize("arch,backbone,size", [("Unet", "resnet18", 64), ("Unet", "resnet34", 64), ("Linknet", "resnet18", 64),
                                                ("FPN", "resnet18", 64), ("PSPNet", "resnet18", 96)])
def test_oracle_matches_golden(golden_dir, arch, backbone, size):
    g = np.load(os.path.join(golden_dir, "%s_%s_%d.npz" % (arch.lower(), backbone, size)))
    P = {"Unet": nets.init_unet_resnet, "Linknet": nets.init_linknet_resnet, "FPN": nets.init_fpn_resnet,
         "PSPNet": nets.init_pspnet_resnet}[arch](backbone, seed=int(g["seed"]))
    tr = step.OracleTrainer(P, backbone=backbone, loss="binary_crossentropy+1.0*dice_loss",
                            optimizer="adam", lr=1e-3, architecture=arch)
    xf, yf = g["x"].astype(np.float32), g["y"].astype(np.float32)
    o1 = tr.step(xf, yf)
    o2 = tr.step(xf, yf)
    keys = ("loss", "bce", "dice_loss", "dice", "binary_accuracy")
    np.testing.assert_allclose([o1[k] for k in keys], g["scalars1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose([o2[k] for k in keys], g["scalars2"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(o1["logits"], g["logits1"], atol=2e-4)
    names = [str(s) for s in g["grad_names"]]
    l2 = np.array([np.sqrt((o1["grads"][k].astype(np.float64) ** 2).sum()) for k in names])
    np.testing.assert_allclose(l2, g["grad_l2_step1"], rtol=1e-3, atol=1e-6)


def test_oracle_matches_the_fullsize_golden(golden_dir):
    """The 512 x 512 batch-2 fixture of BASELINE configs[1]'s network (one step, ~10 s of CPU): the oracle reproduces what it committed."""
    g = np.load(os.path.join(golden_dir, "unet_resnet34_512_bs2.npz"))
    size, n, stride = int(g["size"]), int(g["n"]), int(g["stride"])
    tr = step.OracleTrainer(nets.init_unet_resnet("resnet34", seed=int(g["seed"])), backbone="resnet34",
                            loss="binary_crossentropy+1.0*dice_loss", optimizer="adam", lr=1e-3)
    x, y = step.synthetic_batch(n, size, size, seed=int(g["data_seed"]))
    o1 = tr.step(x.astype(np.float32), y.astype(np.float32))
    np.testing.assert_allclose([o1[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], g["scalars1"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(o1["logits"][:, ::stride, ::stride, :], g["logits1_sampled"], atol=2e-4)
    np.testing.assert_allclose(o1["logits"].astype(np.float64).sum(axis=(2, 3)), g["logits1_row_sums"], atol=2e-4 * size)


@pytest.mark.parametrize("fname", ["fpn_resnet50_128_fp16.npz", "pspnet_resnet101_96_bf16.npz", "linknet_resnet34_128_bf16.npz"])
def test_storage_quantised_oracle_matches_its_golden(golden_dir, fname):
    """The storage-quantised oracle (nets._Ctx(storage=...)) on the non-U-Net graphs - BASELINE.json configs[3] / [4]'s networks and
    Linknet at their benchmarked precision - reproduces the fixture it committed (tests/golden/make_golden.py --storage-cases-only):
    every stored value IS a value of the storage format, the quantised step stays within the format's drift of the fp32 step, and the
    tap-channel class head (nets._class_head) equals the direct 3x3 convolution when nothing is rounded."""
    g = np.load(os.path.join(golden_dir, fname))
    arch, backbone, classes, storage = str(g["arch"]), str(g["backbone"]), int(g["classes"]), str(g["storage"])
    init = {"Linknet": nets.init_linknet_resnet, "FPN": nets.init_fpn_resnet, "PSPNet": nets.init_pspnet_resnet}[arch]
    P = init(backbone, classes=classes, seed=int(g["seed"]))
    act, spec = ("sigmoid", "binary_crossentropy+1.0*dice_loss") if classes == 1 else ("softmax", "categorical_crossentropy+1.0*dice_loss")
    tr = step.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="adam", lr=1e-3, architecture=arch, activation=act, storage=storage,
                            grad_scale=float(g["grad_scale"]))
    o = tr.step(g["x"].astype(np.float32), g["y"].astype(np.float32), apply=False)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[storage]
    lg = torch.from_numpy(o["logits"])
    assert torch.equal(lg.to(dt).to(torch.float32), lg)              # the logits are a STORED tensor
    ref = g["logits1"]
    ulp = 2.0 ** (np.floor(np.log2(float(np.abs(ref).max()))) - (7 if storage == "bf16" else 10))
    d = np.abs(o["logits"] - ref)
    assert d.mean() <= 0.25 * ulp and d.max() <= 8 * ulp, (d.mean() / ulp, d.max() / ulp)       # (bit-identical on the host that wrote it)
    np.testing.assert_allclose([o[k] for k in ("loss", "bce", "dice_loss", "dice", "binary_accuracy")], g["scalars1"], rtol=2e-3, atol=2e-4)
    d32 = np.abs(o["logits"] - g["logits1_fp32"].astype(np.float32))
    assert 0 < d32.mean() < 0.08 * float(np.abs(ref).max())          # quantised != fp32, and within the format's drift of it
    names = [str(s) for s in g["grad_names"]]
    l2 = np.array([np.sqrt((o["grads"][k].astype(np.float64) ** 2).sum()) for k in names])
    np.testing.assert_allclose(l2, g["grad_l2_step1"], rtol=0.1, atol=1e-3 * float(g["grad_l2_step1"].max()))


def test_tap_channel_class_head_equals_the_direct_convolution_without_rounding():
    """nets._class_head's tap-channel restatement (what graph.Plan.conv3x3_taps computes) against F.conv2d in float64: the identity
    W_t . shift_t(x) = shift_t(W_t . x) with zero padding, bias added once."""
    rng = np.random.RandomState(3)
    P = {"final_conv/kernel": torch.from_numpy(rng.randn(3, 3, 128, 5)).double(), "final_conv/bias": torch.from_numpy(rng.randn(5)).double()}
    x = torch.from_numpy(rng.randn(2, 128, 7, 9)).double()

    class _F64(nets._Ctx):          # storage "on" (the tap form is taken) with an identity rounding
        def st(self, t):
            return t

        def wq(self, w):
            return w
    ctx = _F64(P, True, None, storage=torch.float64)
    assert nets.class_head_uses_taps(128, 5)
    got = nets._class_head(ctx, x)
    want = F.conv2d(x, P["final_conv/kernel"].permute(3, 2, 0, 1), P["final_conv/bias"], padding=1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-10)


def test_rle_golden_vectors_from_reference(golden_dir):
    """The in-repo RLE restatement against vectors produced by the reference's own rle.py."""
    from segmentation_pipeline.impl import rle
    with open(os.path.join(golden_dir, "rle_golden.json")) as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 20
    for c in cases:
        m = np.array(c["mask"], np.uint8).reshape(c["shape"])
        assert rle.rle_encode(m) == c["rle"]
        if c["rle"]:
            dec = rle.rle_decode(c["rle"], tuple(c["shape"]))
            assert dec.shape == tuple(c["decoded_shape"])
            np.testing.assert_array_equal(dec, np.array(c["decoded"], np.uint8))


def test_np_conv_matches_torch():
    rng = np.random.RandomState(0)
    for (h, w, ci, co, k, s, p) in [(9, 11, 5, 7, 3, 1, 1), (12, 12, 4, 6, 3, 2, 1), (16, 14, 3, 8, 7, 2, 3),
                                    (8, 8, 6, 4, 1, 2, 0)]:
        x = rng.randn(2, h, w, ci).astype(np.float32)
        wt = rng.randn(k, k, ci, co).astype(np.float32)
        ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(wt).permute(3, 2, 0, 1),
                       stride=s, padding=p).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(np_ops.conv2d(x, wt, s, p), ref, atol=1e-4)
        # gradients vs autograd
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
        wtt = torch.from_numpy(wt).permute(3, 2, 0, 1).requires_grad_(True)
        out = F.conv2d(xt, wtt, stride=s, padding=p)
        dy = rng.randn(*out.shape).astype(np.float32)
        out.backward(torch.from_numpy(dy))
        dy_nhwc = dy.transpose(0, 2, 3, 1)
        np.testing.assert_allclose(np_ops.conv2d_dgrad(dy_nhwc, wt, (h, w), s, p),
                                   xt.grad.permute(0, 2, 3, 1).numpy(), atol=1e-4)
        np.testing.assert_allclose(np_ops.conv2d_wgrad(x, dy_nhwc, (k, k), s, p),
                                   wtt.grad.permute(2, 3, 1, 0).numpy(), atol=2e-4)


def test_np_bn_pool_upsample_match_oracle_net_ops():
    rng = np.random.RandomState(1)
    x = rng.randn(2, 8, 8, 6).astype(np.float32) * 3 + 1
    gamma = rng.rand(6).astype(np.float32) + 0.5
    beta = rng.randn(6).astype(np.float32)
    P = {"b/gamma": torch.from_numpy(gamma), "b/beta": torch.from_numpy(beta),
         "b/moving_mean": torch.zeros(6), "b/moving_variance": torch.ones(6)}
    ctx = nets._Ctx(P, True, None)
    xt = torch.from_numpy(x).permute(0, 3, 1, 2).requires_grad_(True)
    y = nets._bn_apply(ctx, xt, "b", 1e-3, relu=False)
    yn, mean, var = np_ops.bn_train(x, gamma, beta, 1e-3)
    np.testing.assert_allclose(y.detach().permute(0, 2, 3, 1).numpy(), yn, atol=1e-5)
    n = 2 * 8 * 8
    np.testing.assert_allclose(ctx.bn_updates["b/moving_variance"].numpy(),
                               0.99 + 0.01 * var * n / (n - 1), rtol=1e-5)
    dy = rng.randn(2, 8, 8, 6).astype(np.float32)
    y.backward(torch.from_numpy(dy).permute(0, 3, 1, 2))
    dx, dg, db = np_ops.bn_train_bwd(x, dy, gamma, 1e-3)
    np.testing.assert_allclose(xt.grad.permute(0, 2, 3, 1).numpy(), dx, atol=1e-5)
    np.testing.assert_allclose(P["b/gamma"].grad if P["b/gamma"].grad is not None else dg, dg)
    # pool / upsample
    xp = F.max_pool2d(F.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(np_ops.maxpool3x3s2(x), xp)
    up = F.interpolate(torch.from_numpy(x).permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1).numpy()
    np.testing.assert_array_equal(np_ops.upsample2x(x), up)
    np.testing.assert_allclose(np_ops.upsample2x_bwd(np_ops.upsample2x(x)), 4 * x, rtol=1e-6)


def test_keras_adam_known_answer():
    # one parameter, g = 0.5: m=0.05, v=2.5e-4, lr_t = 1e-3*sqrt(1-.999)/(1-.9)
    opt = optim.Adam(lr=1e-3)
    p = {"w": np.array([1.0], np.float32)}
    opt.step(p, {"w": np.array([0.5], np.float32)})
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    expect = 1.0 - lr_t * 0.05 / (np.sqrt(2.5e-4) + 1e-7)
    np.testing.assert_allclose(p["w"], [expect], rtol=1e-6)
    # SGD momentum form v = mu*v - lr*g ; p += v
    sgd = optim.SGD(lr=0.1, momentum=0.9)
    p = {"w": np.array([1.0], np.float32)}
    sgd.step(p, {"w": np.array([1.0], np.float32)})
    sgd.step(p, {"w": np.array([1.0], np.float32)})
    np.testing.assert_allclose(p["w"], [1.0 - 0.1 - (0.09 + 0.1)], rtol=1e-6)


def test_losses_basic_properties():
    y = torch.tensor([[1.0, 0.0, 1.0, 0.0]])
    p = torch.tensor([[0.9, 0.1, 0.8, 0.3]])
    bce = losses.binary_crossentropy(y, p)
    ref = -(np.log(0.9) + np.log(0.9) + np.log(0.8) + np.log(0.7)) / 4
    assert abs(float(bce) - ref) < 1e-6
    d = losses.dice_loss(y, p)
    assert abs(float(d) - (1 - (2 * 1.7 + 1) / (2 + 2.1 + 1))) < 1e-6
    assert losses.parse_loss("binary_crossentropy+0.1*dice_loss") == [(1.0, "binary_crossentropy"), (0.1, "dice_loss")]
    # clipping: p == 1 exactly with y == 0 stays finite (Keras epsilon clip)
    assert np.isfinite(float(losses.binary_crossentropy(torch.zeros(1), torch.ones(1))))


def test_nadam_and_rmsprop_first_steps_by_hand():
    """Keras 2.2.4 update rules written out for one scalar: guards the oracle the HIP optimizers are held to."""
    from oracle import optim
    g = np.array([0.5], np.float32)
    P = {"w": np.array([1.0], np.float32)}
    optim.RMSprop(lr=1e-3).step(P, {"w": g})
    a = 0.1 * 0.25
    assert abs(P["w"][0] - (1.0 - 1e-3 * 0.5 / (np.sqrt(a) + 1e-7))) < 1e-7
    P = {"w": np.array([1.0], np.float32)}
    o = optim.Nadam(lr=2e-3)
    o.step(P, {"w": g})
    mu1 = 0.9 * (1 - 0.5 * 0.96 ** 0.004)
    mu2 = 0.9 * (1 - 0.5 * 0.96 ** 0.008)
    m, v = 0.1 * 0.5, 0.001 * 0.25
    m_bar = (1 - mu1) * 0.5 / (1 - mu1) + mu2 * m / (1 - mu1 * mu2)
    want = 1.0 - 2e-3 * m_bar / (np.sqrt(v / (1 - 0.999)) + 1e-7)
    assert abs(P["w"][0] - want) < 1e-6 and abs(o.m_schedule - mu1) < 1e-12
