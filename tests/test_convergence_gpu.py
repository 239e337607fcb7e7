"""bf16 storage mode is a legitimate TRAINING mode, not only a throughput mode: the same learnable synthetic
task (bright ellipses on noise, as tests/test_fit_gpu.py) is trained for 240 steps from identical initial weights on identical
batches in fp32 mode (the parity mode: exact-fp32 MFMA, held to the oracle at 1e-3 / 1e-5) and in bf16 mode (the benchmarked
mode); both must learn it, and the held-out Dice of the two runs must agree.  Per-step bf16 gradients differ from fp32 ones by
accumulated 8-bit roundings; what matters for training is that the optimisation trajectory lands in the same place."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SIZE, BATCH, STEPS = 64, 8, 240


def ellipses(n, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:SIZE, 0:SIZE]
    xs, ys = np.empty((n, SIZE, SIZE, 3), np.uint8), np.empty((n, SIZE, SIZE, 1), np.uint8)
    for i in range(n):
        m = (((yy - rng.uniform(14, 50)) / rng.uniform(7, 20)) ** 2 + ((xx - rng.uniform(14, 50)) / rng.uniform(7, 20)) ** 2 <= 1)
        img = rng.randint(0, 80, (SIZE, SIZE, 3)).astype(np.uint8)
        img[m] += 150
        xs[i], ys[i, :, :, 0] = img, m
    return xs, ys


def run(dtype, xs, ys, xv, yv):
    from segmentation_training_pipeline_amd.backend import HipSegModel
    m = HipSegModel("Unet", "resnet18", (SIZE, SIZE, 3), 1, "sigmoid", batch=BATCH, dtype=dtype, loss="binary_crossentropy+1.0*dice_loss",
                    optimizer="Adam", lr=1e-3, use_graph=True, device="cuda:0", seed=11)
    order = np.random.RandomState(5).randint(0, len(xs), size=(STEPS, BATCH))
    losses = []
    for idx in order:
        losses.append(m.train_on_batch(xs[idx], ys[idx])["loss"])
    p = m.predict(xv) > 0.5
    g = yv > 0
    dice = 2.0 * np.logical_and(p, g).sum() / max(1, p.sum() + g.sum())
    return np.array(losses), float(dice)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_16bit_modes_train_to_the_same_dice_as_fp32_mode(mode):
    """mode "fp16": the IEEE-half build (libstp_hip_f16.so) under its static loss scale 2^14 - 240 steps without a non-finite value."""
    assert torch.cuda.is_available()
    xs, ys = ellipses(96, 1)
    xv, yv = ellipses(48, 2)
    l32, d32 = run("fp32", xs, ys, xv, yv)
    l16, d16 = run(mode, xs, ys, xv, yv)
    print(("held-out Dice after %d steps: fp32 mode %.4f, " + mode + " mode %.4f; loss %.3f -> %.3f (fp32), %.3f -> %.3f (16-bit)")
          % (STEPS, d32, d16, l32[:20].mean(), l32[-20:].mean(), l16[:20].mean(), l16[-20:].mean()))
    for l in (l32, l16):
        assert np.all(np.isfinite(l))
        assert l[-20:].mean() < 0.35 * l[:20].mean()                  # both learn the task ...
        blocks = l[: STEPS // 40 * 40].reshape(-1, 40).mean(axis=1)
        assert np.all(np.diff(blocks) < 0.05 * blocks[0])              # ... without the 40-step mean loss ever rising noticeably
    assert d32 > 0.9 and d16 > 0.9
    assert abs(d32 - d16) <= 1e-2, (d32, d16)
    # the two loss curves track each other: same batches, same initial weights
    assert abs(l32[-40:].mean() - l16[-40:].mean()) < 0.15 * l32[:20].mean()


def ellipses_at(size, n, seed):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    xs, ys = np.empty((n, size, size, 3), np.uint8), np.empty((n, size, size, 1), np.uint8)
    for i in range(n):
        m = np.zeros((size, size), bool)
        for _ in range(3):
            m |= (((yy - rng.uniform(0.15, 0.85) * size) / (rng.uniform(0.05, 0.2) * size)) ** 2 +
                  ((xx - rng.uniform(0.15, 0.85) * size) / (rng.uniform(0.05, 0.2) * size)) ** 2 <= 1)
        img = rng.randint(0, 80, (size, size, 3)).astype(np.uint8)
        img[m] += 150
        xs[i], ys[i, :, :, 0] = img, m
    return xs, ys


def test_headline_bf16_trains_like_fp32():
    """The BENCHMARKED workload in the benchmarked precision: U-Net/ResNet34, 512 x 512, batch 16 - the shapes, tiles and
    grouped weight gradients bench.py times - trained for 400 Adam steps on the ellipse task from identical initial weights on
    identical batches in fp32 mode (the parity mode, 35 ms / step) and in bf16 mode.  One bf16 step's stage-1..3 gradients sit at
    cosine 0.79-0.81 to the storage-quantised oracle's (tests/test_model_gpu.py: rounding ties spread by ~50 layers); what that is
    worth is decided here: both runs must learn the task, land on the same held-out Dice (gap <= 1e-2) and keep their loss curves
    within 15 % of the initial loss of each other."""
    from segmentation_training_pipeline_amd.backend import HipSegModel
    size, batch, steps = 512, 16, 400      # (160 steps: the moving statistics that predict() normalises with are 80 % converged at
    #                                        momentum 0.99 - held-out Dice 0.92 vs 0.99 at equal training loss; 400: 98 %)
    xs, ys = ellipses_at(size, 64, 1)
    xv, yv = ellipses_at(size, 16, 2)
    order = np.random.RandomState(5).randint(0, len(xs), size=(steps, batch))
    res = {}
    for dtype in ("fp32", "bf16"):
        m = HipSegModel("Unet", "resnet34", (size, size, 3), 1, "sigmoid", batch=batch, dtype=dtype, loss="binary_crossentropy+1.0*dice_loss",
                        optimizer="Adam", lr=1e-3, use_graph=True, device="cuda:0", seed=11)
        losses = np.array([m.train_on_batch(xs[idx], ys[idx])["loss"] for idx in order])
        p, g = m.predict(xv) > 0.5, yv > 0
        res[dtype] = (losses, 2.0 * np.logical_and(p, g).sum() / max(1, p.sum() + g.sum()))
        del m
        torch.cuda.empty_cache()
    (l32, d32), (l16, d16) = res["fp32"], res["bf16"]
    print("U-Net/ResNet34 512x512 bs16, %d steps: held-out Dice fp32 mode %.4f, bf16 mode %.4f; loss %.3f -> %.3f (fp32), %.3f -> %.3f (bf16)"
          % (steps, d32, d16, l32[:10].mean(), l32[-20:].mean(), l16[:10].mean(), l16[-20:].mean()))
    for l in (l32, l16):
        assert np.all(np.isfinite(l)) and l[-20:].mean() < 0.35 * l[:10].mean()
    assert d32 > 0.9 and d16 > 0.9
    assert abs(d32 - d16) <= 1e-2, (d32, d16)
    b32, b16 = l32.reshape(-1, 20).mean(axis=1), l16.reshape(-1, 20).mean(axis=1)
    assert np.all(np.abs(b32 - b16) < 0.15 * l32[:10].mean()), (b32, b16)      # the 20-step mean losses track each other all the way
