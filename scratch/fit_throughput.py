"""Throughput of the full host path (dataset items in host memory -> pinned prefetch -> H2D -> on-device augment -> step),
i.e. the PCIe-inclusive rate of the fit() loop, next to bench.py's resident-batch rate."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import augment, pipeline
from segmentation_training_pipeline_amd.backend import HipSegModel
from segmentation_pipeline.impl.datasets import PredictionItem
H = W = 512
rng = np.random.RandomState(0)
imgs = [rng.randint(0, 256, (H, W, 3)).astype(np.uint8) for _ in range(64)]
msks = [(rng.rand(H, W, 1) < 0.2).astype(np.uint8) for _ in range(64)]
class DS(object):
    def __len__(self): return 64
    def __getitem__(self, i): return PredictionItem("s%d" % i, imgs[i], msks[i])
m = HipSegModel("Unet", "resnet34", (H, W, 3), 1, "sigmoid", batch=16, dtype="bf16", loss="binary_crossentropy+1.0*dice_loss", optimizer="Adam", lr=1e-3)
feeder = pipeline.DeviceFeeder(m.device, (H, W), augment.BENCH_SPEC, seed=1)
tr = pipeline.Trainer(m, feeder, DS(), [], 0, 1)
idx = list(range(64)) * 24      # 96 steps
tr.run_epoch(idx[:64], True); torch.cuda.synchronize()
t0 = time.time(); logs = tr.run_epoch(idx, True); torch.cuda.synchronize(); dt = time.time() - t0
print("fit loop: %.1f images/s (%.2f ms/step), loss %.4f" % (len(idx) / dt, 1e3 * dt / (len(idx) / 16), logs["loss"]))
