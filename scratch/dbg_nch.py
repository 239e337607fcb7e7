import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
rng = np.random.RandomState(3)
n, size, ch = 2, 64, 4
x = rng.randint(0, 60, (n, size, size, ch)).astype(np.uint8)
yy, xx = np.mgrid[0:size, 0:size]
m = ((yy - 32) ** 2 + (xx - 32) ** 2 <= 200)
x[:, :, :, 3][:, m] += 180
y = np.repeat(m[None, :, :, None].astype(np.uint8), n, 0)
mod = HipSegModel("Unet", "resnet18", (size, size, ch), 1, "sigmoid", batch=n, dtype="bf16", loss="binary_crossentropy", optimizer="Adam", lr=0.01)
for i in range(30):
    met = mod.train_on_batch(x, y)
print("train loss", met["loss"], "logits", mod.logits().min(), mod.logits().max())
p = mod.predict(x)
print("predict", p.shape, p.min(), p.max(), p[0, :, :, 0][m].mean(), p[0, :, :, 0][~m].mean())
