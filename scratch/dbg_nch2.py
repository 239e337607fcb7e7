import sys, os, tempfile, numpy as np, yaml
sys.path.insert(0, ".")
from segmentation_pipeline import segmentation
from segmentation_pipeline.impl.datasets import PredictionItem
rng = np.random.RandomState(3)
yy, xx = np.mgrid[0:64, 0:64]
items = []
for i in range(6):
    m = (((yy - rng.uniform(20, 44)) / rng.uniform(8, 18)) ** 2 + ((xx - rng.uniform(20, 44)) / rng.uniform(8, 18)) ** 2 <= 1)
    img = rng.randint(0, 60, (64, 64, 4)).astype(np.uint8)
    img[:, :, 3][m] += 180
    items.append(PredictionItem("s%d" % i, img, m[:, :, None].astype(np.uint8)))
class DS(object):
    def __len__(self): return len(items)
    def __getitem__(self, i): return items[i]
d = tempfile.mkdtemp()
base = {"architecture": "Unet", "backbone": "resnet18", "classes": 1, "activation": "sigmoid", "encoder_weights": None,
        "shape": [64, 64, 4], "optimizer": "Adam", "lr": 0.01, "batch": 2, "folds_count": 2, "loss": "binary_crossentropy",
        "metrics": ["binary_accuracy", "dice"], "primary_metric": "val_loss", "draw_examples": False,
        "augmentation": {"Fliplr": 0.5}, "stages": [{"epochs": 8}]}
p = os.path.join(d, "config.yaml")
yaml.safe_dump(base, open(p, "w"))
cfg = segmentation.parse(p)
cfg.fit(DS(), foldsToExecute=[0])
model = cfg.load_model(0, 0)
xs = cfg._resize_to_net(model.impl, [items[0].x])
print("xs", xs.shape, xs[..., 3].max(), xs[..., 0].max())
pr = cfg.predict_on_batch(model, False, xs)
print("p", pr.min(), pr.max(), pr.mean())
impl = model.impl
print("logits", impl.eval_plan().tensors["final_conv"].buf.float().min().item(), impl.eval_plan().tensors["final_conv"].buf.float().max().item())
w = impl.get_weights()
print("conv0", w["conv0/kernel"].shape, np.abs(w["conv0/kernel"]).max(), "bn_data beta", w["bn_data/beta"])
st = impl.get_state() if hasattr(impl, "get_state") else None
