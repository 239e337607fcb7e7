import sys, os, numpy as np
sys.path.insert(0, ".")
from oracle import nets as onets, step as ostep
from segmentation_training_pipeline_amd.backend import HipSegModel
def rel_l2(a, b): return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
n, size, classes, backbone = 1, 384, 4, "resnet18"
P = onets.init_pspnet_resnet(backbone, classes=classes, seed=42)
x, _ = ostep.synthetic_batch(n, size, size, seed=8)
yy, xx = np.mgrid[0:size, 0:size]
y = ((yy // 48 + xx // 64) % classes).astype(np.uint8)[None, :, :, None].repeat(n, axis=0)
spec = "categorical_crossentropy+1.0*dice_loss"
tr = ostep.OracleTrainer(P, backbone=backbone, loss=spec, optimizer="sgd", lr=0.02, architecture="PSPNet", activation="softmax")
o = tr.step(x.astype(np.float32), y.astype(np.float32))
for env in ({"STP_POOL_PYRAMID": "1", "STP_UP_LOSS": "1"}, {"STP_POOL_PYRAMID": "0", "STP_UP_LOSS": "1"}, {"STP_POOL_PYRAMID": "0", "STP_UP_LOSS": "0"}):
    os.environ.update(env)
    m = HipSegModel("PSPNet", backbone, (size, size, 3), classes, "softmax", batch=n, dtype="fp32", loss=spec, optimizer="SGD", lr=0.02, use_graph=False)
    m.set_weights(P)
    met = m.train_on_batch(x, y)
    g = m.get_gradients()
    errs = sorted(((rel_l2(g[k], ref), k) for k, ref in o["grads"].items()), reverse=True)
    print(env, "loss", met["loss"], o["loss"], "logits", float(np.abs(m.logits() - o["logits"]).max()), "worst grads", [(round(e, 4), k) for e, k in errs[:6]], flush=True)
