"""Wall time of the step's phases (eager launch lists, events around each list)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
m = HipSegModel("Unet", "resnet34", (512, 512, 3), 1, "sigmoid", batch=16, dtype="bf16", loss="binary_crossentropy+1.0*dice_loss", use_graph=False)
p = m.plan
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (16, 512, 512, 3)).astype(np.uint8), (rng.rand(16, 512, 512, 1) < 0.2).astype(np.uint8))
def t(lst, n=10):
    p.run(lst); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): p.run(lst)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(2): m.train_on_batch(None, None)
print("prep %.3f ms  fwd %.3f ms  bwd %.3f ms  opt %.3f ms" % (t(p.prep), t(p.fwd), t(p.bwd), t(p.opt)))
names = {}
for fn, a, n, meta in p.fwd: names[n] = names.get(n, 0) + 1
print("fwd launches", sum(names.values()), names)
names = {}
for fn, a, n, meta in p.bwd: names[n] = names.get(n, 0) + 1
print("bwd launches", sum(names.values()), names)
