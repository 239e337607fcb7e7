"""What-if: step time in graph mode with a named launch skipped (results are garbage; only the time matters).
usage: python scratch/whatif_skip.py stp_bn_finalize"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import graph
skip = set(sys.argv[1:])
orig = graph.Plan.run
def run(self, launches):
    return orig(self, [l for l in launches if l[2] not in skip])
graph.Plan.run = run
from segmentation_training_pipeline_amd.backend import HipSegModel
m = HipSegModel("Unet", "resnet34", (512, 512, 3), 1, "sigmoid", batch=16, dtype="bf16", loss="binary_crossentropy+1.0*dice_loss", use_graph=True)
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (16, 512, 512, 3)).astype(np.uint8), (rng.rand(16, 512, 512, 1) < 0.2).astype(np.uint8))
for _ in range(5): m.train_on_batch(None, None, fetch=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): m.train_on_batch(None, None, fetch=False)
torch.cuda.synchronize()
print("skip %s: %.3f ms/step" % (sorted(skip), (time.perf_counter() - t0) / 30 * 1e3))
