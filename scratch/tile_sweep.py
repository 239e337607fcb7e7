"""Autotune probe: every forward / data-gradient GEMM launch of the headline plan timed under each tile id (fused epilogues as in
the real step; statistics redirected to a scratch buffer because their tile count depends on the tile)."""
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
arch, bb, size, batch, classes = (sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else ("Unet", "resnet34", 512, 16, 1)
m = HipSegModel(arch, bb, (size, size, 3), classes, "sigmoid" if classes == 1 else "softmax", batch=batch, dtype="bf16",
                loss="binary_crossentropy+1.0*dice_loss" if classes == 1 else "categorical_crossentropy+1.0*dice_loss", use_graph=False)
p = m.plan
rng = np.random.RandomState(0)
m.load_batch(rng.randint(0, 256, (batch, size, size, 3)).astype(np.uint8), (rng.rand(batch, size, size, 1) < 0.2).astype(np.uint8))
m.train_on_batch(None, None)
st = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
lib = p.lib
TILES = [65, 69, 70, 71, 97, 101, 102, 103, 133, 134]
def timeit(cp, n=10):
    rc = lib.stp_conv2d(cp, st)
    if rc != 0:
        return None
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.stp_conv2d(cp, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
tot_auto = tot_best = 0.0
for lst in (p.fwd, p.bwd):
    for fn, args, name, meta in lst:
        if name != "stp_conv2d" or not meta:
            continue
        cp = args[0]._obj
        auto = int(meta["tile"])
        if auto >= 512 or auto < 64:
            continue                      # small-channel / stem kernels: not part of this sweep
        TL = TILES if auto < 256 else [257, 258, 259, 260, 261, 262, 263]
        keep_stats, keep_tile = cp.stats_partial, cp.tile
        if cp.stats_partial:
            cp.stats_partial = scratch.data_ptr()
        res = {}
        for t in [0] + TL:
            cp.tile = t
            us = timeit(cp)
            if us is not None:
                res[t] = us
        cp.stats_partial, cp.tile = keep_stats, keep_tile
        best = min((t for t in res if t), key=lambda t: res[t])
        tot_auto += res[0]; tot_best += res[best]
        flag = "" if res[best] > 0.95 * res[0] else "   <-- %.0f%%" % (100 * (1 - res[best] / res[0]))
        print("%-24s %-5s auto %3d %6.1f us | best %3d %6.1f us%s | %s" % (meta["layer"], meta["pass"], auto, res[0], best, res[best], flag,
              " ".join("%d:%.0f" % (t, res[t]) for t in TL if t in res)))
print("sum auto %.1f us, sum best %.1f us" % (tot_auto, tot_best))
