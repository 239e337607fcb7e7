"""Step time of BASELINE.json configs[3] and [4] (not the headline metric): training step on a resident batch, hipGraph."""
import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd.backend import HipSegModel
import bench

# (.., dtype): configs[3] is named "fp16 MFMA" - the IEEE-half build (libstp_hip_f16.so); the bf16 line of the same workload beside it
CASES = [("FPN", "resnet50", 1024, 4, 3, "fp16"), ("FPN", "resnet50", 1024, 4, 3, "bf16"), ("PSPNet", "resnet101", 768, 8, 20, "bf16"),
         ("Linknet", "resnet34", 512, 16, 1, "bf16"), ("Unet", "resnet34", 512, 16, 1, "bf16"), ("Unet", "resnet34", 512, 16, 1, "fp16")]
if len(sys.argv) > 1:      # optional filter: "FPN:bf16 PSPNet:bf16"
    CASES = [c for c in CASES if "%s:%s" % (c[0], c[5]) in sys.argv[1:]]
for arch, bb, size, batch, classes, dtype in CASES:
    act = "sigmoid" if classes == 1 else "softmax"
    spec = "binary_crossentropy+1.0*dice_loss" if classes == 1 else "categorical_crossentropy+1.0*dice_loss"
    m = HipSegModel(arch, bb, (size, size, 3), classes, act, batch=batch, dtype=dtype, loss=spec, optimizer="Adam", lr=1e-3, use_graph=True)
    rng = np.random.RandomState(0)
    x = rng.randint(0, 256, (batch, size, size, 3)).astype(np.uint8)
    y = rng.randint(0, max(classes, 2), (batch, size, size, 1)).astype(np.uint8)
    m.load_batch(x, y)
    for _ in range(5):
        m.train_on_batch(None, None, fetch=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        m.train_on_batch(None, None, fetch=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = bench.flop_per_image(m)
    print(json.dumps({"workload": "%s/%s %dx%d %d-class bs%d %s" % (arch, bb, size, size, classes, batch, dtype), "ms_per_step": round(dt * 1e3, 3),
                      "images_per_sec": round(batch / dt, 1), "gflop_per_image": round(fl / 1e9, 1),
                      "step_mfma_frac": round(batch / dt * fl / 2.5e15, 4), "launches": len([l for l in m.plan.fwd + m.plan.bwd if l[0] is not None])}))
    del m
    torch.cuda.empty_cache()
