"""stp_bn_stats on the raw uint8 image batch (16 x 512 x 512 x 3): us per call."""
import torch
from segmentation_training_pipeline_amd import ops
rows, C = 16 * 512 * 512, 3
dev = "cuda"
x = torch.randint(0, 256, (rows, C), device=dev, dtype=torch.uint8)
mean, rstd = torch.empty(C, device=dev), torch.empty(C, device=dev)
mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
ws = torch.empty(1024 * 2 * 8, device=dev)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print("u8 stats %.1f us" % t(lambda: ops.bn_stats(x, rows, C, 1e-3, 0.99, mean, rstd, mm, mv, ws)))
xf = x.float()
print(mean.cpu().numpy(), xf.mean(0).cpu().numpy(), (1 / torch.sqrt(xf.var(0, unbiased=False) + 1e-3)).cpu().numpy(), rstd.cpu().numpy())
