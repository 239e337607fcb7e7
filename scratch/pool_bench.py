"""maxpool 3x3 s2 forward / backward at the headline shape (16 x 256 x 256 x 64 bf16): us per launch and TB/s of the
algorithmic bytes (x + y + idx; dy + idx + dx)."""
import torch
from segmentation_training_pipeline_amd import ops
N, H, W, C = 16, 256, 256, 64
dev = "cuda"
x = torch.randn(N, H, W, C, device=dev).to(torch.bfloat16)
y = torch.empty(N, H // 2, W // 2, C, device=dev, dtype=torch.bfloat16)
idx = torch.empty(N, H // 2, W // 2, C, device=dev, dtype=torch.uint8)
dy = torch.randn_like(y)
dx = torch.empty_like(x)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
f = t(lambda: ops.maxpool3x3s2(x, y, idx, N, H, W, C))
b = t(lambda: ops.maxpool3x3s2_bwd(idx, dy, dx, N, H, W, C))
print("fwd %.1f us  %.2f TB/s" % (f, (x.numel() * 2 + y.numel() * 3) / f / 1e6))
print("bwd %.1f us  %.2f TB/s" % (b, (x.numel() * 2 + y.numel() * 3) / b / 1e6))
