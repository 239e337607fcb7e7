"""stp_sigmoid_bce_dice at the headline shape (16 x 512 x 512 logits, bf16, 8-channel gradient rows): us per call."""
import torch
from segmentation_training_pipeline_amd import ops
count = 16 * 512 * 512
dev = "cuda"
z = torch.randn(count, device=dev).to(torch.bfloat16)
y = (torch.rand(count, device=dev) < 0.3).to(torch.uint8)
scal = torch.zeros(16, device=dev)
dl = torch.empty(count, 8, device=dev, dtype=torch.bfloat16)
ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=dev)
def t(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
print("loss fwd+grad %.1f us" % t(lambda: ops.sigmoid_bce_dice(z, y, count, 1.0, 1.0, scal, dl, 8, 1.0, ws)))
print("loss fwd only %.1f us" % t(lambda: ops.sigmoid_bce_dice(z, y, count, 1.0, 1.0, scal, None, 8, 1.0, ws)))
