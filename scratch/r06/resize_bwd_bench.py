"""stp_resize_bilinear_bwd of FPN's pyramid levels (slices of the 512-channel concatenation at 4 x 256 x 256): tile kernel on / off (set STP_RESIZE_BWD_TILE)."""
import sys, os, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops, _lib
DEV = "cuda"
lib = _lib.load()
gy = [torch.randn(4, 256, 256, 512, device=DEV).to(torch.bfloat16) for _ in range(3)]      # 3 x 268 MB: rotated so no launch finds dY cached
for f, coff in ((2, 128), (4, 256), (8, 384)):
    h = 256 // f
    dx = torch.empty(4, h, h, 128, dtype=torch.bfloat16, device=DEV)
    wsb = int(lib.stp_resize_bilinear_bwd_workspace_bytes(4, h, h, 128, f))
    ws = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device=DEV)
    st = ops.stream()
    def run(i):
        _lib.call("stp_resize_bilinear_bwd", ops.ptr(gy[i % 3]), ops.ptr(dx), 4, h, h, 128, f, 512, coff, ops.BF16, 0, ops.ptr(ws) if wsb else None, wsb, st)
    run(0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): run(i)
    e1.record(); torch.cuda.synchronize()
    print("STP_RESIZE_BWD_TILE=%s x%d: %.1f us (67 MB of dY)" % (os.environ.get("STP_RESIZE_BWD_TILE", "1"), f, e0.elapsed_time(e1) * 1e3 / 30), flush=True)
