"""What-if timing of the row-of-taps weight-gradient kernel: EXP=1 no MFMA work, EXP=2 no loads in the loop (scratch/exp_build.sh)."""
import sys, os, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import _lib
exp = os.environ.get("EXP")
if exp:
    _lib.LIB_PATH = os.path.abspath("scratch/_exp/libstp_exp%s.so" % exp)
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, n, h, w, ci, co in [("stage1 64->64 @128", 16, 128, 128, 64, 64), ("stage2 128->128 @64", 16, 64, 64, 128, 128), ("stage4 512->512 @16", 16, 16, 16, 512, 512)]:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16); dy = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
    dw = torch.empty(co, 3, 3, ci, device=DEV)
    for splits in (0, 8):
        W = ops.wgrad_params(x, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, splits=splits)
        ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 4, dtype=torch.float32, device=DEV)
        for v in (2, 4):
            print("EXP=%s %-22s splits %d variant %d: %7.1f us" % (exp, name, splits, v, timeit(lambda: ops.conv2d_wgrad_partial(W, ws, v))))
