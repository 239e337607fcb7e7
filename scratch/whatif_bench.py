"""What-if timing of the DMA GEMM kernels: full kernel vs loads-only (exp1) vs MFMA-only (exp2)."""
import sys, os, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import _lib
exp = os.environ.get("EXP")
if exp:
    _lib.LIB_PATH = os.path.abspath("scratch/_exp/libstp_exp%s.so" % exp)
from segmentation_training_pipeline_amd import ops
DEV = "cuda"
LAYERS = [("stage1 64->64 @128", 16, 128, 128, 64, 64, [69, 71, 103]), ("stage2 128->128 @64", 16, 64, 64, 128, 128, [65, 70]),
          ("stage3 256->256 @32", 16, 32, 32, 256, 256, [70, 65]), ("stage4 512->512 @16", 16, 16, 16, 512, 512, [133, 69])]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, n, h, w, ci, co, tiles in LAYERS:
    x = torch.randn(n, h, w, ci, device=DEV).to(torch.bfloat16)
    wt = (torch.randn(co, 3, 3, ci, device=DEV) / (9 * ci) ** 0.5).to(torch.bfloat16)
    fl = 2.0 * n * h * w * co * 9 * ci
    for tile in tiles:
        y = torch.empty(n, h, w, co, device=DEV, dtype=torch.bfloat16)
        P = ops.conv_params(x, wt, y, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16, tile=tile)
        us = timeit(lambda: ops.conv2d(P))
        print("EXP=%s fwd   %-22s tile %3d: %8.1f us %7.1f TF" % (exp, name, tile, us, fl / us / 1e6))
    dy = torch.randn(n, h, w, co, device=DEV).to(torch.bfloat16)
    dw = torch.empty(co * 9 * ci, device=DEV)
    W = ops.wgrad_params(x, dy, dw, N=n, Hs0=h, Ws0=w, Hv=h, Wv=w, C0=ci, KH=3, KW=3, stride=1, pad=1, Ho=h, Wo=w, Cout=co, dtype=ops.BF16)
    ws = torch.empty(ops.wgrad_workspace_bytes(W) // 4 + 16, device=DEV)
    lib = _lib.load()
    import ctypes as C
    for variant in (2,):
        us = timeit(lambda: _lib.check(lib.stp_conv2d_wgrad_partial(C.byref(W), ws.data_ptr(), ws.numel() * 4, variant, torch.cuda.current_stream().cuda_stream)))
        print("EXP=%s wgrad %-22s var %d   : %8.1f us %7.1f TF" % (exp, name, variant, us, fl / us / 1e6))
