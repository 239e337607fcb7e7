"""stp_softmax_cce_dice_up against the chain it replaces, at the head shapes of configs[4] (PSPNet, 20 classes, x8) and configs[3] (FPN, 3 classes, x4)."""
import sys, torch
sys.path.insert(0, ".")
from segmentation_training_pipeline_amd import ops, _lib
DEV = "cuda"
lib = _lib.load()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, n, h, w, f, classes, ldc, dlc in (("pspnet 8x96x96 x8 20 classes", 8, 96, 96, 8, 20, 20, 24), ("fpn 4x256x256 x4 3 classes", 4, 256, 256, 4, 3, 3, 8)):
    ho, wo = h * f, w * f
    z = (torch.randn(n, h, w, ldc, device=DEV) * 2).to(torch.bfloat16)
    t = torch.randint(0, classes, (n, ho, wo), device=DEV, dtype=torch.uint8)
    ws = torch.empty(ops.loss_workspace_bytes() // 4, dtype=torch.float32, device=DEV)
    up = torch.empty((n, ho, wo, ldc), dtype=torch.bfloat16, device=DEV)
    dl = torch.empty((n, ho, wo, dlc), dtype=torch.bfloat16, device=DEV)
    dlow = torch.empty((n, h, w, dlc), dtype=torch.bfloat16, device=DEV)
    scal = torch.zeros(12, device=DEV)
    wsb = int(lib.stp_resize_bilinear_bwd_workspace_bytes(n, h, w, dlc, f))
    wsr = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device=DEV)
    nb = int(lib.stp_softmax_cce_dice_up_corner_bytes(n, h, w, classes))
    corners = torch.empty(nb // 4, dtype=torch.float32, device=DEV)
    st = ops.stream()
    a = timeit(lambda: _lib.call("stp_resize_bilinear", ops.ptr(z), ops.ptr(up), n, h, w, ldc, f, ldc, 0, ops.BF16, st))
    b = timeit(lambda: _lib.call("stp_softmax_cce_dice", ops.ptr(up), ops.ptr(t), n * ho * wo, classes, ldc, ops.BF16, 1.0, 1.0, ops.ptr(scal), ops.ptr(dl), dlc, 1.0, ops.ptr(ws), ws.numel() * 4, st))
    c = timeit(lambda: _lib.call("stp_resize_bilinear_bwd", ops.ptr(dl), ops.ptr(dlow), n, h, w, dlc, f, dlc, 0, ops.BF16, 0, ops.ptr(wsr) if wsb else None, wsb, st))
    d = timeit(lambda: _lib.call("stp_softmax_cce_dice_up", ops.ptr(z), ops.ptr(t), n, h, w, f, classes, ldc, ops.BF16, 1.0, 1.0, ops.ptr(scal), ops.ptr(dlow), dlc, 1.0, None, None, ops.ptr(ws), ws.numel() * 4, ops.ptr(corners), nb, st))
    e = timeit(lambda: _lib.call("stp_softmax_cce_dice_up", ops.ptr(z), ops.ptr(t), n, h, w, f, classes, ldc, ops.BF16, 1.0, 1.0, ops.ptr(scal), None, 0, 1.0, None, None, ops.ptr(ws), ws.numel() * 4, None, 0, st))
    # agreement at the full shape (the op test holds it on small maps)
    scal0, scal1 = torch.zeros(12, device=DEV), torch.zeros(12, device=DEV)
    dlow0, dlow1 = torch.empty_like(dlow), torch.empty_like(dlow)
    _lib.call("stp_resize_bilinear", ops.ptr(z), ops.ptr(up), n, h, w, ldc, f, ldc, 0, ops.BF16, st)
    _lib.call("stp_softmax_cce_dice", ops.ptr(up), ops.ptr(t), n * ho * wo, classes, ldc, ops.BF16, 1.0, 1.0, ops.ptr(scal0), ops.ptr(dl), dlc, 1024.0, ops.ptr(ws), ws.numel() * 4, st)
    _lib.call("stp_resize_bilinear_bwd", ops.ptr(dl), ops.ptr(dlow0), n, h, w, dlc, f, dlc, 0, ops.BF16, 0, ops.ptr(wsr) if wsb else None, wsb, st)
    _lib.call("stp_softmax_cce_dice_up", ops.ptr(z), ops.ptr(t), n, h, w, f, classes, ldc, ops.BF16, 1.0, 1.0, ops.ptr(scal1), ops.ptr(dlow1), dlc, 1024.0, None, None, ops.ptr(ws), ws.numel() * 4, ops.ptr(corners), nb, st)
    torch.cuda.synchronize()
    g0, g1 = dlow0.float(), dlow1.float()
    print("   scalars chain", [round(v, 6) for v in scal0[:5].tolist()], "fused", [round(v, 6) for v in scal1[:5].tolist()])
    print("   gradient: max |chain| %.4g, max |diff| %.4g, mean |diff| %.4g, equal %.4f, finite %s" % (g0.abs().max().item(), (g0 - g1).abs().max().item(), (g0 - g1).abs().mean().item(), (g0 == g1).float().mean().item(), bool(torch.isfinite(g1).all())))
    print("%-32s chain: resize %.1f + loss %.1f + resize gradient %.1f = %.1f us   fused: %.1f us (value pass alone %.1f)" % (name, a, b, c, a + b + c, d, e), flush=True)
