"""Thin Python wrappers over the C-ABI (one function per entry point of include/stp_hip.h).

They take torch tensors that live on the GPU, pass raw device pointers + the current HIP stream,
and raise on any error.  No arithmetic happens in Python or in torch here.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F16, F32, U8, SRC_DIRECT, SRC_NEAREST2X, SRC_ZEROINS2X  # noqa: F401

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16, torch.uint8: U8}


def dt(t):
    return _DT[t.dtype]


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t):
    if t is not None and not t.is_cuda:
        raise _lib.StpError("tensor must live on the GPU (no CPU fallback)")
    return t


def conv_params(src0, weight, dst0, *, N, Hs0, Ws0, Hv, Wv, C0, KH, KW, stride, pad, Ho, Wo, Cout, dtype,
                src1=None, C1=0, mode=SRC_DIRECT, bias=None, residual=None, dst1=None, Cd0=None,
                accumulate0=0, accumulate1=0, relu=0, tile=0):
    p = _lib.ConvParams()
    p.src0, p.src1, p.weight, p.bias, p.residual = ptr(src0), ptr(src1), ptr(weight), ptr(bias), ptr(residual)
    p.dst0, p.dst1 = ptr(dst0), ptr(dst1)
    p.N, p.Hs0, p.Ws0, p.Hv, p.Wv, p.C0, p.C1 = N, Hs0, Ws0, Hv, Wv, C0, C1
    p.src0_mode, p.KH, p.KW, p.stride, p.pad = mode, KH, KW, stride, pad
    p.Ho, p.Wo, p.Cout, p.Cd0 = Ho, Wo, Cout, (Cout if Cd0 is None else Cd0)
    p.accumulate0, p.accumulate1, p.relu, p.dtype, p.tile = accumulate0, accumulate1, relu, dtype, tile
    return p


def conv2d(p, st=None):
    _lib.check(_lib.load().stp_conv2d(C.byref(p), stream() if st is None else st), "stp_conv2d")


def conv2d_stats_floats(p):
    return int(_lib.load().stp_conv2d_stats_floats(C.byref(p)))


def wgrad_params(src0, dy, dw, *, N, Hs0, Ws0, Hv, Wv, C0, KH, KW, stride, pad, Ho, Wo, Cout, dtype,
                 src1=None, C1=0, mode=SRC_DIRECT, accumulate=0, splits=0):
    p = _lib.WgradParams()
    p.src0, p.src1, p.dy, p.dw = ptr(src0), ptr(src1), ptr(dy), ptr(dw)
    p.N, p.Hs0, p.Ws0, p.Hv, p.Wv, p.C0, p.C1, p.src0_mode = N, Hs0, Ws0, Hv, Wv, C0, C1, mode
    p.KH, p.KW, p.stride, p.pad, p.Ho, p.Wo, p.Cout = KH, KW, stride, pad, Ho, Wo, Cout
    p.accumulate, p.dtype, p.splits = accumulate, dtype, splits
    return p


def wgrad_workspace_bytes(p):
    return int(_lib.load().stp_conv2d_wgrad_workspace_bytes(C.byref(p)))


def conv2d_wgrad(p, workspace, st=None):
    _lib.check(_lib.load().stp_conv2d_wgrad(C.byref(p), ptr(workspace), workspace.numel() * workspace.element_size(),
                                            stream() if st is None else st), "stp_conv2d_wgrad")


def conv2d_wgrad_partial(p, workspace, variant=0):
    _lib.check(_lib.load().stp_conv2d_wgrad_partial(C.byref(p), ptr(workspace), workspace.numel() * workspace.element_size(),
                                                    variant, stream()), "stp_conv2d_wgrad_partial")


def conv2d_wgrad_reduce(p, workspace, variant=0):
    _lib.check(_lib.load().stp_conv2d_wgrad_reduce(C.byref(p), ptr(workspace), variant, stream()), "stp_conv2d_wgrad_reduce")


class WgradGroup(object):
    """Descriptor table of a grouped weight gradient (stp_wgrad_group_*): built from the layers' parameter blocks once their
    device pointers are final; holds the host table, its device copy and the partial-slab workspace."""

    def __init__(self, params, device="cuda"):
        lib = _lib.load()
        self.params = list(params)
        n = len(self.params)
        self.arr = (C.POINTER(_lib.WgradParams) * n)(*[C.pointer(p) for p in self.params])
        tb = int(lib.stp_wgrad_group_table_bytes(self.arr, n))
        wsb = int(lib.stp_wgrad_group_workspace_bytes(self.arr, n))
        if tb <= 0 or wsb <= 0:
            raise _lib.StpError("the layers do not form a weight-gradient group")
        self.host = (C.c_char * tb)()
        _lib.check(lib.stp_wgrad_group_build(self.arr, n, C.addressof(self.host), tb), "stp_wgrad_group_build")
        self.dev = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8).to(device)
        self.ws = torch.empty(wsb // 4, dtype=torch.float32, device=device)
        self.header = list((C.c_int32 * 16).from_buffer_copy(bytes(self.host)[:64]))   # magic, bm, layers, segments, workgroups, tiles, ..., [14] = pbn

    def partial(self):
        _lib.check(_lib.load().stp_wgrad_group_partial(C.addressof(self.host), self.dev.data_ptr(), self.ws.data_ptr(), self.ws.numel() * 4, stream()),
                   "stp_wgrad_group_partial")

    def reduce(self):
        _lib.check(_lib.load().stp_wgrad_group_reduce(C.addressof(self.host), self.dev.data_ptr(), self.ws.data_ptr(), stream()), "stp_wgrad_group_reduce")

    def run(self):
        self.partial()
        self.reduce()


def wgrad_group_class(p):
    return int(_lib.load().stp_wgrad_group_class(C.byref(p)))


def weight_prepare(master, fwd, bwd, Cout, KH, KW, Cin, KWp, Cinp, CoutB, dtype):
    _lib.call("stp_weight_prepare", ptr(master), ptr(fwd), ptr(bwd), Cout, KH, KW, Cin, KWp, Cinp, CoutB, dtype, stream())


def weight_grad_unpad(padded, grad, Cout, KH, KW, Cin, KWp, Cinp, accumulate=0):
    _lib.call("stp_weight_grad_unpad", ptr(padded), ptr(grad), Cout, KH, KW, Cin, KWp, Cinp, accumulate, stream())


def stem_beta_grad(padded_dw, master, dbeta, Cout, KH, KW, Cin, KWp, Cinp, one_ch):
    _lib.call("stp_stem_beta_grad", ptr(padded_dw), ptr(master), ptr(dbeta), Cout, KH, KW, Cin, KWp, Cinp, one_ch, stream())


def bn_workspace_bytes(Cn):
    return int(_lib.load().stp_bn_workspace_bytes(Cn))


def bn_stats(x, rows, Cn, eps, momentum, mean, rstd, moving_mean, moving_var, workspace):
    _lib.call("stp_bn_stats", ptr(_dev(x)), dt(x), rows, Cn, eps, momentum, ptr(mean), ptr(rstd), ptr(moving_mean),
              ptr(moving_var), ptr(workspace), workspace.numel() * workspace.element_size(), stream())


def bn_apply(x, y, rows, Cn, Cy, mean, rstd, gamma, beta, relu, pad_value=0.0):
    _lib.call("stp_bn_apply", ptr(x), dt(x), ptr(y), dt(y), rows, Cn, Cy, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
              int(relu), float(pad_value), stream())


def bn_inference(x, y, rows, Cn, Cy, mm, mv, eps, gamma, beta, relu, pad_value=0.0):
    _lib.call("stp_bn_inference", ptr(x), dt(x), ptr(y), dt(y), rows, Cn, Cy, ptr(mm), ptr(mv), float(eps), ptr(gamma),
              ptr(beta), int(relu), float(pad_value), stream())


def bn_backward(x, dy, dx, rows, Cn, mean, rstd, gamma, beta, dgamma, dbeta, relu, accumulate_dx, workspace):
    _lib.call("stp_bn_backward", ptr(x), ptr(dy), ptr(dx), dt(x), rows, Cn, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
              ptr(dgamma), ptr(dbeta), int(relu), int(accumulate_dx), ptr(workspace),
              workspace.numel() * workspace.element_size(), stream())


def bn_backward_fused(x, g, dx, rows, Cn, mean, rstd, gamma, partial, tiles, dgamma, dbeta, accumulate_dx, workspace):
    _lib.call("stp_bn_backward_fused", ptr(x), ptr(g), ptr(dx), dt(x), rows, Cn, ptr(mean), ptr(rstd), ptr(gamma), ptr(partial),
              int(tiles), ptr(dgamma), ptr(dbeta), int(accumulate_dx), ptr(workspace),
              workspace.numel() * workspace.element_size(), stream())


def maxpool3x3s2(x, y, idx, N, H, W, Cn):
    _lib.call("stp_maxpool3x3s2", ptr(x), ptr(y), ptr(idx), N, H, W, Cn, dt(x), stream())


def maxpool3x3s2_bwd(idx, dy, dx, N, H, W, Cn, accumulate=0):
    _lib.call("stp_maxpool3x3s2_bwd", ptr(idx), ptr(dy), ptr(dx), N, H, W, Cn, dt(dy), int(accumulate), stream())


def upsample2x_bwd(dy, dx, N, H, W, Cn, ldy, accumulate=0):
    _lib.call("stp_upsample2x_bwd", ptr(dy), ptr(dx), N, H, W, Cn, ldy, dt(dy), int(accumulate), stream())


def channel_sum(x, rows, Cn, out, accumulate, workspace):
    _lib.call("stp_channel_sum", ptr(x), dt(x), rows, Cn, ptr(out), int(accumulate), ptr(workspace),
              workspace.numel() * workspace.element_size(), stream())


def add_inplace(dst, src, count):
    _lib.call("stp_add_inplace", ptr(dst), ptr(src), count, dt(dst), stream())


def loss_workspace_bytes():
    return int(_lib.load().stp_loss_workspace_bytes())


def sigmoid_bce_dice(logits, target, count, w_bce, w_dice, scalars, dlogits, dl_channels, grad_scale, workspace):
    _lib.call("stp_sigmoid_bce_dice", ptr(logits), ptr(target), count, dt(logits), float(w_bce), float(w_dice),
              ptr(scalars), ptr(dlogits), dl_channels, float(grad_scale), ptr(workspace),
              workspace.numel() * workspace.element_size(), stream())


def sigmoid_loss_ex(logits, target, count, weights5, scalars, dlogits, dl_channels, grad_scale, workspace):
    """weights5 = (binary_crossentropy, dice_loss, iou_loss, jaccard_loss, focal_loss) weights."""
    import ctypes
    w = (ctypes.c_float * 5)(*[float(v) for v in weights5])
    _lib.call("stp_sigmoid_loss_ex", ptr(logits), ptr(target), count, dt(logits), ctypes.addressof(w), ptr(scalars), ptr(dlogits),
              dl_channels, float(grad_scale), ptr(workspace), workspace.numel() * workspace.element_size(), stream())


def sigmoid(logits, probs, count):
    _lib.call("stp_sigmoid", ptr(logits), ptr(probs), count, dt(logits), stream())


def adam(param, grad, m, v, count, lr, b1, b2, eps, state, mask=None, gscale=None, clipvalue=0.0):
    _lib.call("stp_adam", ptr(param), ptr(grad), ptr(m), ptr(v), count, ptr(lr), b1, b2, eps, ptr(state), ptr(mask),
              ptr(gscale), float(clipvalue), stream())


def rmsprop(p, g, acc, n, lr, rho, eps, mask=None, gscale=None, clipvalue=0.0):
    _lib.call("stp_rmsprop", ptr(p), ptr(g), ptr(acc), n, ptr(lr), rho, eps, ptr(mask), ptr(gscale), clipvalue, stream())


def nadam(p, g, m, v, n, lr, b1, b2, eps, schedule_decay, state, fstate, mask=None, gscale=None, clipvalue=0.0):
    _lib.call("stp_nadam", ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(lr), b1, b2, eps, schedule_decay, ptr(state), ptr(fstate),
              ptr(mask), ptr(gscale), clipvalue, stream())


def sgd(param, grad, vel, count, lr, momentum, nesterov, mask=None, gscale=None, clipvalue=0.0):
    _lib.call("stp_sgd", ptr(param), ptr(grad), ptr(vel), count, ptr(lr), momentum, int(nesterov), ptr(mask), ptr(gscale),
              float(clipvalue), stream())


def grad_global_scale(grad, count, clipnorm, base, gscale, workspace):
    _lib.call("stp_grad_global_scale", ptr(grad), count, float(clipnorm), float(base), ptr(gscale), ptr(workspace),
              workspace.numel() * workspace.element_size(), stream())


def augment_u8(img, mask, img_out, mask_out, params, N, Hin, Win, Hout, Wout, Cn, field=None):
    # the C entry point takes raw pointers: a destination smaller than N x Hout x Wout (x Cn) would be written past its end
    if img_out.numel() < N * Hout * Wout * Cn or img.numel() < N * Hin * Win * Cn:
        raise ValueError("augment_u8: image buffers are smaller than N x H x W x C (%d < %d or %d < %d)" % (
            img_out.numel(), N * Hout * Wout * Cn, img.numel(), N * Hin * Win * Cn))
    if mask is not None and mask_out is not None and (mask_out.numel() < N * Hout * Wout or mask.numel() < N * Hin * Win):
        raise ValueError("augment_u8: mask buffers are smaller than N x H x W")
    if field is not None:
        if field.numel() < N * Hout * Wout:
            raise ValueError("augment_u8: the displacement field is smaller than N x Hout x Wout")
        _lib.call("stp_augment_field_u8", ptr(img), ptr(mask), ptr(img_out), ptr(mask_out), ptr(params), ptr(field), N, Hin, Win, Hout,
                  Wout, Cn, stream())
        return
    _lib.call("stp_augment_u8", ptr(img), ptr(mask), ptr(img_out), ptr(mask_out), ptr(params), N, Hin, Win, Hout, Wout, Cn,
              stream())


def background_replace_u8(img, mask, bg, out, N, H, W, Cn, erosion):
    if min(img.numel(), bg.numel(), out.numel()) < N * H * W * Cn or mask.numel() < N * H * W:
        raise ValueError("background_replace_u8: buffers are smaller than N x H x W (x C)")
    _lib.call("stp_background_replace_u8", ptr(img), ptr(mask), ptr(bg), ptr(out), N, H, W, Cn, int(erosion), stream())


def field_piecewise(field, grid, N, H, W, rows, cols):
    if field.numel() < N * H * W or grid.numel() < N * rows * cols * 2:
        raise ValueError("field_piecewise: buffers are smaller than N x H x W / N x rows x cols x 2")
    _lib.call("stp_field_piecewise", ptr(field), ptr(grid), N, H, W, rows, cols, stream())


def field_elastic(field, tmp, params, N, H, W):
    if field.numel() < N * H * W or tmp.numel() < N * H * W:
        raise ValueError("field_elastic: buffers are smaller than N x H x W")
    _lib.call("stp_field_elastic", ptr(field), ptr(tmp), ptr(params), N, H, W, stream())


def filter_u8(src, dst, params, N, H, W, Cn):
    _lib.call("stp_filter_u8", ptr(src), ptr(dst), ptr(params), N, H, W, Cn, stream())


def cast_f32_to_bf16(src, dst, count):
    _lib.call("stp_cast_f32_to_bf16", ptr(src), ptr(dst), count, stream())


def cast_bf16_to_f32(src, dst, count, scale=1.0):
    _lib.call("stp_cast_bf16_to_f32", ptr(src), ptr(dst), count, float(scale), stream())
