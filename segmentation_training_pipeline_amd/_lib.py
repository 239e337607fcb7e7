"""ctypes binding of libstp_hip.so (C-ABI declared in include/stp_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call fails this
raises.  PyTorch is used by callers only for device memory, streams and process groups;
pointers cross the boundary as plain integers (``tensor.data_ptr()``).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STP_LIB") or os.path.join(HERE, "libstp_hip.so")   # STP_LIB: another build of the same library (A/B and what-if runs)

LIB_F16_PATH = os.environ.get("STP_LIB_F16") or os.path.join(HERE, "libstp_hip_f16.so")   # the IEEE-half build of the same sources

F32, BF16, U8, F16 = 0, 1, 2, 3
SRC_DIRECT, SRC_NEAREST2X, SRC_ZEROINS2X = 0, 1, 2

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class StpError(RuntimeError):
    pass


class ConvParams(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("weight", vp), ("bias", vp), ("residual", vp), ("dst0", vp), ("dst1", vp),
                ("N", i32), ("Hs0", i32), ("Ws0", i32), ("Hv", i32), ("Wv", i32), ("C0", i32), ("C1", i32),
                ("src0_mode", i32), ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
                ("Ho", i32), ("Wo", i32), ("Cout", i32), ("Cd0", i32),
                ("accumulate0", i32), ("accumulate1", i32), ("relu", i32), ("dtype", i32), ("tile", i32),
                ("stats_tiles", i32), ("stats_partial", vp),
                ("bnb_x", vp), ("bnb_mean", vp), ("bnb_rstd", vp), ("bnb_gamma", vp), ("bnb_beta", vp), ("bnb_relu", i32),
                ("dst_sum2x2", i32), ("stats_slots", i32),
                ("src_bn_mean", vp), ("src_bn_rstd", vp), ("src_bn_gamma", vp), ("src_bn_beta", vp), ("src_bn_relu", i32),
                ("weight_up", vp), ("fold_src", vp), ("fold_weight", vp), ("fold_C", i32),
                ("stats_group_out", vp), ("stats_group_counters", vp), ("stats_group", i32), ("s2d_dgrad", i32)]


class WgradParams(C.Structure):
    _fields_ = [("src0", vp), ("src1", vp), ("dy", vp), ("dw", vp),
                ("N", i32), ("Hs0", i32), ("Ws0", i32), ("Hv", i32), ("Wv", i32), ("C0", i32), ("C1", i32),
                ("src0_mode", i32), ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
                ("Ho", i32), ("Wo", i32), ("Cout", i32), ("accumulate", i32), ("dtype", i32), ("splits", i32),
                ("src_bn_mean", vp), ("src_bn_rstd", vp), ("src_bn_gamma", vp), ("src_bn_beta", vp), ("src_bn_relu", i32)]


# name -> (restype, argtypes); every symbol declared in include/stp_hip.h
SIGNATURES = {
    "stp_abi_version": (i32, []),
    "stp_storage_dtype": (i32, []),
    "stp_conv2d": (i32, [C.POINTER(ConvParams), vp]),
    "stp_weight_prepare_upcollapse": (i32, [vp, vp, i32, i32, i32, i32, vp]),
    "stp_weight_prepare_upcollapse_desc_bytes": (sz, []),
    "stp_weight_prepare_upcollapse_batched": (i32, [vp, i32, i32, vp]),
    "stp_weight_prepare_upcollapse_bwd_batched": (i32, [vp, i32, i32, vp]),
    "stp_conv2d_fold_ok": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_stats_floats": (sz, [C.POINTER(ConvParams)]),
    "stp_conv2d_stats_group_for": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_stats_group_counters": (sz, [C.POINTER(ConvParams), i32]),
    "stp_conv2d_tile_for": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_sc_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_sc": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_sc_stats_tiles": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_scw": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_scw_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_scw_stats_tiles": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_scn": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_scn_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_scn_stats_tiles": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_s64": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_s64_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_s64_stats_tiles": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_pw_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_pw_cols": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_pw": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_stem_eligible": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_stem": (i32, [C.POINTER(ConvParams), vp]),
    "stp_conv2d_halo_variant": (i32, [C.POINTER(ConvParams)]),
    "stp_conv2d_halo_tiles": (i32, [C.POINTER(ConvParams), i32]),
    "stp_conv2d_halo": (i32, [C.POINTER(ConvParams), i32, vp]),
    "stp_conv2d_wgrad_workspace_bytes": (sz, [C.POINTER(WgradParams)]),
    "stp_conv2d_wgrad": (i32, [C.POINTER(WgradParams), vp, sz, vp]),
    "stp_conv2d_wgrad_partial": (i32, [C.POINTER(WgradParams), vp, sz, i32, vp]),
    "stp_conv2d_wgrad_reduce": (i32, [C.POINTER(WgradParams), vp, i32, vp]),
    "stp_wgrad_reduce_desc_bytes": (sz, []),
    "stp_wgrad_reduce_desc_fill": (i64, [vp, i32, C.POINTER(WgradParams), vp]),
    "stp_wgrad_reduce_batched": (i32, [vp, i32, i64, vp]),
    "stp_wgrad_sc_eligible": (i32, [C.POINTER(WgradParams)]),
    "stp_wgrad_sc_slabs": (i32, [C.POINTER(WgradParams)]),
    "stp_wgrad_sc_partial": (i32, [C.POINTER(WgradParams), vp, vp]),
    "stp_conv2d_wgrad_kernel_id": (i32, [C.POINTER(WgradParams)]),
    "stp_wgrad_group_class": (i32, [C.POINTER(WgradParams)]),
    "stp_wgrad_group_table_bytes": (sz, [vp, i32]),
    "stp_wgrad_group_workspace_bytes": (sz, [vp, i32]),
    "stp_wgrad_group_build": (i32, [vp, i32, vp, sz]),
    "stp_wgrad_group_partial": (i32, [vp, vp, vp, sz, vp]),
    "stp_wgrad_group_reduce": (i32, [vp, vp, vp, vp]),
    "stp_weight_prepare": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_weight_prepare_desc_bytes": (sz, []),
    "stp_weight_prepare_desc_fill": (i64, [vp, i32, i64, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32]),
    "stp_weight_prepare_batched": (i32, [vp, i32, i64, i32, vp]),
    "stp_weight_grad_unpad": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_stem_beta_grad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_bn_workspace_bytes": (sz, [i32]),
    "stp_bn_stats": (i32, [vp, i32, i64, i32, f32, f32, vp, vp, vp, vp, vp, sz, vp]),
    "stp_bn_finalize": (i32, [vp, i32, i64, i32, f32, f32, vp, vp, vp, vp, vp]),
    "stp_bn_apply": (i32, [vp, i32, vp, i32, i64, i32, i32, vp, vp, vp, vp, i32, f32, vp]),
    "stp_bn_inference": (i32, [vp, i32, vp, i32, i64, i32, i32, vp, vp, f32, vp, vp, i32, f32, vp]),
    "stp_bn_backward": (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, vp, vp, i32, i32, vp, sz, vp]),
    "stp_bn_apply_slots": (i32, [vp, vp, i32, i64, i32, vp, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "stp_bn_backward_slots": (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp]),
    "stp_zero_bytes": (i32, [vp, i64, vp]),
    "stp_bn_backward_fused": (i32, [vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, sz, vp]),
    "stp_bn_finalize_apply_ok": (i32, [i32, i64, i32, i32]),
    "stp_bn_finalize_apply": (i32, [vp, i32, vp, vp, i32, i64, i32, f32, f32, vp, vp, vp, vp, vp, vp, i32, vp]),
    "stp_bn_backward_fused_add": (i32, [vp, vp, vp, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, i32, vp, sz, vp]),
    "stp_maxpool3x3s2": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "stp_maxpool3x3s2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_bn_apply_maxpool3x3s2": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp]),
    "stp_maxpool3x3s2_bwd_bn_tiles": (i32, [i32, i32, i32, i32, i32]),
    "stp_maxpool3x3s2_bwd_bn": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    "stp_maxpool2x2": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "stp_maxpool2x2_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_avgpool_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "stp_avgpool": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "stp_avgpool_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_avgpool_pyramid_ok": (i32, [i32, i32, i32, i32, i32, i32, i32, i32, i32]),
    "stp_avgpool_pyramid_workspace_bytes": (sz, [i32, i32, i32, i32, i32, i32]),
    "stp_avgpool_pyramid": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "stp_avgpool_pyramid_bwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_maxpool_k": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_maxpool_k_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_relu_bwd": (i32, [vp, vp, i64, i32, vp]),
    "stp_upsample2x_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_tapsum_fwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_tapsum_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_scatter2x_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_scatter2x_bwd_bn_tiles": (i32, [i32, i32, i32, i32, i32]),
    "stp_scatter2x_bwd_bn": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    "stp_upsample2x_bwd_bn_tiles": (i32, [i32, i32, i32, i32, i32, i32]),
    "stp_upsample2x_bwd_bn": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp]),
    "stp_upsample2x_add": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "stp_resize_bilinear": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_upsample_sum": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, i32, vp]),
    "stp_copy_cols_f32": (i32, [vp, i32, vp, i32, i32, i32, i32, vp]),
    "stp_resize_bilinear_bwd_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
    "stp_resize_bilinear_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "stp_resize_nearest": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_resize_nearest_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_channel_sum": (i32, [vp, i32, i64, i32, vp, i32, vp, sz, vp]),
    "stp_add_inplace": (i32, [vp, vp, i64, i32, vp]),
    "stp_loss_workspace_bytes": (sz, []),
    "stp_sigmoid_bce_dice": (i32, [vp, vp, i64, i32, f32, f32, vp, vp, i32, f32, vp, sz, vp]),
    "stp_sigmoid_loss_ex": (i32, [vp, vp, i64, i32, vp, vp, vp, i32, f32, vp, sz, vp]),
    "stp_sigmoid_loss_bias_grad": (i32, [vp, i64, vp, i32, vp]),
    "stp_lovasz_workspace_bytes": (sz, [i64, i32]),
    "stp_lovasz_hinge": (i32, [vp, vp, i32, i64, i32, f32, vp, vp, i32, vp, sz, vp]),
    "stp_sigmoid": (i32, [vp, vp, i64, i32, vp]),
    "stp_softmax_cce_dice": (i32, [vp, vp, i64, i32, i32, i32, f32, f32, vp, vp, i32, f32, vp, sz, vp]),
    "stp_softmax_cce_dice_up_ok": (i32, [i32, i32, i32]),
    "stp_softmax_cce_dice_up_corner_bytes": (sz, [i32, i32, i32, i32]),
    "stp_softmax_cce_dice_up": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp, vp, i32, f32, vp, vp, vp, sz, vp, sz, vp]),
    "stp_softmax": (i32, [vp, vp, i64, i32, i32, i32, vp]),
    "stp_adam": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, vp, vp, vp, f32, vp]),
    "stp_sgd": (i32, [vp, vp, vp, i64, vp, f32, i32, vp, vp, f32, vp]),
    "stp_rmsprop": (i32, [vp, vp, vp, i64, vp, f32, f32, vp, vp, f32, vp]),
    "stp_nadam": (i32, [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, vp, vp, vp, f32, vp]),
    "stp_grad_global_scale": (i32, [vp, i64, f32, f32, vp, vp, sz, vp]),
    "stp_scale_by_device": (i32, [vp, i64, i32, vp, vp, vp]),
    "stp_grad_global_scale_dls": (i32, [vp, i64, f32, f32, vp, vp, vp, sz, vp]),
    "stp_dwconv": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_dwconv_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_dwconv_wgrad_workspace_bytes": (sz, [i32, i32]),
    "stp_dwconv_wgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "stp_resize_bilinear_ac": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_resize_bilinear_ac_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "stp_counter_tick": (i32, [vp, vp]),
    "stp_dropout": (i32, [vp, vp, i64, f32, vp, C.c_uint32, i32, vp]),
    "stp_dropout_spatial": (i32, [vp, vp, i32, i64, i32, f32, vp, C.c_uint32, i32, vp]),
    "stp_sigmoid_act": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "stp_sigmoid_act_bwd": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "stp_prob_bce_dice": (i32, [vp, vp, i64, i32, f32, f32, vp, vp, i32, vp, sz, vp]),
    "stp_softmax_act": (i32, [vp, vp, i64, i32, i32, i32, i32, vp]),
    "stp_softmax_act_bwd": (i32, [vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "stp_prob_cce_dice": (i32, [vp, vp, i64, i32, i32, i32, f32, f32, vp, vp, i32, vp, sz, vp]),
    "stp_augment_u8": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_filter_u8": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "stp_augment_field_u8": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "stp_field_piecewise": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "stp_field_elastic": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "stp_background_replace_u8": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "stp_cast_f32_to_bf16": (i32, [vp, vp, i64, vp]),
    "stp_cast_bf16_to_f32": (i32, [vp, vp, i64, f32, vp]),
    "stp_calib_mfma_flops": (i64, [i32, i32]),
    "stp_calib_mfma": (i32, [vp, i32, i32, vp]),
    "stp_calib_copy": (i32, [vp, vp, i64, vp]),
}

_libs = {}          # storage format ("bf16" | "fp16") -> loaded library
_active = "bf16"    # the library immediate calls (call()) go to; plans hold their own library


def load(storage=None):
    """Loads the shared library serving the 16-bit storage format ``storage`` ("bf16": libstp_hip.so, also the fp32 mode's;
    "fp16": libstp_hip_f16.so; None: the active one).  Building is ``segmentation_training_pipeline_amd.build``.
    Raises StpError if it is absent: there is no fallback path."""
    storage = storage or _active
    if storage in _libs:
        return _libs[storage]
    if storage not in ("bf16", "fp16"):
        raise ValueError("storage must be 'bf16' or 'fp16'")
    path = LIB_PATH if storage == "bf16" else LIB_F16_PATH
    if not os.path.exists(path):
        raise StpError("%s is missing (%s): build it with "
                       "`python -m segmentation_training_pipeline_amd.build`; there is no CPU fallback" % (os.path.basename(path), path))
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.stp_storage_dtype() != (BF16 if storage == "bf16" else F16):
        raise StpError("%s was not built for %s storage" % (path, storage))
    _libs[storage] = lib
    return lib


class storage(object):
    """``with _lib.storage("fp16"):`` - immediate calls (ops.*) inside the block go to that build's library."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _active
        load(self.name)
        self.prev, _active = _active, self.name
        return self

    def __exit__(self, *exc):
        global _active
        _active = self.prev
        return False


_ERR = {-1: "STP_E_BADARG", -2: "STP_E_LAUNCH", -3: "STP_E_WORKSPACE"}


def check(rc, what=""):
    if rc != 0:
        raise StpError("%s failed: %s (%d)" % (what or "stp call", _ERR.get(rc, "?"), rc))


def call(name, *args):
    """Immediate call with error check."""
    check(getattr(load(), name)(*args), name)
