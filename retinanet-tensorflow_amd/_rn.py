"""ctypes binding of librn_hip.so (C ABI: include/rn_hip.h).

PyTorch-ROCm tensors are only the device-memory / stream / autograd carrier: every call
below hands raw device pointers and the current HIP stream to a hand-written gfx950
kernel.  There is no CPU or eager-PyTorch fallback -- a missing library, or a tensor that is
not a contiguous fp32 device tensor, raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RN_LIB_PATH") or os.path.join(_HERE, "librn_hip.so")   # (RN_LIB_PATH: A/B measurements of two builds on one box)

MAX_SEG = 16
ACT = {None: 0, "none": 0, "linear": 0, "relu": 1, "elu": 2, "relu6": 3, "sigmoid": 4}
LOSS_MODE = {"bce_dice": 0, "focal": 1}
OPT = {"momentum": 0, "rmsprop": 1, "adam": 2}
OPT_BLOCK = 1024
LOSS_STATS_HEADER = 8
API_VERSION = 407        # RN_API_VERSION of include/rn_hip.h these bindings were written against


class RnError(RuntimeError):
    pass


class ConvSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("wgt", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
                ("dy", C.c_void_p), ("dx", C.c_void_p),
                ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cout", C.c_int32),
                ("x_ld", C.c_int32), ("x_coff", C.c_int32), ("wgt_bytes", C.c_int64)]


class AssignLevel(C.Structure):
    _fields_ = [("anchor_sizes", C.c_void_p), ("grid_h", C.c_int32), ("grid_w", C.c_int32), ("cls_out", C.c_void_p),
                ("reg_out", C.c_void_p), ("trainable_out", C.c_void_p), ("argmax_out", C.c_void_p)]


class ConvGeom(C.Structure):
    _fields_ = [("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("cin", C.c_int32),
                ("groups", C.c_int32)]


class WinoGn(C.Structure):
    _fields_ = [("in_rows", C.c_void_p), ("in_gamma", C.c_void_p), ("in_beta", C.c_void_p), ("in_groups", C.c_int32),
                ("in_act", C.c_int32), ("in_eps", C.c_float), ("out_rows", C.c_void_p), ("out_groups", C.c_int32),
                ("u_ready", C.c_void_p), ("urot_ready", C.c_int32)]


class WinoGnBwd(C.Structure):
    _fields_ = [("in_rows", C.c_void_p), ("in_gamma", C.c_void_p), ("in_beta", C.c_void_p), ("in_groups", C.c_int32),
                ("in_act", C.c_int32), ("in_eps", C.c_float), ("in_g_rows_group", C.c_void_p), ("in_g_rows_chan", C.c_void_p),
                ("out_rows", C.c_void_p), ("out_g_rows_group", C.c_void_p), ("out_gamma", C.c_void_p),
                ("out_groups", C.c_int32), ("out_eps", C.c_float), ("defer_wgrad", C.c_int32)]


class ReduceDesc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("count", C.c_int32), ("nrows", C.c_uint16), ("accumulate", C.c_uint16)]


class ReduceList(C.Structure):
    """rn_reduce_list: deferred row reductions recorded by the gradient entry points, owned by the caller (host memory)."""
    _fields_ = [("desc", C.POINTER(ReduceDesc)), ("capacity", C.c_int32), ("count", C.c_int32)]

    @classmethod
    def make(cls, capacity=1024):
        arr = (ReduceDesc * capacity)()
        lst = cls(C.cast(arr, C.POINTER(ReduceDesc)), capacity, 0)
        lst._keep = arr
        return lst


class AddSeg(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out", C.c_void_p), ("count", C.c_int64)]


class DwGnParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("stride", C.c_int32),
                ("groups", C.c_int32), ("act", C.c_int32), ("eps", C.c_float), ("drop_rate", C.c_float),
                ("drop_seed1", C.c_uint64), ("drop_seed2", C.c_uint64), ("drop_seed_dev", C.c_void_p)]


class GnSeg(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("residual", C.c_void_p), ("dy", C.c_void_p),
                ("dx", C.c_void_p), ("dresidual", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("n", C.c_int32), ("hw", C.c_int32), ("x_ld", C.c_int32), ("dx_ld", C.c_int32), ("dx_accumulate", C.c_int32)]


class GnParams(C.Structure):
    _fields_ = [("c", C.c_int32), ("groups", C.c_int32), ("act", C.c_int32), ("act_after_residual", C.c_int32),
                ("in_f16", C.c_int32), ("out_f16", C.c_int32), ("eps", C.c_float),
                ("drop_rate", C.c_float), ("drop_seed", C.c_uint64), ("drop_seed_dev", C.c_void_p), ("sync", C.c_void_p),
                ("stat_rows", C.c_void_p)]


class GnRows(C.Structure):
    """rn_gn_rows: partial (sum, sum of squares) rows a conv / depthwise forward wrote for the GroupNorm that follows"""
    _fields_ = [("rows", C.c_void_p), ("rows_per_sample", C.c_int32), ("per_group", C.c_int32), ("groups", C.c_int32)]


class GnResidualNorm(C.Structure):
    _fields_ = [("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("groups", C.c_int)]


class F16Fold(C.Structure):
    """rn_f16_fold"""
    _fields_ = [("in_mean", C.c_void_p), ("in_rstd", C.c_void_p), ("in_gamma", C.c_void_p), ("in_beta", C.c_void_p),
                ("in_groups", C.c_int32), ("in_act", C.c_int32), ("partial", C.c_void_p),
                ("seg_chunk_start", C.c_void_p), ("total_chunks", C.c_int32)]


class MbRows(C.Structure):
    """rn_mb_rows"""
    _fields_ = [("rows", C.c_void_p), ("rows_per_sample", C.c_int32), ("width", C.c_int32), ("bn", C.c_int32)]


class MbNorm(C.Structure):
    """rn_mb_norm: one GroupNorm (+ activation + dropout) block, applied by the kernels on either side of it"""
    _fields_ = [("y", C.c_void_p), ("stat", MbRows), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("gamma", C.c_void_p),
                ("beta", C.c_void_p), ("c", C.c_int32), ("groups", C.c_int32), ("act", C.c_int32), ("eps", C.c_float),
                ("drop_rate", C.c_float), ("drop_seed", C.c_uint64), ("drop_seed_dev", C.c_void_p)]


class MbPhase(C.Structure):
    """rn_mb_phase: one phase of the XCD-resident section (rn_mb_resident_fwd)"""
    _fields_ = [("kind", C.c_int32), ("in_", C.POINTER(MbNorm)), ("residual", C.c_void_p), ("materialise", C.c_void_p),
                ("w", C.c_void_p), ("y", C.c_void_p), ("h", C.c_int32), ("wd", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("stride", C.c_int32), ("stat_out", MbRows), ("stat_groups", C.c_int32)]


MB_PHASE_POINTWISE, MB_PHASE_DEPTHWISE = 0, 1


class MbDy(C.Structure):
    """rn_mb_dy"""
    _fields_ = [("dy", C.c_void_p), ("norm", C.POINTER(MbNorm)), ("g", C.c_void_p), ("g_plain", C.c_int32), ("grows", MbRows)]


class MbGout(C.Structure):
    """rn_mb_gout"""
    _fields_ = [("out", C.c_void_p), ("add1", C.c_void_p), ("add2", C.c_void_p), ("norm", C.POINTER(MbNorm)),
                ("store_plain", C.c_int32), ("grows", MbRows), ("planes", C.c_void_p)]


class LossSeg(C.Structure):
    _fields_ = [("cls_logit", C.c_void_p), ("cls_label", C.c_void_p), ("reg_pred", C.c_void_p),
                ("reg_label", C.c_void_p), ("trainable", C.c_void_p), ("d_cls_logit", C.c_void_p),
                ("d_reg_pred", C.c_void_p), ("rows", C.c_int64)]


class DetLevel(C.Structure):
    _fields_ = [("prob", C.c_void_p), ("boxes", C.c_void_p), ("rows_per_image", C.c_int64),
                ("regression", C.c_void_p), ("anchor_sizes", C.c_void_p),
                ("grid_h", C.c_int32), ("grid_w", C.c_int32), ("num_anchors", C.c_int32),
                ("prob_f16", C.c_int32), ("regression_f16", C.c_int32), ("prob_is_logit", C.c_int32)]


class DetParams(C.Structure):
    _fields_ = [("n", C.c_int32), ("num_classes", C.c_int32), ("max_per_class", C.c_int32),
                ("score_threshold", C.c_float), ("iou_threshold", C.c_float), ("max_candidates", C.c_int64)]


_lib = None

# every symbol include/rn_hip.h declares (tests/test_abi.py checks the .so exports them all)
SYMBOLS = [
    "rn_version", "rn_last_error", "rn_same_pad",
    "rn_conv2d_fwd_workspace", "rn_conv2d_dgrad_workspace",
    "rn_zero", "rn_conv2d_fwd", "rn_conv2d_stats_rows", "rn_conv2d_fwd_stats", "rn_conv2d_dropout_rows", "rn_conv2d_fwd_dropout", "rn_depthwise_stats_rows", "rn_depthwise_fwd_stats", "rn_group_norm_rows_ok",
    "rn_conv2d_dgrad", "rn_conv2d_wgrad_workspace", "rn_conv2d_wgrad", "rn_conv2d_bwd",
    "rn_conv2d_bias_grad_workspace", "rn_conv2d_bias_grad", "rn_conv3x3_winograd_workspace", "rn_conv3x3_winograd",
    "rn_conv3x3_winograd_wgrad_workspace", "rn_conv3x3_winograd_wgrad", "rn_conv3x3_winograd_keep_bytes",
    "rn_conv3x3_winograd_bwd_workspace", "rn_conv3x3_winograd_bwd", "rn_flush_reductions", "rn_gemm_batched",
    "rn_winograd_bwd_products_workspace", "rn_winograd_bwd_products", "rn_wino_gn_rows", "rn_conv3x3_winograd_gn", "rn_conv3x3_winograd_gn_bwd", "rn_conv3x3_winograd_gn_bwd_wgrad", "rn_reduce_rows", "rn_resize_bilinear_normalize",
    "rn_dwgn_supported", "rn_dwgn_fwd", "rn_dwgn_bwd", "rn_depthwise_fwd", "rn_depthwise_dgrad", "rn_depthwise_wgrad_workspace", "rn_depthwise_wgrad", "rn_depthwise_bwd",
    "rn_group_norm_sync_bytes", "rn_group_norm_workspace", "rn_group_norm_fwd", "rn_group_norm_bwd",
    "rn_act_fwd", "rn_act_bwd", "rn_upsample_add_fwd", "rn_upsample_add_bwd_top",
    "rn_pack_weights_f16", "rn_pack_weights_f16_bytes", "rn_cast_f32_to_f16", "rn_pad_cast_rgb_f16", "rn_conv2d_fwd_f16", "rn_conv2d_f16_fold_rows", "rn_conv2d_fwd_f16_fold", "rn_group_norm_finalize", "rn_group_norm_apply_f16", "rn_group_norm_apply_res_f16", "rn_maxpool_fwd_f16", "rn_maxpool_gn_fwd_f16", "rn_upsample_add_fwd_f16",
    "rn_act_fwd_f16", "rn_flip_width", "rn_dropout", "rn_dropout_strided", "rn_maxpool_fwd", "rn_maxpool_bwd", "rn_maxpool_bwd_arg", "rn_avgpool_fwd", "rn_avgpool_bwd",
    "rn_loss_workspace", "rn_loss_fwd", "rn_loss_bwd",
    "rn_iou", "rn_anchor_assign", "rn_anchor_assign_levels", "rn_anchor_assign_levels_pair", "rn_decode_boxes", "rn_detect_workspace", "rn_detect",
    "rn_boxes_decode", "rn_nms_classwise_workspace", "rn_nms_classwise",
    "rn_optimizer_workspace", "rn_grad_norm_l2reg", "rn_optimizer_step", "rn_counter_add", "rn_add_segs",
    "rn_mb_rows_max", "rn_mb_compact_rows_layout", "rn_mb_compact_rows", "rn_mb_pointwise_rows", "rn_mb_pointwise_fwd", "rn_mb_depthwise_rows", "rn_mb_depthwise_fwd", "rn_mb_apply",
    "rn_mb_resident_sync_bytes", "rn_mb_resident_rows", "rn_mb_resident_fwd", "rn_set_product_mode", "rn_get_product_mode",
    "rn_set_x3_bfrag", "rn_get_x3_bfrag", "rn_x3_bfrag_ok", "rn_x3_bfrag_bytes", "rn_x3_pack_bfrag", "rn_gemm_batched_bfrag", "rn_conv3x3_winograd_gn_u_bytes", "rn_conv3x3_winograd_gn_weights",
    "rn_conv2d_f16_stats_tiles", "rn_group_norm_fwd_f16_tiles",
    "rn_mb_pointwise_bwd_rows", "rn_mb_pointwise_bwd_workspace", "rn_mb_pointwise_bwd",
    "rn_mb_depthwise_bwd_rows", "rn_mb_depthwise_bwd_workspace", "rn_mb_depthwise_bwd",
    "rn_debug_collective_standin", "rn_optimizer_norm_pairs", "rn_optimizer_step_norm", "rn_norm_reg_finalize",
]


def lib():
    """Load librn_hip.so (built by `make -C retinanet-tensorflow_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RnError("librn_hip.so not found at %s: build it with "
                          "`make -C retinanet-tensorflow_amd/csrc` (there is no fallback path)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        if not hasattr(L, "rn_version") or L.rn_version() != API_VERSION:
            raise RnError("%s reports ABI version %s, these bindings need %d: rebuild it (make -C retinanet-tensorflow_amd/csrc)"
                          % (LIB_PATH, L.rn_version() if hasattr(L, "rn_version") else "none", API_VERSION))
        L.rn_last_error.restype = C.c_char_p
        L.rn_group_norm_sync_bytes.argtypes = []
        for name in ("rn_conv2d_wgrad_workspace", "rn_depthwise_wgrad_workspace", "rn_group_norm_workspace", "rn_group_norm_sync_bytes",
                     "rn_loss_workspace", "rn_detect_workspace", "rn_optimizer_workspace",
                     "rn_conv2d_bias_grad_workspace", "rn_nms_classwise_workspace", "rn_conv3x3_winograd_workspace", "rn_conv2d_fwd_workspace",
                     "rn_conv2d_dgrad_workspace",
                     "rn_conv3x3_winograd_wgrad_workspace", "rn_conv3x3_winograd_bwd_workspace", "rn_wino_gn_rows",
                     "rn_winograd_bwd_products_workspace", "rn_conv2d_stats_rows", "rn_depthwise_stats_rows", "rn_conv2d_dropout_rows",
                     "rn_x3_bfrag_bytes", "rn_conv3x3_winograd_gn_u_bytes"):
            getattr(L, name).restype = C.c_size_t
        L.rn_optimizer_workspace.argtypes = [C.c_int64]
        L.rn_dwgn_supported.argtypes = [C.c_void_p, C.c_int]
        L.rn_dwgn_fwd.argtypes = [C.c_void_p] * 10
        L.rn_dwgn_bwd.argtypes = [C.c_void_p] * 12
        L.rn_depthwise_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_depthwise_dgrad.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_depthwise_bwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_depthwise_wgrad_workspace.argtypes = [C.c_int] * 6
        L.rn_depthwise_wgrad.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_conv2d_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rn_conv2d_dgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.rn_zero.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.rn_optimizer_norm_pairs.argtypes = [C.c_int64]
        L.rn_optimizer_norm_pairs.restype = C.c_int64
        L.rn_optimizer_step_norm.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int64, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_uint64,
                                             C.c_void_p, C.c_void_p]
        L.rn_norm_reg_finalize.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.rn_debug_collective_standin.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]
        L.rn_conv2d_stats_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.rn_conv2d_fwd_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_conv2d_dropout_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rn_conv2d_fwd_dropout.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_depthwise_stats_rows.argtypes = [C.c_int] * 7 + [C.c_void_p]
        L.rn_group_norm_rows_ok.argtypes = [C.c_int] * 4
        L.rn_depthwise_fwd_stats.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
        L.rn_conv2d_fwd_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv2d_dgrad_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv2d_bwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_conv2d_wgrad_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv2d_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_conv2d_bias_grad_workspace.argtypes = [C.c_int]
        L.rn_conv3x3_winograd_workspace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rn_resize_bilinear_normalize.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p,
                                                                                                         C.c_void_p]
        L.rn_gemm_batched.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        L.rn_flush_reductions.argtypes = [C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd_wgrad_workspace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.rn_conv3x3_winograd_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd_bwd_workspace.argtypes = [C.c_void_p] + [C.c_int] * 6
        L.rn_conv3x3_winograd_bwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd_keep_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_wino_gn_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rn_winograd_bwd_products_workspace.argtypes = [C.c_int] * 4
        L.rn_winograd_bwd_products.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p] * 2 + [C.c_int] * 3 + \
                                              [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_x3_bfrag_ok.argtypes = [C.c_int] * 3
        L.rn_conv2d_f16_stats_tiles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.rn_group_norm_fwd_f16_tiles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd_gn_u_bytes.argtypes = [C.c_int] * 3
        L.rn_conv3x3_winograd_gn_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_x3_bfrag_bytes.argtypes = [C.c_int] * 3
        L.rn_x3_pack_bfrag.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        L.rn_gemm_batched_bfrag.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        L.rn_set_x3_bfrag.argtypes = [C.c_int]
        L.rn_conv3x3_winograd_gn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_conv3x3_winograd_gn_bwd_wgrad.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                       C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv3x3_winograd_gn_bwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_reduce_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rn_conv2d_bias_grad.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_void_p]
        L.rn_group_norm_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_group_norm_fwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_size_t, C.c_void_p]
        L.rn_group_norm_bwd.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_act_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.rn_act_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.rn_upsample_add_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_upsample_add_bwd_top.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_pack_weights_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rn_pack_weights_f16_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        L.rn_pack_weights_f16_bytes.restype = C.c_size_t
        L.rn_cast_f32_to_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rn_pad_cast_rgb_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.rn_conv2d_fwd_f16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv2d_f16_fold_rows.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_conv2d_fwd_f16_fold.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_group_norm_finalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_group_norm_apply_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rn_group_norm_apply_res_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.rn_maxpool_fwd_f16.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_maxpool_gn_fwd_f16.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_void_p] * 4 + [C.c_int] * 2 + [C.c_void_p]
        L.rn_upsample_add_fwd_f16.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_act_fwd_f16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.rn_flip_width.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rn_dropout_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 5 + [C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]
        L.rn_dropout.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_void_p, C.c_void_p]
        L.rn_maxpool_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_maxpool_bwd_arg.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_maxpool_bwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_avgpool_fwd.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_avgpool_bwd.argtypes = [C.c_void_p] * 2 + [C.c_int] * 6 + [C.c_void_p]
        L.rn_loss_workspace.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rn_loss_fwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                  C.c_void_p]
        L.rn_loss_bwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]
        L.rn_iou.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rn_anchor_assign.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 4 + \
                                      [C.c_void_p] * 4 + [C.c_void_p]
        L.rn_anchor_assign_levels.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p]
        L.rn_anchor_assign_levels_pair.argtypes = L.rn_anchor_assign_levels.argtypes
        L.rn_decode_boxes.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
        L.rn_detect_workspace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_detect.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 6 + [C.c_void_p, C.c_size_t,
                                                                                     C.c_void_p]
        L.rn_boxes_decode.argtypes = L.rn_detect.argtypes
        L.rn_nms_classwise_workspace.argtypes = [C.c_void_p]
        L.rn_nms_classwise.argtypes = [C.c_void_p] * 5 + [C.c_void_p] + [C.c_void_p] * 6 + [C.c_void_p, C.c_size_t,
                                                                                        C.c_void_p]
        L.rn_grad_norm_l2reg.argtypes = [C.c_void_p] * 3 + [C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                                             C.c_size_t, C.c_void_p]
        L.rn_optimizer_step.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_int64, C.c_float, C.c_float,
                                                                       C.c_float, C.c_void_p, C.c_int64,
                                                                       C.c_void_p, C.c_uint64, C.c_void_p]
        L.rn_counter_add.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.rn_add_segs.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        for name in ("rn_mb_pointwise_rows", "rn_mb_depthwise_rows", "rn_mb_pointwise_bwd_rows", "rn_mb_pointwise_bwd_workspace",
                     "rn_mb_depthwise_bwd_rows", "rn_mb_depthwise_bwd_workspace"):
            getattr(L, name).restype = C.c_size_t
        L.rn_mb_resident_sync_bytes.restype = C.c_size_t
        L.rn_mb_resident_rows.restype = C.c_size_t
        L.rn_mb_resident_rows.argtypes = [C.c_int] * 8 + [C.c_void_p]
        L.rn_mb_resident_fwd.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.rn_mb_compact_rows_layout.restype = C.c_size_t
        L.rn_mb_compact_rows_layout.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.rn_mb_compact_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rn_mb_pointwise_rows.argtypes = [C.c_int] * 5 + [C.c_void_p]
        L.rn_mb_pointwise_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_mb_depthwise_rows.argtypes = [C.c_int] * 6 + [C.c_void_p]
        L.rn_mb_depthwise_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p, C.c_int, C.c_void_p]
        L.rn_mb_apply.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_void_p]
        L.rn_mb_pointwise_bwd_rows.argtypes = [C.c_int] * 5 + [C.c_void_p]
        L.rn_mb_pointwise_bwd_workspace.argtypes = [C.c_int] * 4
        L.rn_mb_pointwise_bwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_mb_depthwise_bwd_rows.argtypes = [C.c_int] * 6 + [C.c_void_p]
        L.rn_mb_depthwise_bwd_workspace.argtypes = [C.c_int] * 5
        L.rn_mb_depthwise_bwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.rn_same_pad.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.rn_same_pad.restype = None
        _lib = L
    return _lib


def check(status, what):
    if status != 0:
        raise RnError("%s failed (status %d): %s" % (what, status, lib().rn_last_error().decode()))


def same_pad(n, k, s):
    """TF SAME rule -> (out, pad_before); host-side mirror of rn_same_pad."""
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


def ptr(t):
    """Raw device pointer of a contiguous fp32/u8/int tensor on the GPU (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RnError("rn_hip kernels need device tensors (got a CPU tensor): there is no CPU fallback")
    if not t.is_contiguous():
        raise RnError("rn_hip kernels need contiguous tensors")
    return t.data_ptr()


def f32(t):
    if t.dtype != torch.float32:
        raise RnError("this rn_hip kernel is fp32 (got %s)" % t.dtype)
    return ptr(t)


def f16(t):
    if t.dtype != torch.float16:
        raise RnError("this rn_hip kernel takes fp16 storage (got %s)" % t.dtype)
    return ptr(t)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# one scratch arena per device, reused by every op (ops are stream-ordered); grown on demand
_workspaces = {}
WORKSPACE_MIN_BYTES = 256 << 20


SIDE_STREAMS = set()   # raw handles of streams that run concurrently with the main one
_side_streams = {}


_sync_words = {}


def sync_counters(device):
    """The zeroed region (rn_group_norm_sync_bytes()) per (device, current stream) for rn_gn_params.sync: counters and
    exchange rows of the grid-resident GroupNorm path.  Word 2 is set if a wait ever timed out."""
    # one region per stream HANDLE (two grid-resident kernels that run concurrently on different streams must not share
    # tags / exchange rows).  Nothing may be allocated (and zero-filled) inside a graph capture: a capturing stream without
    # a region of its own takes the region of key 0 -- the one the warm-up pass before the capture sized -- which it shares
    # with nobody while the capture (and later the replay, on the stream that replays it) is the only user.
    # Keys: a registered side stream has its own region (it runs concurrently with the main one); EVERY other stream -- the
    # default stream, a warm-up stream, the capture stream: whichever is "the main stream" at the moment, never two at once --
    # shares the region of key 0.  (A side stream never donates its region to key 0, and warm-up streams made for a
    # re-capture do not leak a region each.)
    h = stream().value or 0
    capturing = torch.cuda.is_current_stream_capturing() if device.type == 'cuda' else False
    key = (device.type, device.index, h if h in SIDE_STREAMS else 0)
    if key[2] == 0:
        _hand_over(('gn',) + key, h, device)
    t = _sync_words.get(key)
    if t is None:
        if capturing:
            raise RnError("the grid-resident GroupNorm's exchange region must exist before graph capture (run the step once eagerly)")
        t = torch.zeros(lib().rn_group_norm_sync_bytes() // 4, dtype=torch.int32, device=device)
        _sync_words[key] = t
    return t


_resident_words = {}
_region_owner = {}      # region key -> raw handle of the stream that used the shared "main stream" region last


def _hand_over(key, h, device):
    """The regions of key 0 are shared by every stream that is not a registered side stream, on the understanding that only one
    of them runs these kernels at a time.  Enforced here instead of assumed: when a DIFFERENT stream asks for the region, it is
    made to wait for everything the previous user has queued (an event recorded on that stream now), so two such streams --
    say an evaluation stream beside the training stream -- serialise on the region instead of sharing barrier words.  Inside a
    graph capture nothing can be recorded on an outside stream: the capture's owner (train.Trainer) orders it behind the warm-up
    stream itself."""
    last = _region_owner.get(key)
    if last is not None and last != h and device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
        try:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.ExternalStream(last, device=device) if last else torch.cuda.default_stream(device))
            torch.cuda.current_stream(device).wait_event(ev)
        except RuntimeError:
            pass                # (the previous stream no longer exists: nothing left to wait for)
    if device.type != 'cuda' or not torch.cuda.is_current_stream_capturing():
        # (a capture stream is not an owner anyone can wait for later: it may be gone by then, and nothing could be recorded on
        # it from outside anyway -- the previous REAL owner stays on record)
        _region_owner[key] = h


def resident_sync(device):
    """The zeroed region (rn_mb_resident_sync_bytes()) per (device, main / side stream) for rn_mb_resident_fwd: the clusters'
    barrier counters; word 512 is the error word (bit 0: a wait timed out, bit 1: a cluster was not on one XCD).  Same keying
    as sync_counters: every stream that is not a registered side stream is 'the main stream'."""
    h = stream().value or 0
    key = (device.type, device.index, h if h in SIDE_STREAMS else 0)
    if key[2] == 0:
        _hand_over(('mb',) + key, h, device)
    t = _resident_words.get(key)
    if t is None:
        if device.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RnError("the resident section's counters must exist before graph capture (run the step once eagerly)")
        t = torch.zeros(lib().rn_mb_resident_sync_bytes() // 4, dtype=torch.int32, device=device)
        _resident_words[key] = t
    return t


def resident_errors():
    """OR of the resident sections' error words over all regions (0: fine; synchronises)."""
    e = 0
    for t in _resident_words.values():
        e |= int(t[512].item())
    return e


def barrier_timeout_sources():
    """(GroupNorm exchange regions with their error word set, MobileNetV2 resident-section counter sets with theirs set)."""
    return (sum(int(t[2].item()) for t in {id(t): t for t in _sync_words.values()}.values()),
            sum(1 for t in _resident_words.values() if int(t[512].item()) != 0))


def barrier_timeouts():
    """Number of (device, stream) counter sets whose error word is set (should always be 0): the grid-resident GroupNorm's
    exchange regions and the MobileNetV2 resident section's clusters."""
    return (sum(int(t[2].item()) for t in {id(t): t for t in _sync_words.values()}.values()) +
            sum(1 for t in _resident_words.values() if int(t[512].item()) != 0))


def reset_barrier_timeouts():
    """Clear the error words (after the caller has dealt with a reported timeout, e.g. by switching the path off); the
    resident section's counters are zeroed entirely (a timed-out launch leaves them dirty)."""
    for t in {id(t): t for t in _sync_words.values()}.values():
        t[2] = 0
    for t in _resident_words.values():
        t.zero_()


def side_stream(device, index=0):
    """A stream (per device and index) for work that may overlap the main stream; it gets its own workspace."""
    key = (device.type, device.index, index)
    st = _side_streams.get(key)
    if st is None:
        lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, 0)
        st = torch.cuda.Stream(device=device, priority=int(os.environ.get('RN_SIDE_PRIO', lo)))  # lowest priority
        _side_streams[key] = st
        SIDE_STREAMS.add(st.cuda_stream)
    return st


def join_side_streams(device, skip=()):
    """Make the current stream wait for everything queued on this device's side streams (except the indices in `skip`)."""
    cur = torch.cuda.current_stream(device)
    for (t, i, idx), st in _side_streams.items():
        if t == device.type and i == device.index and idx not in skip:
            cur.wait_stream(st)


def workspace(nbytes, device):
    # one arena for the main stream (whichever stream that currently is: default, warm-up or graph-capture
    # stream) and one per registered side stream: ops on a side stream run concurrently with the main one
    h = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
    key = (device.type, device.index, h if h in SIDE_STREAMS else 0)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RnError("workspace must be sized before graph capture (need %d bytes)" % nbytes)
        size = max(int(nbytes), WORKSPACE_MIN_BYTES)
        ws = torch.empty(size, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws
