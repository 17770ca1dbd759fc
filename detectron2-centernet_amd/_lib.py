"""ctypes binding of libctdet_hip.so (the C ABI in include/ctdet_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (the reference raises RuntimeError from TORCH_CHECK/AT_ERROR the same way,
detectron2/layers/csrc/deformable/deform_conv.h:282-311).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libctdet_hip.so")

F16, F32, U8, F16X3 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_SIGMOID_CLAMP = 0, 1, 2


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "H", "W", "Cin", "in_stride", "Cout", "Ho", "Wo", "out_stride",
        "R", "S", "stride", "pad", "dil", "Kpad", "Cout_pad",
        "compute_dtype", "out_dtype", "act", "res_stride")] + [("clamp_lo", C.c_float), ("clamp_hi", C.c_float), ("korder", C.c_int32), ("in_dil", C.c_int32)]


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
# name -> (restype, argtypes); must list every symbol declared in include/ctdet_hip.h
SIGNATURES = {
    "ctdet_last_error": (C.c_char_p, []),
    "ctdet_abi_version": (_i32, []),
    "ctdet_conv_cout_tile": (_i32, [_i32]),
    "ctdet_conv2d_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_conv1x1_cat_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_dcnv2_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_dcnv2_offset_supported": (_i32, [C.POINTER(ConvDesc)]),
    "ctdet_dcnv2_cols_supported": (_i32, [C.POINTER(ConvDesc), _vp, _vp]),
    "ctdet_dcnv2_fwd_cols": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_dcnv2_offset_fwd": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_preprocess": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i64, _vp, _vp, _i32, _i32, _vp]),
    "ctdet_head_fused_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "ctdet_dla_base_fwd": (_i32, [_vp] * 14),
    "ctdet_dla_base_x3_fwd": (_i32, [_vp] * 14),
    "ctdet_maxpool2x2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_maxpool3x3s2": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_maxpool3x3s2_ceil": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_global_avgpool": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctdet_ese_scale": (_i32, [_vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_pack_weights": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_pack_weights_batch": (_i32, [_vp, _i32, _i32, _vp]),
    "ctdet_split_weights": (_i32, [_vp, _vp, _i64, _vp]),
    "ctdet_pack_weights_x3": (_i32, [_vp, _vp, _vp] + [_i32] * 10 + [_vp]),
    "ctdet_pack_weights_x3_batch": (_i32, [_vp, _i32, _i32, _vp]),
    "ctdet_dwconvT_add": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_decode_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32, _i32]),
    "ctdet_decode": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_postprocess": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_decode_status": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_gaussian_targets": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_gaussian_radius": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "ctdet_focal_loss_workspace_bytes": (_sz, [_i64]),
    "ctdet_focal_loss": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    "ctdet_reg_l1_loss": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _vp]),
    "ctdet_chan_workspace_bytes": (_sz, [_i32]),
    "ctdet_bn_train_fwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _i32, _i32, _vp]),
    "ctdet_bn_train_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32,
                                   _vp, _vp, _f32, _vp, _i32, _vp]),
    "ctdet_conv_wgrad": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _f32, _vp]),
    "ctdet_conv_wgrad_oihw": (_i32, [C.POINTER(ConvDesc), _vp, _vp, _vp, _f32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_grad_scatter_oihw": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ctdet_depth_to_space2": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_maxpool3x3s2_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_finite_flag": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "ctdet_ese_dot": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "ctdet_ese_bwd": (_i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_maxpool2x2_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_dwconvT_bwd": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_dcn_cols": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_dcn_col2im_coord": (_i32, [_vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_dcn_col2im_fused": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "ctdet_sgd_momentum": (_i32, [_vp, _vp, _vp, _i64, _vp, _f32, _f32, _i32, _vp]),
    "ctdet_sgd_momentum_runs": (_i32, [_vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _i32, _f32, _i32, _vp]),
    "ctdet_set_tuning_flags": (_i32, [C.c_uint32]),
    "ctdet_get_tuning_flags": (C.c_uint32, []),
    "ctdet_comm_unique_id": (_i32, [_vp]),
    "ctdet_comm_init": (_i32, [_vp, _i32, _i32, C.POINTER(C.c_void_p)]),
    "ctdet_allreduce_bucket": (_i32, [_vp, _vp, _i64, _vp]),
    "ctdet_bcast": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "ctdet_comm_destroy": (_i32, [_vp]),
}

_lib = None


def lib():
    """Loads the shared library once; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C detectron2-centernet_amd/csrc`). There is no CPU fallback for the HIP path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = l
        if os.environ.get("CTDET_TUNE_FLAGS"):      # kernel-selection bits for a whole process (A/B runs of bench.py's ranks)
            l.ctdet_set_tuning_flags(int(os.environ["CTDET_TUNE_FLAGS"], 0))
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().ctdet_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


class PackDesc(C.Structure):
    """mirrors ctdet_pack_desc"""
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p)] + [(n, C.c_int32) for n in (
        "O", "I", "R", "S", "chans_pad", "rows_pad", "Kpad", "korder", "transposed", "blk0")]


class Pack3Desc(C.Structure):
    """mirrors ctdet_pack3_desc"""
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("scale_out", C.c_void_p)] + [(n, C.c_int32) for n in (
        "O", "I", "R", "S", "chans_pad", "rows_pad", "Kpad", "layout", "transposed", "scale_n", "blk0", "pad_")]


class DlaBaseDesc(C.Structure):
    """mirrors ctdet_dla_base_desc"""
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Hp", C.c_int32), ("Wp", C.c_int32),
                ("img_dtype", C.c_int32), ("img_batch_stride", C.c_int64), ("mean", C.c_float * 3), ("std", C.c_float * 3),
                ("out_stride", C.c_int32), ("pool_stride", C.c_int32)]


class HeadDesc(C.Structure):
    """mirrors ctdet_head_desc"""
    _fields_ = [("nheads", C.c_int32), ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("in_stride", C.c_int32), ("w2", C.c_void_p * 4), ("b2", C.c_void_p * 4), ("y", C.c_void_p * 4),
                ("y_stride", C.c_int32 * 4), ("cout", C.c_int32 * 4), ("act", C.c_int32 * 4),
                ("clamp_lo", C.c_float), ("clamp_hi", C.c_float)]


TUNE_NO_HALO, TUNE_NO_WIN, TUNE_DCN_MIXED, TUNE_NO_WGRAD_WINDOW, TUNE_NO_COL2IM_WINDOW, TUNE_NO_F32_DCN_WINDOW, TUNE_DCN_WINDOW_V1, TUNE_NO_SMALL_GRID_TILES = 1, 2, 4, 8, 16, 32, 64, 128
TUNE_DCN_SPLIT_4W = 256
TUNE_NO_HALO_TAP2 = 512
TUNE_PAIR2_128, TUNE_DCN_SPLIT_8W64, TUNE_TARGETS_MEMSET = 1024, 2048, 4096


class tuning:
    """`with _lib.tuning(_lib.TUNE_DCN_MIXED): ...` -- kernel-selection switches of the library for the block (tests, tools)"""

    def __init__(self, flags):
        self.flags = flags

    def __enter__(self):
        self.prev = lib().ctdet_get_tuning_flags()
        lib().ctdet_set_tuning_flags(self.prev | self.flags)
        return self

    def __exit__(self, *exc):
        lib().ctdet_set_tuning_flags(self.prev)
        return False
