"""ctypes binding of libdgv2.so -- the C ABI declared in include/dgv2.h.

This is the only place the host side touches the native library.  There is no
fallback: if the shared object is missing the import fails loudly, and every
entry point raises RuntimeError on a non-zero return code (the reference's
TORCH_CHECK -> c10::Error -> RuntimeError convention,
gans/models/ops/fused_act/fused_bias_act.cpp:10-16).

Tensors are passed as raw device pointers; the launch goes to torch's CURRENT HIP
stream of the calling thread, so the calls are captured by torch.cuda.graph like any
ATen op (outputs are allocated by the caller from torch's caching allocator).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGV2_LIB_PATH: another build of the SAME ABI (A/B benchmarking of kernel variants); there is still no fallback
LIB_PATH = os.environ.get("DGV2_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libdgv2.so")

F32, BF16 = 0, 1
ABI_VERSION = 47

_c_int, _c_i64, _c_f32, _c_ptr = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> argument ctypes (return type is always int); mirrors include/dgv2.h one to one
SIGNATURES = {
    "dgv2_abi_version": [],
    "dgv2_fused_bias_act": [_c_ptr] * 4 + [_c_i64] * 3 + [_c_int, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_bias_grad": [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr],
    "dgv2_bias_act_bwd": [_c_ptr] * 4 + [_c_i64, _c_int, _c_f32, _c_f32, _c_ptr, _c_i64, _c_int, _c_ptr],
    "dgv2_bias_act_bwd_rs": [_c_ptr] * 4 + [_c_i64, _c_int, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr],
    "dgv2_surface_normal": [_c_ptr, _c_ptr] + [_c_int] * 5 + [_c_ptr],
    "dgv2_mbstd_cat_fwd": [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr],
    "dgv2_mbstd_cat_bwd": [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr],
    "dgv2_gemm_x3": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_int, _c_int, _c_i64, _c_i64, _c_i64,
                     _c_int, _c_f32, _c_ptr],
    "dgv2_pe_wgrad_scratch": [_c_ptr] + [_c_int] * 4,
    "dgv2_pe_wgrad": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 4 + [_c_i64, _c_int, _c_ptr],
    "dgv2_mbstd_cat_fwd_x": [_c_ptr] * 3 + [_c_int] * 8 + [_c_ptr],
    "dgv2_mbstd_cat_bwd_x": [_c_ptr] * 3 + [_c_int] * 8 + [_c_ptr],
    "dgv2_scale_cast": [_c_ptr] * 3 + [_c_i64, _c_int, _c_int, _c_int, _c_ptr],
    "dgv2_upfirdn2d": [_c_ptr] * 3 + [_c_int] * 15 + [_c_ptr],
    "dgv2_resample": [_c_ptr] * 4 + [_c_int] * 19 + [_c_ptr],
    "dgv2_resample_tab": [_c_ptr] * 5 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 10 + [_c_ptr],
    "dgv2_resample_tab_add": [_c_ptr] * 6 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 8 + [_c_ptr],
    "dgv2_resample_tab_add_affine": [_c_ptr] * 8 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 8 + [_c_ptr],
    "dgv2_fourier_feature": [_c_ptr] * 5 + [_c_int] * 8 + [_c_ptr],
    "dgv2_downsample_angle": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr],
    "dgv2_bmm_nn": [_c_ptr] * 3 + [_c_int] * 6 + [_c_i64, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_int, _c_ptr],
    "dgv2_bmm_tn": [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr],
    "dgv2_bmm_nn_cat": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_int, _c_ptr],
    "dgv2_bmm_nn_cat_sq": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_int, _c_ptr, _c_int, _c_ptr,
                           _c_ptr],
    "dgv2_bmm_nn_sq": [_c_ptr] * 3 + [_c_int] * 6 + [_c_i64, _c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_ptr, _c_int, _c_int, _c_ptr, _c_int,
                       _c_ptr, _c_ptr],
    "dgv2_modconv_pe_fwd": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_modconv_pe_fwd_sq": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr, _c_int, _c_ptr,
                               _c_ptr],
    "dgv2_modconv_pe_fwd_head": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr, _c_int, _c_ptr,
                                 _c_ptr, _c_ptr, _c_ptr],
    "dgv2_kitti_rows": [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_ptr],
    "dgv2_kitti_project": [_c_ptr] * 4 + [_c_int] * 4 + [_c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_fps_scratch": [_c_ptr, _c_int, _c_int],
    "dgv2_fps": [_c_ptr] * 3 + [_c_int] * 3 + [_c_ptr],
    "dgv2_gather_points": [_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr],
    "dgv2_gather_points_grad": [_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr],
    "dgv2_chamfer_fwd": [_c_ptr] * 6 + [_c_int] * 3 + [_c_ptr],
    "dgv2_nn_search": [_c_ptr] * 4 + [_c_int] * 4 + [_c_ptr],
    "dgv2_chamfer_bwd": [_c_ptr] * 8 + [_c_int] * 3 + [_c_ptr],
    "dgv2_emd_approxmatch": [_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr],
    "dgv2_emd_matchcost": [_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr],
    "dgv2_emd_matchcost_grad": [_c_ptr] * 5 + [_c_int] * 3 + [_c_ptr],
    "dgv2_nsgan_loss": [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_f32, _c_ptr, _c_ptr, _c_ptr],
    "dgv2_modconv_pe_dgrad_actbwd": [_c_ptr, _c_ptr, _c_ptr, _c_i64] + [_c_ptr] * 5 + [_c_f32, _c_f32] + [_c_int] * 4 + [_c_ptr],
    "dgv2_colmean_lerp": [_c_ptr, _c_ptr, _c_int, _c_int, _c_i64, _c_f32, _c_ptr],
    "dgv2_d_tail_fwd": [_c_ptr] * 6 + [_c_int, _c_int] + [_c_f32] * 4 + [_c_ptr],
    "dgv2_d_tail_bwd": [_c_ptr] * 7 + [_c_int, _c_int] + [_c_f32] * 4 + [_c_ptr],
    "dgv2_rng_fill": [_c_ptr] * 5 + [_c_int, _c_ptr, _c_ptr],
    "dgv2_modconv_up_fwd": [_c_ptr] * 4 + [_c_int] * 7 + [_c_ptr] * 6 + [_c_int, _c_f32, _c_f32, _c_int, _c_ptr, _c_int, _c_ptr,
                            _c_ptr],
    "dgv2_modconv_up_t_lag": [_c_ptr] * 4 + [_c_f32] + [_c_ptr] * 4 + [_c_int] * 9 + [_c_ptr, _c_int, _c_ptr, _c_ptr],
    "dgv2_fir_same_mfma_q8": [_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr],
    "dgv2_resample_tab_q8": [_c_ptr] * 5 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 7 + [_c_ptr],
    "dgv2_fp8_quant_weights": [_c_ptr] * 6 + [_c_int] + [_c_ptr] * 3,
    "dgv2_fp8_dequant": [_c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr],
    "dgv2_conv_taps_fp8": [_c_ptr] * 4 + [_c_int] * 14 + [_c_ptr, _c_int, _c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_ptr],
    "dgv2_modconv_up_t": [_c_ptr] * 5 + [_c_f32] + [_c_int] * 9 + [_c_ptr],
    "dgv2_up2_lag_sumsq": [_c_ptr] * 5 + [_c_int] * 5 + [_c_ptr, _c_int, _c_ptr, _c_ptr],
    "dgv2_resample_tab_actbwd": [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr] + [_c_ptr] * 3 + [_c_int] + [_c_ptr] * 3
                                + [_c_int] * 7 + [_c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_fir_same_mfma_prep": [_c_ptr, _c_i64, _c_ptr] + [_c_ptr] * 3 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 3 + [_c_ptr, _c_ptr],
    "dgv2_fir_same_mfma": [_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr],
    "dgv2_fir_same_mfma_actbwd": [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_int] * 4
                                 + [_c_f32, _c_f32, _c_ptr],
    "dgv2_resample_tab_sq": [_c_ptr] * 5 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 10 + [_c_ptr, _c_int, _c_ptr, _c_ptr],
    "dgv2_bmm_tn_cat": [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr],
    "dgv2_lerp_list": [_c_ptr] * 3 + [_c_int, _c_f32, _c_ptr],
    "dgv2_adam_prep": [_c_ptr, _c_ptr, _c_f32, _c_f32, _c_ptr],
    "dgv2_adam_step": [_c_ptr] * 5 + [_c_int, _c_ptr] + [_c_f32] * 4 + [_c_ptr],
    "dgv2_pack2d": [_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr],
    "dgv2_unpack2d": [_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr],
    "dgv2_ema_scalar": [_c_ptr] * 3 + [_c_int] + [_c_f32] * 3 + [_c_int, _c_ptr, _c_int, _c_ptr],
    "dgv2_ema_scalar_group": [_c_ptr, _c_ptr, _c_int, _c_ptr, _c_int] + [_c_f32] * 3 + [_c_int, _c_ptr, _c_ptr],
    "dgv2_bmm_nn_small": [_c_ptr] * 4 + [_c_int] * 5 + [_c_ptr],
    "dgv2_bmm_nn_small_act": [_c_ptr] * 4 + [_c_int] * 4 + [_c_ptr, _c_ptr, _c_f32, _c_f32, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_int,
                              _c_ptr],
    "dgv2_bmm_tn_small": [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr],
    "dgv2_head_bwd": [_c_ptr] * 5 + [_c_i64] + [_c_ptr] * 7 + [_c_f32, _c_f32] + [_c_int] * 5 + [_c_ptr],
    "dgv2_transpose_list": [_c_ptr] * 5 + [_c_int] * 3 + [_c_ptr],
    "dgv2_mod_prep_all_fwd": [_c_ptr] * 13 + [_c_ptr, _c_int, _c_int, _c_ptr],
    "dgv2_mod_prep_all_bwd": [_c_ptr, _c_i64] + [_c_ptr] * 15 + [_c_ptr, _c_int, _c_int, _c_ptr, _c_i64, _c_ptr],
    "dgv2_mod_prep_all_bwd_scratch": [_c_ptr, _c_ptr, _c_ptr, _c_int, _c_int],
    "dgv2_mod_prep_fwd": [_c_ptr] * 8 + [_c_int] * 9 + [_c_ptr],
    "dgv2_mod_prep_bwd": [_c_ptr] * 11 + [_c_int] * 9 + [_c_ptr],
    "dgv2_sum_squares": [_c_ptr, _c_ptr, _c_i64, _c_int, _c_int, _c_int, _c_ptr],
    "dgv2_conv_fwd": [_c_ptr] * 3 + [_c_int] * 10 + [_c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_conv_taps": [_c_ptr] * 3 + [_c_int] * 17 + [_c_ptr] + [_c_int] * 3 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int,
                                                                               _c_ptr],
    "dgv2_conv_taps_ex": [_c_ptr] * 3 + [_c_int] * 14 + [_c_ptr, _c_int, _c_int, _c_ptr, _c_int, _c_ptr] + [_c_int] * 3
    + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_conv_taps_ld": [_c_ptr, _c_int, _c_ptr, _c_ptr] + [_c_int] * 14 + [_c_ptr, _c_int, _c_int, _c_ptr, _c_int, _c_ptr] + [_c_int] * 3
    + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_conv_wgrad_direct": [_c_ptr] * 3 + [_c_int] * 10 + [_c_ptr],
    "dgv2_bmm_tn_stream_scratch": [_c_ptr] + [_c_int] * 6,
    "dgv2_bmm_tn_stream": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 6 + [_c_ptr],
    "dgv2_bmm_tn_stream_x": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 7 + [_c_ptr],
    "dgv2_bmm_tn_stream_ld": [_c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 7 + [_c_ptr],
    "dgv2_conv_weight_bank": [_c_ptr] * 8 + [_c_int, _c_int, _c_ptr],
    "dgv2_conv_weight_bank_ex": [_c_ptr] * 10 + [_c_int, _c_int, _c_ptr],
    "dgv2_glin_fwd": [_c_ptr] * 6 + [_c_int] * 3 + [_c_f32, _c_f32, _c_int, _c_f32, _c_int, _c_ptr, _c_ptr],
    "dgv2_glin_dinput": [_c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_int, _c_int, _c_f32, _c_f32,
                         _c_int, _c_ptr],
    "dgv2_glin_dweight": [_c_ptr] * 7 + [_c_int] * 3 + [_c_f32, _c_f32, _c_f32, _c_ptr, _c_ptr],
    "dgv2_conv3x3_dgrad8": [_c_ptr] * 3 + [_c_int] * 5 + [_c_ptr, _c_int, _c_ptr],
    "dgv2_conv3x3_s2_dgrad8": [_c_ptr] * 3 + [_c_int] * 6 + [_c_ptr],
    "dgv2_conv3x3_x3_wgrad_scratch": [_c_ptr] + [_c_int] * 6,
    "dgv2_conv3x3_x3_wgrad": [_c_ptr, _c_ptr, _c_i64] + [_c_ptr] * 2 + [_c_int] * 7 + [_c_f32, _c_int, _c_ptr, _c_ptr],
    "dgv2_conv_x3_images": [_c_ptr] * 3 + [_c_int] * 2 + [_c_ptr],
    "dgv2_conv3x3_x3_dgrad": [_c_ptr] * 4 + [_c_int] * 6 + [_c_ptr, _c_ptr],
    "dgv2_conv3x3_x3_fwd": [_c_ptr] * 3 + [_c_int] * 6 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_ptr, _c_ptr],
    "dgv2_conv3x3_fwd8": [_c_ptr] * 3 + [_c_int] * 6 + [_c_ptr, _c_ptr, _c_int, _c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_conv_wgrad_stream_scratch": [_c_ptr] + [_c_int] * 9,
    "dgv2_conv_wgrad_stream": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 10 + [_c_ptr],
    "dgv2_conv_wgrad_stream_pl": [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr] + [_c_int] * 9 + [_c_f32, _c_int, _c_int, _c_ptr],
    "dgv2_conv_dgrad": [_c_ptr] * 4 + [_c_int] * 11 + [_c_ptr],
    "dgv2_conv_wgrad": [_c_ptr] * 3 + [_c_int] * 11 + [_c_ptr],
    "dgv2_stem_fwd": [_c_ptr] * 4 + [_c_int] * 5 + [_c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_stem_bwd_scratch": [_c_ptr] + [_c_int] * 4,
    "dgv2_stem_bwd": [_c_ptr] * 4 + [_c_i64] + [_c_ptr] * 4 + [_c_int] * 5 + [_c_f32, _c_f32, _c_int, _c_ptr],
    "dgv2_stem_bwd_skip": [_c_ptr] * 4 + [_c_i64] + [_c_ptr] * 8 + [_c_int] + [_c_ptr] * 3 + [_c_int] * 8 + [_c_f32, _c_f32, _c_int,
                                                                                                        _c_ptr],
    "dgv2_gen_tail_fwd": [_c_ptr] * 7 + [_c_int] * 3 + [_c_f32] * 3 + [_c_ptr],
    "dgv2_gen_tail_bwd": [_c_ptr] * 11 + [_c_int] * 3 + [_c_f32] * 3 + [_c_ptr],
    "dgv2_ada_apply": [_c_ptr] * 8 + [_c_int] * 5 + [_c_ptr],
    "dgv2_ada_sample": [_c_ptr] * 7 + [_c_int] * 3 + [_c_ptr],
    "dgv2_ada_build": [_c_ptr] * 8 + [_c_int] * 4 + [_c_ptr],
    "dgv2_coords_convert": [_c_ptr] * 4 + [_c_int] * 3 + [_c_f32] * 3 + [_c_int, _c_ptr],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()). "
            "The MI355X path has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    got = lib.dgv2_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libdgv2.so ABI version {got}, binding expects {ABI_VERSION}")
    return lib


lib = _load()


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise RuntimeError(f"dgv2: unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def check(*tensors):
    """Mirror of the reference's CHECK_INPUT: device-resident and contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dgv2: tensor must be a CUDA/HIP tensor (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("dgv2: tensor must be contiguous")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


ENOTSUP = -3


def call(name, *args):
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def try_call(name, *args):
    """Like call(), but returns False on DGV2_ENOTSUP (the documented 'use the fallback' code)."""
    rc = getattr(lib, name)(*args)
    if rc == ENOTSUP:
        return False
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")
    return True


# ---------------------------------------------------------------------------------------
# status words (include/dgv2.h "Status words"): the library keeps no flag of its own; entries that check a promise of
# the caller on the values they stage OR a bit into a device int32 the caller owns.  One word per device, owned here.
# ---------------------------------------------------------------------------------------
STATUS_X_INEXACT, STATUS_FIR_TABLE = 1, 2
_STATUS_TEXT = {
    STATUS_X_INEXACT: "a launch with an x_exact promise (fp32 conv on conv_x3.hip / its weight gradient: 'these input "
                      "channels hold bf16-representable values') staged a value that is not: that launch computed on the "
                      "bf16 rounding of its input",
    STATUS_FIR_TABLE: "dgv2_fir_same_mfma_prep met a table entry outside its contract: the band operands are unusable",
}
_status_words = {}


def status_word(device=None):
    """The int32 device word of `device` (current device by default) that status-reporting entries OR their bits into."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    w = _status_words.get(idx)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("dgv2: the status word must exist before a hipGraph capture (call dgv2_native.status_word() "
                               "once outside the capture; gans.trainer.Trainer does)")
        w = _status_words[idx] = torch.zeros(1, device=torch.device("cuda", idx), dtype=torch.int32)
    return w


def status_read(clear=True):
    """OR of the status words of every device that has one (synchronises those devices); clears them."""
    bits = 0
    for w in _status_words.values():
        v = int(w.item())
        bits |= v
        if v and clear:
            w.zero_()
    return bits


def status_check():
    """Raise if a status bit is set (and clear it).  Called wherever the host synchronises anyway."""
    bits = status_read()
    if bits:
        raise RuntimeError("dgv2: " + "; ".join(t for b, t in _STATUS_TEXT.items() if bits & b))
