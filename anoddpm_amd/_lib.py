"""ctypes binding of libanoddpm_hip.so (C ABI declared in include/anoddpm_hip.h).

The library is mandatory: there is NO CPU or eager-PyTorch fallback anywhere in this package.
`lib()` raises if the shared object is missing, and `require_cuda()` raises for non-HIP tensors.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_double, c_float, c_int16, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANODDPM_LIB_TAG=<tag>: load lib/libanoddpm_hip_<tag>.so instead -- a second build of the same sources with other compiler flags
# (ANODDPM_BUILD_TAG / ANODDPM_EXTRA_FLAGS of anoddpm_amd.build), for A/B measurements of one gpurun session.  Same ABI, same checks.
SO_PATH = os.path.join(_HERE, "lib", "libanoddpm_hip%s.so" % ("_" + os.environ["ANODDPM_LIB_TAG"] if os.environ.get("ANODDPM_LIB_TAG") else ""))
ABI_VERSION = 24

OP_IGEMM, OP_GN_STATS, OP_SOFTMAX, OP_RESAMPLE, OP_LINEAR, OP_POSEMB, OP_STEM, OP_LAYOUT, OP_CHAN_STATS, OP_GN_FINALIZE, OP_HEAD = range(1, 12)
(OP_WGRAD3, OP_WGRAD1, OP_GN_BWD, OP_PACK, OP_SOFTMAX_BWD, OP_TRANSPOSE, OP_LINEAR_BWD, OP_STEM_BWD, OP_HEAD_BWD,
 OP_COLSUM_FOLD, OP_ATTENTION, OP_PACK_BATCH, OP_LINEAR_BWD_BATCH, OP_DROPOUT) = range(16, 30)
OP_MAX = 32


class SimplexArgs(Structure):
    _fields_ = [("out", c_void_p), ("zvals", c_void_p), ("tables", c_void_p), ("table_sel", c_void_p),
                ("z0", c_int64), ("out_slice_stride", c_int64),
                ("nslices", c_int32), ("H", c_int32), ("W", c_int32),
                ("table_slice_stride", c_int32), ("table_sel_scale", c_int32), ("octaves", c_int32),
                ("persistence", c_double), ("frequency", c_double)]


class PUpdateArgs(Structure):
    _fields_ = [("x_prev", c_void_p), ("pred_x0", c_void_p), ("mean_out", c_void_p), ("x_t", c_void_p),
                ("eps", c_void_p), ("noise", c_void_p), ("t", c_void_p),
                ("c_recip", c_void_p), ("c_recipm1", c_void_p), ("c_coef1", c_void_p),
                ("c_coef2", c_void_p), ("c_sigma", c_void_p),
                ("n", c_int64), ("B", c_int32), ("T", c_int32)]


class IgemmArgs(Structure):
    _fields_ = [("a0", c_void_p), ("a1", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p),
                ("bmat", c_void_p), ("bias", c_void_p), ("temb", c_void_p), ("res", c_void_p),
                ("out", c_void_p), ("ws", c_void_p),
                ("a0_bs", c_int64), ("a0_hs", c_int64), ("a1_bs", c_int64), ("a1_hs", c_int64),
                ("b_bs", c_int64), ("b_hs", c_int64),
                ("o_bs", c_int64), ("o_hs", c_int64), ("r_bs", c_int64), ("r_hs", c_int64),
                ("c0", c_int32), ("c1", c_int32), ("a0_ld", c_int32), ("a1_ld", c_int32),
                ("H", c_int32), ("W", c_int32), ("ks", c_int32), ("a_mode", c_int32), ("act", c_int32),
                ("b_mode", c_int32), ("ldb", c_int32), ("N", c_int32), ("temb_ld", c_int32),
                ("out_ld", c_int32), ("res_ld", c_int32), ("B", c_int32), ("heads", c_int32),
                ("ksplit", c_int32), ("cfg", c_int32), ("alpha", c_float), ("gn_ld", c_int32), ("stats", c_void_p), ("stats_rows", c_int32),
                ("tail_csum", c_void_p), ("tail_other", c_void_p), ("tail_gamma", c_void_p), ("tail_beta", c_void_p),
                ("tail_scale", c_void_p), ("tail_shift", c_void_p), ("tail_mean", c_void_p), ("tail_rstd", c_void_p),
                ("tail_c1", c_int32), ("tail_groups", c_int32), ("tail_eps", c_float),
                ("fold_stats0", c_void_p), ("fold_stats1", c_void_p), ("fold_gamma", c_void_p), ("fold_beta", c_void_p),
                ("fold_rows0", c_int32), ("fold_rows1", c_int32), ("fold_fmt0", c_int32), ("fold_fmt1", c_int32),
                ("fold_groups", c_int32), ("fold_eps", c_float), ("res_mode", c_int32), ("stats_csum", c_void_p),
                ("gnb_partial", c_void_p), ("gnb_x0", c_void_p), ("gnb_x1", c_void_p), ("gnb_gamma", c_void_p), ("gnb_beta", c_void_p),
                ("gnb_mean", c_void_p), ("gnb_rstd", c_void_p), ("gnb_x0_bs", c_int64), ("gnb_x1_bs", c_int64),
                ("gnb_c0", c_int32), ("gnb_x0_ld", c_int32), ("gnb_x1_ld", c_int32), ("gnb_groups", c_int32)]


class GnArgs(Structure):
    _fields_ = [("a0", c_void_p), ("a1", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("scale", c_void_p), ("shift", c_void_p), ("partial", c_void_p),
                ("a0_bs", c_int64), ("a1_bs", c_int64),
                ("c0", c_int32), ("c1", c_int32), ("a0_ld", c_int32), ("a1_ld", c_int32),
                ("P", c_int32), ("B", c_int32), ("groups", c_int32), ("nslab", c_int32), ("eps", c_float)]


class ChanStatsArgs(Structure):
    _fields_ = [("a", c_void_p), ("stats", c_void_p), ("a_bs", c_int64),
                ("C", c_int32), ("a_ld", c_int32), ("P", c_int32), ("B", c_int32), ("nslab", c_int32)]


class GnFinalizeArgs(Structure):
    _fields_ = [("stats0", c_void_p), ("stats1", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("scale", c_void_p), ("shift", c_void_p),
                ("rows0", c_int32), ("rows1", c_int32), ("c0", c_int32), ("c1", c_int32),
                ("P", c_int32), ("B", c_int32), ("groups", c_int32), ("eps", c_float),
                ("mean_out", c_void_p), ("rstd_out", c_void_p), ("fmt0", c_int32), ("fmt1", c_int32)]


class HeadArgs(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p),
                ("out", c_void_p), ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("Cout", c_int32)]


class SoftmaxArgs(Structure):
    _fields_ = [("x", c_void_p), ("rows", c_int64), ("L", c_int32)]


class LinearBwdBatchArgs(Structure):
    _fields_ = [("jobs", c_void_p), ("x", c_void_p), ("dx", c_void_p), ("ws", c_void_p), ("njobs", c_int32), ("max_n", c_int32),
                ("B", c_int32), ("K", c_int32), ("act_in", c_int32), ("acc_w", c_int32), ("acc_x", c_int32)]


class PackBatchArgs(Structure):
    _fields_ = [("jobs", c_void_p), ("block0", c_void_p), ("njobs", c_int32), ("nblocks", c_int32)]


class AttentionArgs(Structure):
    _fields_ = [("qkv", c_void_p), ("out", c_void_p), ("probs", c_void_p), ("B", c_int32), ("L", c_int32), ("heads", c_int32),
                ("ch", c_int32), ("scale", c_float)]


class ResampleArgs(Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("B", c_int32), ("H", c_int32), ("W", c_int32),
                ("C", c_int32), ("mode", c_int32), ("scale", c_float), ("accumulate", c_int32),
                ("gn_scale", c_void_p), ("gn_shift", c_void_p), ("out_act", c_void_p)]


class LinearArgs(Structure):
    _fields_ = [("inp", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
                ("B", c_int32), ("K", c_int32), ("N", c_int32), ("act_in", c_int32), ("act_out", c_int32)]


class PosembArgs(Structure):
    _fields_ = [("t", c_void_p), ("freqs", c_void_p), ("out", c_void_p), ("B", c_int32), ("dim", c_int32),
                ("scale", c_float), ("zero", c_void_p), ("zero_doubles", c_int64)]


class StemArgs(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("out", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("stats", c_void_p), ("stats_rows", c_int32)]


class LayoutArgs(Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("B", c_int32), ("P", c_int32), ("C", c_int32),
                ("in_ld", c_int32)]


class Op(Structure):
    _fields_ = [("code", c_int32), ("flags", c_int32), ("args", c_void_p)]


class AdamwArgs(Structure):
    _fields_ = [("p", c_void_p), ("m", c_void_p), ("v", c_void_p), ("ema", c_void_p), ("g", c_void_p),
                ("grad_scale", c_void_p), ("n", c_int64),
                ("lr", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("weight_decay", c_float), ("ema_decay", c_float), ("step", c_int32)]


class AnomalyArgs(Structure):
    _fields_ = [("recon", c_void_p), ("real", c_void_p), ("mask", c_void_p),
                ("mean", c_void_p), ("sqerr", c_void_p), ("mse_img", c_void_p), ("thr_img", c_void_p), ("pred", c_void_p),
                ("counts", c_void_p), ("workspace", c_void_p), ("workspace_doubles", c_int64),
                ("n", c_int64), ("recon_as", c_int64), ("recon_bs", c_int64),
                ("navg", c_int32), ("B", c_int32), ("threshold", c_float)]


class VlbArgs(Structure):
    _fields_ = [("x0", c_void_p), ("xt", c_void_p), ("eps", c_void_p), ("noise", c_void_p), ("t", c_void_p),
                ("c_recip", c_void_p), ("c_recipm1", c_void_p), ("c_coef1", c_void_p), ("c_coef2", c_void_p),
                ("c_post_logvar", c_void_p), ("c_model_logvar", c_void_p),
                ("pred_x0", c_void_p), ("out", c_void_p), ("workspace", c_void_p), ("workspace_doubles", c_int64),
                ("n", c_int64), ("B", c_int32), ("T", c_int32)]


class LossArgs(Structure):
    _fields_ = [("eps", c_void_p), ("noise", c_void_p), ("x0", c_void_p), ("xt", c_void_p), ("t", c_void_p), ("weights", c_void_p),
                ("c_recip", c_void_p), ("c_recipm1", c_void_p), ("c_coef1", c_void_p), ("c_coef2", c_void_p),
                ("c_post_logvar", c_void_p), ("c_model_logvar", c_void_p),
                ("per_sample", c_void_p), ("vlb", c_void_p), ("total", c_void_p), ("workspace", c_void_p), ("workspace_doubles", c_int64),
                ("g_per", c_void_p), ("g_vlb", c_void_p), ("g_total", c_void_p), ("d_eps", c_void_p),
                ("n", c_int64), ("B", c_int32), ("T", c_int32), ("kind", c_int32)]


class DropoutArgs(Structure):
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p), ("n", c_int64),
                ("seed", ctypes.c_uint64), ("B", c_int32), ("C", c_int32), ("mode", c_int32), ("p", c_float)]


class WgradArgs(Structure):
    _fields_ = [("a0", c_void_p), ("a1", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p), ("dy", c_void_p),
                ("dw", c_void_p), ("ws", c_void_p), ("ws_floats", c_int64),
                ("a0_bs", c_int64), ("a1_bs", c_int64), ("dy_bs", c_int64),
                ("c0", c_int32), ("c1", c_int32), ("a0_ld", c_int32), ("a1_ld", c_int32), ("dy_ld", c_int32),
                ("H", c_int32), ("W", c_int32), ("N", c_int32), ("B", c_int32),
                ("a_mode", c_int32), ("act", c_int32), ("gn_ld", c_int32), ("band", c_int32), ("accumulate", c_int32),
                ("colsum", c_void_p), ("algo", c_int32), ("dimg", c_void_p), ("dbias", c_void_p)]


class GnBwdArgs(Structure):
    _fields_ = [("x0", c_void_p), ("x1", c_void_p), ("da", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("mean", c_void_p), ("rstd", c_void_p), ("dx0", c_void_p), ("dx1", c_void_p),
                ("dgamma", c_void_p), ("dbeta", c_void_p), ("partial", c_void_p), ("coef", c_void_p),
                ("x0_bs", c_int64), ("x1_bs", c_int64), ("da_bs", c_int64), ("dx0_bs", c_int64), ("dx1_bs", c_int64),
                ("c0", c_int32), ("c1", c_int32), ("x0_ld", c_int32), ("x1_ld", c_int32), ("da_ld", c_int32),
                ("dx0_ld", c_int32), ("dx1_ld", c_int32), ("Hs", c_int32), ("Ws", c_int32),
                ("B", c_int32), ("groups", c_int32), ("nslab", c_int32),
                ("act", c_int32), ("a_mode", c_int32), ("acc_dx", c_int32),
                ("dres", c_void_p), ("dres_bs", c_int64), ("dres_ld", c_int32), ("partial_ready", c_int32)]


class Wgrad1Args(Structure):
    _fields_ = [("a0", c_void_p), ("a1", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p), ("dy", c_void_p),
                ("dw", c_void_p), ("dbias", c_void_p), ("ws", c_void_p), ("ws_floats", c_int64),
                ("a0_bs", c_int64), ("a1_bs", c_int64), ("dy_bs", c_int64),
                ("c0", c_int32), ("c1", c_int32), ("a0_ld", c_int32), ("a1_ld", c_int32), ("dy_ld", c_int32),
                ("P", c_int32), ("N", c_int32), ("B", c_int32), ("act", c_int32), ("gn_ld", c_int32),
                ("span", c_int32), ("accumulate", c_int32)]


class PackArgs(Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("N", c_int32), ("K", c_int32), ("kind", c_int32),
                ("bwd", c_int32), ("k0", c_int32), ("kc", c_int32)]


class SoftmaxBwdArgs(Structure):
    _fields_ = [("p", c_void_p), ("dp", c_void_p), ("rows", c_int64), ("L", c_int32)]


class TransposeArgs(Structure):
    _fields_ = [("inp", c_void_p), ("out", c_void_p), ("Z", c_int32), ("L", c_int32)]


class LinearBwdArgs(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("dx", c_void_p),
                ("B", c_int32), ("K", c_int32), ("N", c_int32), ("act_in", c_int32), ("acc_w", c_int32), ("acc_x", c_int32)]


class StemBwdArgs(Structure):
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("dy", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("dx", c_void_p),
                ("ws", c_void_p), ("ws_floats", c_int64),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32)]


class HeadBwdArgs(Structure):
    _fields_ = [("x", c_void_p), ("gn_scale", c_void_p), ("gn_shift", c_void_p), ("w", c_void_p), ("dy", c_void_p),
                ("da", c_void_p), ("dw", c_void_p), ("db", c_void_p), ("ws", c_void_p), ("ws_floats", c_int64),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("C", c_int32), ("Cout", c_int32)]


class ColsumFoldArgs(Structure):
    _fields_ = [("colsum", c_void_p), ("dimg", c_void_p), ("dbias", c_void_p), ("B", c_int32), ("ipb", c_int32), ("N", c_int32)]


class MriSliceArgs(Structure):
    _fields_ = [("vols", c_void_p), ("ydim", c_void_p), ("slice_idx", c_void_p), ("affine", c_void_p), ("out", c_void_p),
                ("B", c_int32), ("X", c_int32), ("Z", c_int32), ("crop", c_int32), ("pad_left", c_int32), ("crop_top", c_int32)]


class ResizeArgs(Structure):
    _fields_ = [("inp", c_void_p), ("tmp", c_void_p), ("out", c_void_p), ("kx", c_void_p), ("kx_min", c_void_p), ("kx_n", c_void_p),
                ("ky", c_void_p), ("ky_min", c_void_p), ("ky_n", c_void_p),
                ("B", c_int32), ("in_h", c_int32), ("in_w", c_int32), ("out_h", c_int32), ("out_w", c_int32),
                ("kmax_x", c_int32), ("kmax_y", c_int32), ("mean", c_float), ("std", c_float), ("normalize", c_int32)]


ANOMALY_NCOUNTS = 12
ANOMALY_BLOCKS = 64

_STRUCTS = [SimplexArgs, PUpdateArgs, IgemmArgs, GnArgs, SoftmaxArgs, ResampleArgs, LinearArgs,
            PosembArgs, StemArgs, LayoutArgs, Op, AdamwArgs, ChanStatsArgs, GnFinalizeArgs, HeadArgs, AnomalyArgs, VlbArgs, WgradArgs, GnBwdArgs,
            Wgrad1Args, PackArgs, SoftmaxBwdArgs, TransposeArgs, LinearBwdArgs, StemBwdArgs, HeadBwdArgs, ColsumFoldArgs,
            MriSliceArgs, ResizeArgs, AttentionArgs, PackBatchArgs, LinearBwdBatchArgs, LossArgs, DropoutArgs]

# every symbol include/anoddpm_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "anoddpm_abi_version", "anoddpm_last_error", "anoddpm_device_count", "anoddpm_struct_size",
    "anoddpm_simplex_perm_init", "anoddpm_simplex3_octaves_f64", "anoddpm_simplex3_octaves_f32",
    "anoddpm_simplex3_grid_f64", "anoddpm_simplex2_octaves_f64", "anoddpm_simplex2_grid_f64",
    "anoddpm_q_sample", "anoddpm_p_sample_update", "anoddpm_chain_advance",
    "anoddpm_igemm", "anoddpm_f43_channel_sliced", "anoddpm_smallmap_tile", "anoddpm_wino23s_tile", "anoddpm_pack_wino43_bf16x3", "anoddpm_gn_stats", "anoddpm_chan_stats", "anoddpm_gn_finalize", "anoddpm_softmax_rows", "anoddpm_resample2x",
    "anoddpm_linear_small", "anoddpm_posemb", "anoddpm_conv_stem", "anoddpm_stem_stats_rows", "anoddpm_conv_head", "anoddpm_nhwc_to_nchw",
    "anoddpm_run_ops", "anoddpm_prof_enable", "anoddpm_prof_active", "anoddpm_prof_collect", "anoddpm_prof_list",
    "anoddpm_adamw_ema", "anoddpm_sumsq", "anoddpm_anomaly_map", "anoddpm_vlb_terms", "anoddpm_conv3x3_wgrad", "anoddpm_gn_silu_backward", "anoddpm_pack_conv3x3",
    "anoddpm_wgrad_pointwise", "anoddpm_pack_weights", "anoddpm_softmax_rows_backward", "anoddpm_transpose_square",
    "anoddpm_linear_small_backward", "anoddpm_conv_stem_backward", "anoddpm_conv_head_backward", "anoddpm_colsum_fold",
    "anoddpm_volume_normalise", "anoddpm_mri_slice_prepare", "anoddpm_resize_bilinear_pil", "anoddpm_attention", "anoddpm_wgrad43_groups", "anoddpm_wgrad43_colsum_items", "anoddpm_pack_batch", "anoddpm_pack_job_blocks", "anoddpm_linear_small_backward_batch",
    "anoddpm_loss_forward", "anoddpm_loss_backward", "anoddpm_dropout",
]

_lib = None


class AnoddpmError(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """libanoddpm_hip.so must share ONE HIP runtime with PyTorch (streams, device pointers).  The torch
    wheel bundles its own libamdhip64 (SONAME libamdhip64.so.7, the same as /opt/rocm's); whichever copy is
    mapped first satisfies later DT_NEEDED lookups, so torch's copy is mapped before our library."""
    import glob
    import torch  # noqa: F401  (maps libtorch_hip and its bundled runtime)
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        for path in glob.glob(os.path.join(tl, name + "*")):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass
            break


def lib():
    """Load the HIP library (once).  Fails loudly: this package has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise AnoddpmError(
            f"{SO_PATH} is missing: build it with `python -m anoddpm_amd.build` "
            "(hipcc --offload-arch=gfx950).  anoddpm_amd has no CPU / eager fallback.")
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(SO_PATH)
    L.anoddpm_last_error.restype = ctypes.c_char_p
    L.anoddpm_struct_size.argtypes = [c_int32]
    if L.anoddpm_abi_version() != ABI_VERSION:
        raise AnoddpmError("libanoddpm_hip.so ABI version mismatch; rebuild")
    for i, st in enumerate(_STRUCTS):
        if L.anoddpm_struct_size(i) != ctypes.sizeof(st):
            raise AnoddpmError(f"ABI struct {st.__name__}: C sizeof {L.anoddpm_struct_size(i)} != ctypes {ctypes.sizeof(st)}")
    L.anoddpm_simplex_perm_init.argtypes = [c_int64, POINTER(c_int16), POINTER(c_int16)]
    for name in ("anoddpm_simplex3_octaves_f64", "anoddpm_simplex3_octaves_f32"):
        getattr(L, name).argtypes = [POINTER(SimplexArgs), c_void_p]
    L.anoddpm_simplex3_grid_f64.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                            c_void_p, c_void_p]
    L.anoddpm_simplex2_octaves_f64.argtypes = [c_void_p, c_int32, c_void_p, c_int32, c_double, c_double, c_void_p]
    L.anoddpm_simplex2_grid_f64.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
    L.anoddpm_q_sample.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_int32, c_int64, c_int32, c_void_p]
    L.anoddpm_p_sample_update.argtypes = [POINTER(PUpdateArgs), c_void_p]
    L.anoddpm_chain_advance.argtypes = [c_void_p, c_int32, c_void_p, c_void_p]
    L.anoddpm_igemm.argtypes = [POINTER(IgemmArgs), c_void_p]
    L.anoddpm_gn_stats.argtypes = [POINTER(GnArgs), c_void_p]
    L.anoddpm_chan_stats.argtypes = [POINTER(ChanStatsArgs), c_void_p]
    L.anoddpm_gn_finalize.argtypes = [POINTER(GnFinalizeArgs), c_void_p]
    L.anoddpm_softmax_rows.argtypes = [POINTER(SoftmaxArgs), c_void_p]
    L.anoddpm_attention.argtypes = [POINTER(AttentionArgs), c_void_p]
    L.anoddpm_pack_batch.argtypes = [POINTER(PackBatchArgs), c_void_p]
    L.anoddpm_linear_small_backward_batch.argtypes = [POINTER(LinearBwdBatchArgs), c_void_p]
    L.anoddpm_pack_job_blocks.argtypes = [POINTER(PackArgs)]
    L.anoddpm_pack_job_blocks.restype = c_int64
    L.anoddpm_resample2x.argtypes = [POINTER(ResampleArgs), c_void_p]
    L.anoddpm_linear_small.argtypes = [POINTER(LinearArgs), c_void_p]
    L.anoddpm_posemb.argtypes = [POINTER(PosembArgs), c_void_p]
    L.anoddpm_conv_stem.argtypes = [POINTER(StemArgs), c_void_p]
    L.anoddpm_stem_stats_rows.argtypes = [c_int32, c_int32, c_int32, c_int32]
    L.anoddpm_conv_head.argtypes = [POINTER(HeadArgs), c_void_p]
    L.anoddpm_nhwc_to_nchw.argtypes = [POINTER(LayoutArgs), c_void_p]
    L.anoddpm_run_ops.argtypes = [POINTER(Op), c_int32, c_void_p]
    L.anoddpm_prof_enable.argtypes = [c_int32]
    L.anoddpm_prof_collect.argtypes = [POINTER(c_double), POINTER(c_int64)]
    L.anoddpm_prof_list.argtypes = [POINTER(c_int32), POINTER(c_float), c_int32]
    L.anoddpm_pack_wino43_bf16x3.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_void_p]
    L.anoddpm_adamw_ema.argtypes = [POINTER(AdamwArgs), c_void_p]
    L.anoddpm_sumsq.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_float, c_void_p]
    L.anoddpm_anomaly_map.argtypes = [POINTER(AnomalyArgs), c_void_p]
    L.anoddpm_vlb_terms.argtypes = [POINTER(VlbArgs), c_void_p]
    L.anoddpm_dropout.argtypes = [POINTER(DropoutArgs), c_void_p]
    L.anoddpm_loss_forward.argtypes = [POINTER(LossArgs), c_void_p]
    L.anoddpm_loss_backward.argtypes = [POINTER(LossArgs), c_void_p]
    L.anoddpm_conv3x3_wgrad.argtypes = [POINTER(WgradArgs), c_void_p]
    L.anoddpm_wgrad43_groups.argtypes = [c_int32] * 5
    L.anoddpm_f43_channel_sliced.argtypes = [c_int32] * 4
    L.anoddpm_wgrad43_colsum_items.argtypes = [c_int32] * 5
    L.anoddpm_gn_silu_backward.argtypes = [POINTER(GnBwdArgs), c_void_p]
    L.anoddpm_pack_conv3x3.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]
    L.anoddpm_wgrad_pointwise.argtypes = [POINTER(Wgrad1Args), c_void_p]
    L.anoddpm_pack_weights.argtypes = [POINTER(PackArgs), c_void_p]
    L.anoddpm_softmax_rows_backward.argtypes = [POINTER(SoftmaxBwdArgs), c_void_p]
    L.anoddpm_transpose_square.argtypes = [POINTER(TransposeArgs), c_void_p]
    L.anoddpm_linear_small_backward.argtypes = [POINTER(LinearBwdArgs), c_void_p]
    L.anoddpm_conv_stem_backward.argtypes = [POINTER(StemBwdArgs), c_void_p]
    L.anoddpm_conv_head_backward.argtypes = [POINTER(HeadBwdArgs), c_void_p]
    L.anoddpm_colsum_fold.argtypes = [POINTER(ColsumFoldArgs), c_void_p]
    L.anoddpm_volume_normalise.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]
    L.anoddpm_mri_slice_prepare.argtypes = [POINTER(MriSliceArgs), c_void_p]
    L.anoddpm_resize_bilinear_pil.argtypes = [POINTER(ResizeArgs), c_void_p]
    # kernel-variant selectors (internal, not in the public header).  A product build accepts only the keys whose values all
    # compute correct results; a key that names a timing ablation raises instead of silently corrupting outputs
    L.anoddpm_internal_variant.argtypes = [ctypes.c_int32, ctypes.c_int32]
    for i in range(16):
        if os.environ.get(f"ANODDPM_DEBUG{i}"):
            if L.anoddpm_internal_variant(i, int(os.environ[f"ANODDPM_DEBUG{i}"], 0)) != 0:
                raise AnoddpmError(f"ANODDPM_DEBUG{i} selects a timing ablation (wrong results by design): rebuild with "
                                   "ANODDPM_ABLATE=1 python -m anoddpm_amd.build --force to use it")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().anoddpm_last_error()
        raise AnoddpmError(f"{what or 'anoddpm'} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


def require_cuda(t, what):
    """The HIP path is the only path: refuse CPU tensors instead of silently computing elsewhere."""
    if not t.is_cuda:
        raise AnoddpmError(
            f"{what}: tensor is on '{t.device}', but anoddpm_amd runs on MI355X (HIP) devices only; "
            "there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).")
    return t


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
