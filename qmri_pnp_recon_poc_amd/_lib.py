"""ctypes loader for libqmri.so (include/qmri.h).

The product path has no CPU fallback: if the shared library is missing it is built with hipcc (which
cross-compiles gfx950 without a GPU); if that fails, or a symbol is absent, loading raises.  Compute entry
points additionally need a gfx950 device and fail with QMRI_ERR_HIP otherwise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libqmri.so")
CSRC = os.path.join(_PKG, "csrc")

# every symbol include/qmri.h declares (checked at load time and by tests/test_abi.py)
SYMBOLS = [
    "qmri_abi_version", "qmri_create", "qmri_destroy", "qmri_last_error", "qmri_set_stream", "qmri_synchronize",
    "qmri_build_spiral", "qmri_build_epi", "qmri_set_operator", "qmri_operator_m", "qmri_forward", "qmri_adjoint",
    "qmri_forward_f32", "qmri_adjoint_f32", "qmri_forward_dev", "qmri_adjoint_dev", "qmri_set_coils", "qmri_forward_mc", "qmri_adjoint_mc", "qmri_xupdate_mc", "qmri_pnp_admm_mc", "qmri_xupdate", "qmri_net_nparams", "qmri_set_denoiser", "qmri_denoise",
    "qmri_net_forward_dev", "qmri_denoiser_scheme", "qmri_pnp_admm", "qmri_pnp_admm_dev", "qmri_pnp_admm_batch", "qmri_set_dictionary", "qmri_dict_match",
    "qmri_dict_match_dev", "qmri_dict_match_xfit", "qmri_dict_match_xfit_dev", "qmri_recon_batch", "qmri_profile_enable", "qmri_profile_get", "qmri_get_health",
    "qmri_debug_lsqr_stamps", "qmri_debug_conv_stamps", "qmri_debug_lsqr_persist", "qmri_debug_dict_filter", "qmri_debug_conv_resident", "qmri_debug_knob",
    "qmri_onnx_read_unetres",
    "qmri_lrtv", "qmri_prox_tv", "qmri_norm_tv", "qmri_synthesize_tsmi", "qmri_synthesize_tsmi_complex",
]


class NetDesc(C.Structure):
    _fields_ = [("arch", C.c_int32), ("in_nc", C.c_int32), ("out_nc", C.c_int32), ("nc", C.c_int32 * 4),
                ("nb", C.c_int32), ("residual_noise", C.c_int32)]


class AdmmParams(C.Structure):
    _fields_ = [("gamma", C.c_double), ("iters", C.c_int32), ("cg_tol", C.c_double), ("cg_maxit", C.c_int32),
                ("solver", C.c_int32), ("denoiser_type", C.c_int32), ("noise_std", C.c_double), ("want_diag", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("s", C.c_int32), ("T", C.c_int32),
                ("V", C.POINTER(C.c_double)), ("frame_ptr", C.POINTER(C.c_int32)), ("kidx", C.POINTER(C.c_int32)),
                ("net", C.POINTER(NetDesc)), ("weights", C.POINTER(C.c_float)), ("weights_nbytes", C.c_size_t),
                ("K", C.c_int32), ("Q", C.c_int32), ("D", C.POINTER(C.c_float)), ("normD", C.POINTER(C.c_float)),
                ("lut", C.POINTER(C.c_float)), ("admm", AdmmParams), ("slices_per_launch", C.c_int32)]


class LrtvParams(C.Structure):
    _fields_ = [("K", C.c_double), ("iters", C.c_int32), ("step", C.c_double), ("tol", C.c_double), ("backtrack", C.c_int32),
                ("prox_tol", C.c_double), ("prox_maxit", C.c_int32)]


class LrtvInfo(C.Structure):
    _fields_ = [("iters", C.c_int32), ("halvings", C.c_int32), ("step", C.c_double), ("obj", C.c_double),
                ("prox_calls", C.c_int32), ("prox_iters_total", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("ms_xupdate", C.c_double), ("ms_denoiser", C.c_double), ("ms_elementwise", C.c_double),
                ("ms_diag", C.c_double), ("ms_match", C.c_double), ("ms_conv3x3", C.c_double),
                ("n_conv3x3", C.c_int64), ("lsqr_iters", C.c_int64), ("admm_iters", C.c_int64),
                ("ms_tv_iter", C.c_double), ("n_tv_iter", C.c_int64),
                ("flop_conv3x3", C.c_double), ("ms_conv2x2", C.c_double), ("n_conv2x2", C.c_int64), ("flop_conv2x2", C.c_double),
                ("ms_lsqr_kernels", C.c_double), ("n_lsqr_launches", C.c_int64), ("ms_net_forward", C.c_double), ("n_net_forward", C.c_int64)]


class Health(C.Structure):
    _fields_ = [("denoiser_scheme", C.c_int32), ("denoiser_fallbacks", C.c_int32), ("resident_armed", C.c_int32), ("resident_timeouts", C.c_int32),
                ("lsqr_one_launch", C.c_int32), ("lsqr_timeouts", C.c_int32), ("repeated_calls", C.c_int32), ("reserved", C.c_int32),
                ("last_call_wall_ms", C.c_double), ("last_call_stage_ms", C.c_double * 4), ("set_denoiser_ms", C.c_double * 3)]


def build(force: bool = False) -> str:
    """Compile libqmri.so for gfx950 with hipcc (in-tree, next to this file)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "qmri.h"))
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        t = os.path.getmtime(LIB_PATH)
        stale = any(os.path.getmtime(s) > t for s in srcs)
    if stale:
        subprocess.run(["make", "-C", CSRC, "-s", "-j4"], check=True)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # QMRI_LIBQMRI: another build of the same library (A/B timing of two builds on one box, tools/ab_build.sh); default: in-tree
    path = os.environ.get("QMRI_LIBQMRI") or LIB_PATH
    if path == LIB_PATH and not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(path)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise ImportError(f"libqmri.so lacks symbols declared in include/qmri.h: {missing}")
    vp, ip, dp, fp, i = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int
    L.qmri_abi_version.restype = i
    L.qmri_create.argtypes = [i, C.POINTER(vp)]
    L.qmri_destroy.argtypes = [vp]
    L.qmri_last_error.restype = C.c_char_p
    L.qmri_last_error.argtypes = [vp]
    L.qmri_set_stream.argtypes = [vp, vp]
    L.qmri_synchronize.argtypes = [vp]
    L.qmri_build_spiral.argtypes = [vp, i, i, i, ip, ip, i, C.POINTER(i)]
    L.qmri_build_epi.argtypes = [vp, i, i, C.c_double, i, ip, ip, i, C.POINTER(i)]
    L.qmri_set_operator.argtypes = [vp, i, i, i, i, dp, ip, ip, i]
    L.qmri_operator_m.argtypes = [vp, C.POINTER(i)]
    L.qmri_forward.argtypes = [vp, vp, i, vp]
    L.qmri_adjoint.argtypes = [vp, vp, vp]
    L.qmri_forward_f32.argtypes = [vp, fp, i, fp]
    L.qmri_adjoint_f32.argtypes = [vp, fp, fp]
    L.qmri_forward_dev.argtypes = [vp, vp, vp, i]
    L.qmri_adjoint_dev.argtypes = [vp, vp, vp, i]
    L.qmri_set_coils.argtypes = [vp, i, vp]
    L.qmri_forward_mc.argtypes = [vp, vp, i, vp]
    L.qmri_adjoint_mc.argtypes = [vp, vp, vp]
    L.qmri_xupdate_mc.argtypes = [vp, vp, vp, C.c_double, C.c_double, i, vp, vp, ip, ip]
    L.qmri_pnp_admm_mc.argtypes = [vp, vp, C.POINTER(AdmmParams), vp, vp, ip]
    L.qmri_xupdate.argtypes = [vp, vp, vp, C.c_double, C.c_double, i, i, vp, ip, ip]
    L.qmri_net_nparams.restype = C.c_size_t
    L.qmri_net_nparams.argtypes = [C.POINTER(NetDesc)]
    L.qmri_set_denoiser.argtypes = [vp, C.POINTER(NetDesc), fp, C.c_size_t, i, i, i]
    L.qmri_denoise.argtypes = [vp, dp, i, i, i, i, dp]
    L.qmri_onnx_read_unetres.argtypes = [C.c_char_p, C.POINTER(NetDesc), fp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.qmri_net_forward_dev.argtypes = [vp, vp, i, vp]
    L.qmri_denoiser_scheme.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    L.qmri_get_health.argtypes = [vp, C.POINTER(Health)]
    L.qmri_pnp_admm.argtypes = [vp, vp, C.POINTER(AdmmParams), vp, vp, vp, dp, ip]
    L.qmri_pnp_admm_dev.argtypes = [vp, i, vp, C.POINTER(AdmmParams), vp, vp, vp, dp, ip]
    L.qmri_pnp_admm_batch.argtypes = [vp, i, i, vp, C.POINTER(AdmmParams), vp, vp, vp, dp, ip]
    L.qmri_lrtv.argtypes = [vp, vp, C.POINTER(LrtvParams), vp, C.POINTER(LrtvInfo)]
    L.qmri_prox_tv.argtypes = [vp, dp, i, i, C.c_double, C.c_double, i, dp, ip, dp]
    L.qmri_norm_tv.argtypes = [vp, dp, i, i, dp]
    L.qmri_synthesize_tsmi.argtypes = [vp, dp, i, fp, ip]
    L.qmri_synthesize_tsmi_complex.argtypes = [vp, dp, dp, i, fp, ip]
    L.qmri_set_dictionary.argtypes = [vp, i, i, i, fp, fp, fp]
    L.qmri_dict_match.argtypes = [vp, vp, i, fp, fp, fp, ip]
    L.qmri_dict_match_dev.argtypes = [vp, vp, i, vp, vp, vp, vp]
    L.qmri_dict_match_xfit.argtypes = [vp, vp, i, fp, fp, fp, ip, fp]
    L.qmri_dict_match_xfit_dev.argtypes = [vp, vp, i, vp, vp, vp, vp, vp]
    L.qmri_recon_batch.argtypes = [i, C.POINTER(i), i, C.POINTER(Problem), vp, vp, fp, fp, C.c_char_p, C.c_size_t]
    L.qmri_debug_knob.argtypes = [C.c_char_p, i]
    L.qmri_profile_enable.argtypes = [vp, i]
    L.qmri_profile_get.argtypes = [vp, C.POINTER(Profile), i]
    if L.qmri_abi_version() != 1:
        raise ImportError("libqmri.so ABI version mismatch")
    _lib = L
    return L
