"""
ctypes binding of ``libtapqir_hip.so`` (C ABI declared in ``include/tapqir_hip.h``).

There is no fallback: if the shared library is missing or fails to load, every entry point
raises ``HipExtensionError``.  Build it with ``python -m tapqir_amd.build`` (or
``__graft_entry__.build()``).
"""

import ctypes as C
import os

from tapqir_amd.exceptions import HipExtensionError

_HERE = os.path.dirname(os.path.abspath(__file__))
# TAPQIR_AMD_LIB selects another build of the same library (A/B timing of compiler flags)
LIB_PATH = os.environ.get("TAPQIR_AMD_LIB") or os.path.join(_HERE, "libtapqir_hip.so")

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class KsmognArgs(C.Structure):
    """``tq_ksmogn_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("images", C.c_void_p), ("images_il", C.c_void_p), ("pixstats", C.c_void_p), ("xy", C.c_void_p),
        ("ndx", C.c_void_p), ("fdx", C.c_void_p), ("background", C.c_void_p), ("height", C.c_void_p), ("width", C.c_void_p),
        ("x", C.c_void_p), ("y", C.c_void_p), ("gain", C.c_void_p),
        ("offset_samples", C.c_void_p), ("offset_logits", C.c_void_p),
        ("gout", C.c_void_p), ("m_logit", C.c_void_p), ("aoi_mask", C.c_void_p),
        ("ll", C.c_void_p), ("g_background", C.c_void_p), ("g_height", C.c_void_p),
        ("g_width", C.c_void_p), ("g_x", C.c_void_p), ("g_y", C.c_void_p), ("g_gain", C.c_void_p),
        ("m_kstride", C.c_int64), ("stats_stride", C.c_int64),
        ("nb", C.c_int32), ("fb", C.c_int32), ("C", C.c_int32), ("F", C.c_int32),
        ("P", C.c_int32), ("K", C.c_int32), ("O", C.c_int32),
        ("nb_full", C.c_int32), ("il_min_units", C.c_int32),
        ("scale", C.c_float), ("pixel_mode", C.c_int32), ("images_by_slot", C.c_int32),
    ]


class CosmosArgs(C.Structure):
    """``tq_cosmos_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("images", C.c_void_p), ("images_il", C.c_void_p), ("pixstats", C.c_void_p), ("xy", C.c_void_p),
        ("is_ontarget", C.c_void_p), ("aoi_mask", C.c_void_p),
        ("ndx", C.c_void_p), ("fdx", C.c_void_p), ("offset_samples", C.c_void_p), ("offset_logits", C.c_void_p),
        ("params", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
        ("lat", C.c_void_p), ("site", C.c_void_p), ("pix", C.c_void_p), ("aoi_part", C.c_void_p), ("blk_part", C.c_void_p),
        ("gsum", C.c_void_p), ("globals", C.c_void_p), ("gbase", C.c_void_p), ("elbo_out", C.c_void_p),
        ("Nt", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("P", C.c_int32), ("K", C.c_int32), ("O", C.c_int32),
        ("nb", C.c_int32), ("fb", C.c_int32), ("n_offset", C.c_int32), ("draw_globals", C.c_int32),
        ("draw_locals", C.c_int32), ("il_min_units", C.c_int32),
        ("scale_n", C.c_float), ("scale", C.c_float), ("global_weight", C.c_float),
        ("eps", C.c_float),
        ("width_min", C.c_float), ("width_max", C.c_float), ("height_std", C.c_float),
        ("background_mean_std", C.c_float), ("background_std_std", C.c_float),
        ("gain_std", C.c_float), ("lamda_rate", C.c_float), ("proximity_rate", C.c_float),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float),
        ("bias_correction1", C.c_float), ("bias_correction2", C.c_float),
        ("zero_grad", C.c_int32), ("fuse_adam", C.c_int32), ("crosstalk", C.c_int32),
        ("seed", C.c_uint64), ("step", C.c_uint32),
        ("last_step", C.c_void_p), ("beta1_d", C.c_double), ("beta2_d", C.c_double),
        ("pixel_mode", C.c_int32), ("tail_kind", C.c_int32), ("sync", C.c_void_p), ("sync_value", C.c_int32),
        ("images_by_slot", C.c_int32), ("next_ndx", C.c_void_p), ("next_fdx", C.c_void_p),
    ]


class XtalkArgs(C.Structure):
    """``tq_xtalk_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("images", C.c_void_p), ("images_il", C.c_void_p), ("pixstats", C.c_void_p), ("xy", C.c_void_p),
        ("ndx", C.c_void_p), ("fdx", C.c_void_p),
        ("background", C.c_void_p), ("height", C.c_void_p), ("width", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p),
        ("gain", C.c_void_p), ("alpha", C.c_void_p), ("offset_samples", C.c_void_p), ("offset_logits", C.c_void_p),
        ("gout", C.c_void_p), ("m_logit", C.c_void_p), ("aoi_mask", C.c_void_p),
        ("ll_joint", C.c_void_p), ("ll", C.c_void_p), ("ell_excess", C.c_void_p),
        ("g_background", C.c_void_p), ("g_height", C.c_void_p), ("g_width", C.c_void_p), ("g_x", C.c_void_p),
        ("g_y", C.c_void_p), ("g_gain", C.c_void_p), ("g_alpha", C.c_void_p),
        ("m_kstride", C.c_int64),
        ("nb", C.c_int32), ("fb", C.c_int32), ("C", C.c_int32), ("F", C.c_int32),
        ("P", C.c_int32), ("K", C.c_int32), ("O", C.c_int32),
        ("nb_full", C.c_int32), ("il_min_units", C.c_int32),
        ("scale", C.c_float),
    ]


class ProbsArgs(C.Structure):
    """``tq_probs_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("params", C.c_void_p), ("is_ontarget", C.c_void_p), ("globals_p", C.c_void_p), ("gbase_p", C.c_void_p),
        ("xy_given", C.c_void_p), ("z_probs", C.c_void_p), ("theta_probs", C.c_void_p),
        ("Nt", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("P", C.c_int32), ("K", C.c_int32),
        ("particles", C.c_int32), ("draw", C.c_int32), ("eps", C.c_float), ("seed", C.c_uint64),
    ]


class RsampleArgs(C.Structure):
    """``tq_rsample_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("height", C.c_void_p), ("width", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("xy", C.c_void_p),
        ("background", C.c_void_p), ("gain", C.c_void_p), ("offset_samples", C.c_void_p), ("offset_logits", C.c_void_p),
        ("out", C.c_void_p), ("B", C.c_int64), ("P", C.c_int32), ("K", C.c_int32), ("O", C.c_int32), ("seed", C.c_uint64),
    ]


class SnrArgs(C.Structure):
    """``tq_snr_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("images", C.c_void_p), ("xy", C.c_void_p), ("height", C.c_void_p), ("width", C.c_void_p), ("x", C.c_void_p),
        ("y", C.c_void_p), ("background", C.c_void_p), ("snr", C.c_void_p), ("chi2", C.c_void_p),
        ("U", C.c_int64), ("P", C.c_int32), ("K", C.c_int32),
        ("gain", C.c_float), ("offset_mean", C.c_float), ("offset_var", C.c_float),
    ]


class GlimpseArgs(C.Structure):
    """``tq_glimpse_args`` (include/tapqir_hip.h)."""

    _fields_ = [
        ("frames", C.c_void_p), ("raw_xy", C.c_void_p), ("images", C.c_void_p), ("target_xy", C.c_void_p),
        ("offset_hist", C.c_void_p), ("status", C.c_void_p),
        ("H", C.c_int32), ("W", C.c_int32), ("N", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("P", C.c_int32),
        ("c", C.c_int32), ("f0", C.c_int32), ("nf", C.c_int32),
        ("offset_x", C.c_int32), ("offset_y", C.c_int32), ("offset_P", C.c_int32),
    ]


# every symbol include/tapqir_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "tq_version", "tq_last_error", "tq_ksmogn_log_prob", "tq_ksmogn_crosstalk_log_prob", "tq_crosstalk_param_count",
    "tq_interleaved_floats", "tq_interleaved_floats_n", "tq_images_interleave_n", "tq_images_interleave", "tq_image_stats",
    "tq_globals_size", "tq_gbase_size", "tq_cosmos_nblk", "tq_cosmos_param_count",
    "tq_cosmos_sample_globals", "tq_cosmos_sample_locals", "tq_cosmos_elbo_grads",
    "tq_cosmos_globals_grad", "tq_cosmos_adam", "tq_cosmos_adam_catchup", "tq_cosmos_step", "tq_cosmos_step_overlapped", "tq_cosmos_tail", "tq_cosmos_tail_reduced", "tq_cosmos_sample_locals_range",
    "tq_cosmos_blk_floats", "tq_cosmos_minibatch_step", "tq_cosmos_pixel_unit",
    "tq_cosmos_probs", "tq_glimpse_extract", "tq_ksmogn_rsample", "tq_snr_chi2",
]

_lib = None


def load():
    """Load the HIP library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -m tapqir_amd.build`). "
            "tapqir_amd has no CPU fallback for the SVI hot path."
        )
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as err:
        raise HipExtensionError(f"failed to load {LIB_PATH}: {err}")
    lib.tq_version.restype = C.c_int
    lib.tq_last_error.restype = C.c_char_p
    for name in ("tq_globals_size", "tq_gbase_size"):
        getattr(lib, name).restype = C.c_int64
    lib.tq_cosmos_nblk.restype = C.c_int64
    lib.tq_cosmos_nblk.argtypes = [C.c_int64]
    lib.tq_cosmos_param_count.restype = C.c_int64
    lib.tq_cosmos_param_count.argtypes = [C.c_int32] * 4
    lib.tq_crosstalk_param_count.restype = C.c_int64
    lib.tq_crosstalk_param_count.argtypes = [C.c_int32] * 4
    lib.tq_ksmogn_crosstalk_log_prob.argtypes = [C.POINTER(XtalkArgs), C.c_void_p]
    lib.tq_ksmogn_crosstalk_log_prob.restype = C.c_int
    lib.tq_interleaved_floats.restype = C.c_int64
    lib.tq_interleaved_floats.argtypes = [C.c_int64, C.c_int32]
    lib.tq_interleaved_floats_n.restype = C.c_int64
    lib.tq_interleaved_floats_n.argtypes = [C.c_int64, C.c_int32]
    lib.tq_images_interleave_n.restype = C.c_int
    lib.tq_images_interleave_n.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.tq_images_interleave.restype = C.c_int
    lib.tq_images_interleave.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.tq_image_stats.restype = C.c_int
    lib.tq_image_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    lib.tq_ksmogn_log_prob.argtypes = [C.POINTER(KsmognArgs), C.c_void_p]
    lib.tq_ksmogn_log_prob.restype = C.c_int
    for name in ("tq_cosmos_sample_globals", "tq_cosmos_sample_locals", "tq_cosmos_elbo_grads",
                 "tq_cosmos_globals_grad", "tq_cosmos_adam", "tq_cosmos_adam_catchup", "tq_cosmos_step", "tq_cosmos_tail",
                 "tq_cosmos_pixel_unit"):
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(CosmosArgs), C.c_void_p]
        fn.restype = C.c_int
    lib.tq_cosmos_blk_floats.argtypes = [C.c_int32] * 4 + [C.c_int64]
    lib.tq_cosmos_blk_floats.restype = C.c_int64
    for name in ("tq_cosmos_step_overlapped", "tq_cosmos_tail_reduced", "tq_cosmos_minibatch_step"):
        fn = getattr(lib, name)
        fn.argtypes = [C.POINTER(CosmosArgs), C.POINTER(CosmosArgs), C.c_void_p]
        fn.restype = C.c_int
    lib.tq_cosmos_sample_locals_range.argtypes = [C.POINTER(CosmosArgs), C.c_int32, C.c_int32, C.POINTER(CosmosArgs), C.c_void_p]
    lib.tq_cosmos_sample_locals_range.restype = C.c_int
    lib.tq_cosmos_adam_catchup.argtypes = [C.POINTER(CosmosArgs), C.c_int32, C.c_void_p]
    lib.tq_cosmos_adam_catchup.restype = C.c_int
    lib.tq_cosmos_probs.argtypes = [C.POINTER(ProbsArgs), C.c_void_p]
    lib.tq_cosmos_probs.restype = C.c_int
    lib.tq_ksmogn_rsample.argtypes = [C.POINTER(RsampleArgs), C.c_void_p]
    lib.tq_ksmogn_rsample.restype = C.c_int
    lib.tq_snr_chi2.argtypes = [C.POINTER(SnrArgs), C.c_void_p]
    lib.tq_snr_chi2.restype = C.c_int
    lib.tq_glimpse_extract.argtypes = [C.POINTER(GlimpseArgs), C.c_void_p]
    lib.tq_glimpse_extract.restype = C.c_int
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().tq_last_error().decode()
        raise HipExtensionError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Raw address of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()
