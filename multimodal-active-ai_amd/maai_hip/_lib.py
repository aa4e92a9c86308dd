"""ctypes binding of libmaai_hip.so (the C ABI in include/maai_hip.h).

The library is the product: there is NO fallback.  ``lib()`` raises if the
shared object is missing, and every compute wrapper in ``kernels.py`` raises
if the tensors are not on a HIP device.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must come first: libmaai_hip.so has to bind to the HIP runtime torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("MAAI_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libmaai_hip.so")   # (MAAI_LIB_PATH: A/B builds of the kernels)

BF16, F32 = 0, 1

c_p, c_i, c_ll, c_f, c_d, c_ull = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_ulonglong


class ConvDesc(C.Structure):
    """maai_conv_desc"""
    _fields_ = [(n, c_i) for n in ("N", "IH", "IW", "Cin", "Cout", "KH", "KW", "stride", "pad_h", "pad_w", "OHg", "OWg",
                                   "OH", "OW", "out_stride", "out_off_h", "out_off_w", "accumulate")]


class ConvEpilogue(C.Structure):
    """maai_conv_epilogue"""
    _fields_ = [("mode", c_i), ("relu", c_i), ("p0", c_p), ("p1", c_p), ("p2", c_p), ("t", c_p), ("mask_bits", c_i), ("sum_increment", c_i),
                ("a2", c_p), ("ak1", c_p), ("ak2", c_p), ("ak3", c_p), ("a_out", c_p),
                ("xs", c_p), ("xt", c_p), ("x_relu", c_i), ("xb", c_p), ("xs2", c_p), ("xt2", c_p), ("x_out", c_p), ("x_bits", c_p),
                ("pre_x", c_p), ("pre_w", c_p), ("pre_xs", c_p), ("pre_xt", c_p), ("pre_relu", c_i), ("pre_cin", c_i), ("pre_y_out", c_p),
                ("x2", c_p), ("cin1", c_i), ("bias", c_p), ("diag", c_p)]


EPI_STORE, EPI_STATS_ONLY, EPI_BN_ACT, EPI_BWD_REDUCE, EPI_BWD_APPLY, EPI_DGRAD_REDUCE = range(6)
_P_DESC = C.POINTER(ConvDesc)
_P_EPI = C.POINTER(ConvEpilogue)

# name -> (restype, argtypes); mirrors include/maai_hip.h line by line
SIGNATURES = {
    "maai_abi_version": (c_i, []),
    "maai_weight_forms": (c_i, [c_p, c_p, c_p, c_i, c_p]),
    "maai_adam_step_multi": (c_i, [c_p, c_p, c_p, c_i, C.c_double, C.c_double, C.c_double, C.c_double, c_i, C.c_float, c_p]),
    "maai_last_error": (C.c_char_p, []),
    "maai_device_count": (c_i, []),
    "maai_kernel_names": (c_i, [c_i]),
    "maai_last_kernel_name": (C.c_char_p, []),
    "maai_conv2d_igemm": (c_i, [_P_DESC, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "maai_conv2d_stats_rows": (c_ll, [_P_DESC, c_i]),
    "maai_conv2d_kernel_family": (c_i, [_P_DESC, c_i]),
    "maai_conv2d_bn_act_fast": (c_i, [_P_DESC, c_i, c_i]),
    "maai_conv2d_stats_rows_fused": (c_ll, [_P_DESC, _P_EPI, c_i]),
    "maai_conv2d_igemm_fused": (c_i, [_P_DESC, c_p, c_p, c_p, c_p, c_p, _P_EPI, c_i, c_p]),
    "maai_conv_bwd3_rows": (c_i, [c_ll]),
    "maai_conv_bwd3": (c_i, [c_p] * 13 + [c_ll, c_i, c_p]),
    "maai_conv2d_wgrad": (c_i, [_P_DESC, c_p, c_p, c_p, c_i, c_p]),
    "maai_conv2d_wgrad_tuned": (c_i, [_P_DESC, c_p, c_p, c_p, c_i, c_i, c_p]),
    "maai_conv2d_wgrad_xf": (c_i, [_P_DESC, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p]),
    "maai_reduce_partials": (c_i, [c_p, c_ll, c_i, c_p, c_p]),
    "maai_bn_finalize": (c_i, [c_p, c_d, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_i, c_p]),
    "maai_bn_running_update_multi": (c_i, [c_p, c_i, c_p]),
    "maai_bn_eval_coeffs": (c_i, [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p]),
    "maai_bn_act_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "maai_bn_act_fwd_mask": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "maai_bn_act_fwd2": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "maai_bn_act_bwd_apply2": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_p]),
    "maai_bn_bwd_rows": (c_ll, [c_ll, c_i, c_i]),
    "maai_bn_act_bwd_reduce": (c_i, [c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "maai_bn_bwd_coeffs": (c_i, [c_p, c_d, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "maai_bn_bwd_coeffs_f32": (c_i, [c_p, c_d, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p]),
    "maai_bn_pack_stats": (c_i, [c_p, c_d, c_p, c_i, c_p]),
    "maai_bn_finalize_gathered": (c_i, [c_p, c_i, c_ll, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    "maai_bn_act_bwd_apply": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_i, c_i, c_p]),
    "maai_pack_views_u8": (c_i, [C.POINTER(c_p), c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_stem_unroll_nchw_f32": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_stem_unroll_u8": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_nchw_f32_to_nhwc": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_nhwc_to_nchw_f32": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_nchw_f32_from_nhwc_grad": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_avgpool_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_avgpool_bwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    "maai_cast_from_f32": (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    "maai_cast_to_f32": (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    "maai_ntxent_normalize": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "maai_ntxent_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p]),
    "maai_ntxent_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_p]),
    "maai_ntxent_normalize_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "maai_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_ll, c_d, c_d, c_d, c_d, c_i, c_f, c_p]),
    "maai_sgd_step": (c_i, [c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_i, c_p]),
    "maai_sgd_step_multi": (c_i, [c_p, c_p, c_p, c_i, c_f, c_f, c_f, c_i, c_p]),
    "maai_fold_s2": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "maai_fold_dw": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "maai_fold_dgrad_w": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_d, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_p]),
    "maai_gram": (c_i, [c_p, c_ll, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p]),
    "maai_gram_partial_rows": (c_i, [c_ll, c_i]),
    "maai_gram_partials": (c_i, [c_p, c_ll, c_i, c_p, c_p, c_i, c_p, c_p, c_p]),
    "maai_fold_stats": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "maai_conv_dfold_rows": (c_ll, [c_ll]),
    "maai_conv_dfold": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_ll, c_i, c_p]),
    "maai_comm_create": (c_i, [c_i, c_i, c_ll, c_p]),
    "maai_comm_handle": (c_i, [c_p, c_p]),
    "maai_comm_attach": (c_i, [c_p, c_i, c_p]),
    "maai_comm_allgather": (c_i, [c_p, c_p, c_ll, c_p, c_p]),
    "maai_comm_status": (c_i, [c_p, c_p]),
    "maai_comm_poll": (c_i, [c_p, c_p]),
    "maai_comm_destroy": (c_i, [c_p]),
    "maai_multi_sqnorm": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p, c_p]),
    "maai_larc_scale": (c_i, [c_p, c_p, c_p, c_i, c_p, c_f, c_f, c_f, c_f, c_i, c_p]),
    "maai_softmax_ce_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "maai_softmax_ce_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "maai_augment_view_u8": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "maai_foveate_views_u8": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "maai_augment_params": (c_i, [c_p, c_i, c_i, c_i, c_ull, c_i, c_f, c_f, c_f, c_f, c_f, c_p]),
}

_LIB = None


class MaaiError(RuntimeError):
    pass


ABI_VERSION = 6   # == MAAI_ABI_VERSION of include/maai_hip.h (checked in tests/test_host.py); bumped with every signature change


def _autobuild():
    """Build the library when only the sources travelled.  Every rank of a torchrun launch gets here at once, so
    the build runs under an exclusive file lock and build.py links to a temporary file that it renames into place:
    the ranks that lose the race wait, then find a complete library."""
    import fcntl
    import importlib.util
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(LIB_PATH):
                spec = importlib.util.spec_from_file_location("maai_build", os.path.join(PKG_ROOT, "build.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.build(verbose=False)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def lib():
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH) and os.environ.get("MAAI_NO_AUTOBUILD", "0") != "1":
            # the library is an in-tree build product: compile it (hipcc) rather than fail when only
            # the sources travelled; there is still no non-HIP fallback.
            try:
                _autobuild()
            except Exception as e:  # noqa: BLE001
                raise MaaiError("libmaai_hip.so is missing and could not be built: %s" % e)
        if not os.path.exists(LIB_PATH):
            raise MaaiError(
                "libmaai_hip.so is missing (%s). Build it with `python multimodal-active-ai_amd/build.py` "
                "(needs hipcc); there is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        if handle.maai_abi_version() != ABI_VERSION:
            raise MaaiError("libmaai_hip.so has ABI version %d, this package binds version %d: rebuild it "
                            "(python multimodal-active-ai_amd/build.py --force)" % (handle.maai_abi_version(), ABI_VERSION))
        _LIB = handle
    return _LIB


def check(rc, what=""):
    if rc != 0:
        msg = lib().maai_last_error()
        raise MaaiError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
