"""ctypes binding of libffwm_hip.so (C ABI: include/ffwm_hip.h).

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing
an op raises.  Build it with ``python -m ffwm_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libffwm_hip.so")

F32, F64 = 0, 1
ABI_VERSION = 5

_lib = None

_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i = ctypes.c_int

_SIGNATURES = {
    # name: argtypes
    "ffwm_block_extractor_forward": [_p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_block_extractor_backward": [_p, _p, _p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_block_extractor_backward_strided": [_p, _p, _p, ctypes.POINTER(_i64), _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_local_attn_reshape_backward_strided": [_p, ctypes.POINTER(_i64), _p] + [_i64] * 3 + [_i, _i, _i, _p],
    "ffwm_resample2d_backward_strided": [_p, _p, _p, ctypes.POINTER(_i64), _p, _p] + [_i64] * 6 + [_i, _i, _i, _i, _p],
    "ffwm_bn_lrelu_forward": [_p] * 9 + [_i64] * 3 + [ctypes.c_double] * 3 + [_i, _p],
    "ffwm_bn_lrelu_backward": [_p] * 10 + [_i64] * 3 + [ctypes.c_double] + [_i, _p],
    "ffwm_bn_res_act_forward": [_p] * 11 + [_i64] * 3 + [ctypes.c_double] * 3 + [_i, _i, _p],
    "ffwm_bn_res_act_backward": [_p] * 11 + [_i64] * 3 + [ctypes.c_double] + [_i, _i, _p],
    "ffwm_mfm_forward": [_p, _p, _p] + [_i64] * 3 + [_i, _p],
    "ffwm_mfm_backward": [_p, _p, _p, _p] + [_i64] * 3 + [_i, _p],
    "ffwm_add_act_forward": [_p, _p, _p, _i64, _i, ctypes.c_double, _i, _p],
    "ffwm_add_act_backward": [_p, _p, _p, _i64, _i, ctypes.c_double, _i, _p],
    "ffwm_sigmoid_gate_forward": [_p, _p, _p, _p, _p, _i64, _i, _p],
    "ffwm_sigmoid_gate_backward": [_p, _p, _p, _p, _p, _i64, _i, _p],
    "ffwm_bias_relu_forward": [_p, _p, _p] + [_i64] * 3 + [_i, _p],
    "ffwm_bias_act_forward": [_p, _p, _p, _p] + [_i64] * 5 + [_i, ctypes.c_double, _i, _p],
    "ffwm_flow_head_forward": [_p, _p, _p, _p] + [_i64] * 4 + [_i, _p],
    "ffwm_conv_thin_forward": [_p, _p, _p, _p] + [_i64] * 5 + [_i, ctypes.c_double, _i, _p],
    "ffwm_flow_up_forward": [_p, _p, _p, _p] + [_i64] * 4 + [_i, _p],
    "ffwm_flow_head_backward": [_p, _p, _p, _p, _p] + [_i64] * 4 + [_i, _p],
    "ffwm_flow_up_backward": [_p, _p, _p] + [_i64] * 4 + [_i, _p],
    "ffwm_conv2d_forward": [_p, _p, _p, _p, _p] + [_i64] * 5 + [_i, _i, _i, _i, _i64, _i64, _i, ctypes.c_double, _p, _i64, _i, _p],
    "ffwm_conv2d_wgrad": [_p, _p, _p] + [_i64] * 7 + [_i, _i, _i, _i, _p],
    "ffwm_conv2d_wgrad_tiled": [_p, _p, _p, _p] + [_i64] * 7 + [_i, _i, _i, _i, _i, _p],
    "ffwm_conv3x3_winograd_forward": [_p, _p, _p, _p, _p] + [_i64] * 5 + [_i, _i, ctypes.c_double, _i, _p],
    "ffwm_adam_step": [_p, _p, _p, _p, _i64] + [ctypes.c_double] * 4 + [_i64, _i, _p],
    "ffwm_adam_step_device": [_p, _p, _p, _p, _i64] + [ctypes.c_double] * 4 + [_p, _i, _p],
    "ffwm_conv3x3_wgrad": [_p, _p, _p, _p] + [_i64] * 5 + [_i, _p],
    "ffwm_conv3x3_wgrad_block": [_p, _p, _p, _p] + [_i64] * 9 + [_i, _p],
    "ffwm_block_attention_forward": [_p, _p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_block_attention_backward": [_p, _p, _p, _p, _p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_local_attn_reshape_forward": [_p, _p] + [_i64] * 3 + [_i, _i, _p],
    "ffwm_local_attn_reshape_backward": [_p, _p] + [_i64] * 3 + [_i, _i, _i, _p],
    "ffwm_resample2d_forward": [_p, _p, _p] + [_i64] * 6 + [_i, _i, _i, _p],
    "ffwm_resample2d_backward": [_p, _p, _p, _p, _p] + [_i64] * 6 + [_i, _i, _i, _i, _p],
    "ffwm_warp_forward": [_p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_warp_backward": [_p, _p, _p, _p, _p] + [_i64] * 6 + [_i, _i, _p],
    "ffwm_warp_multi_forward": [_p, _i, _i, _i, _p],          # (array of ffwm_warp_problem, n, flipcat, dtype, stream)
    "ffwm_warp_multi_backward": [_p, _i, _i, _i, _p],
    # (host array of ffwm_sn_layer / ffwm_sn_grad_layer structs, see ffwm_amd/spectral_norm.py)
    "ffwm_spectral_norm_forward": [_p, _i, _i, ctypes.c_double, _i, _p],
    "ffwm_spectral_norm_backward": [_p, _i, _i, _p],
    "ffwm_guided_filter_forward": [_p, _p, _p, _p, _i64, _i64, _i64, _i, ctypes.c_double, _i, _p],
    "ffwm_affine_regularization": [_p, _p, _p, _p, _i64, _i64, _i64, _i, ctypes.c_double, _i, _p],
    "ffwm_correlation_colmax": [_p, _p, _p, _i64, _i64, _i64, _i, _p],
    "ffwm_guided_filter_backward": [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p],
    "ffwm_l1_multi": [_p, _i, _p, _p, _i, _i, _p],          # (array of ffwm_l1_problem, n, out, grad_out, n_slots, dtype, stream)
    "ffwm_conv3x3_winograd_weights_multi": [_p, _i, _i, _p],      # (array of ffwm_wino_weights, n, dtype, stream)
    "ffwm_prof_enable": [_i],
    "ffwm_prof_collect": [],
    "ffwm_prof_get": [_i, ctypes.c_char_p, _i, ctypes.POINTER(_i64), ctypes.POINTER(ctypes.c_double),
                      ctypes.POINTER(ctypes.c_double)],
    "ffwm_prof_get_flops": [_i, ctypes.POINTER(ctypes.c_double)],
    "ffwm_prof_get_bound": [_i, ctypes.POINTER(ctypes.c_double)],
    "ffwm_prof_reset": [],
    "ffwm_set_option": [ctypes.c_char_p, _i],
    "ffwm_zero_fill": [_p, _i64, _p],
    "ffwm_abi_version": [],
}

EXPORTS = sorted(list(_SIGNATURES) + ["ffwm_last_error", "ffwm_conv3x3_winograd_workspace_bytes", "ffwm_conv3x3_winograd_splits",
                                      "ffwm_conv2d_forward_workspace"])


class FFWMError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FFWMError(
            "libffwm_hip.so is missing (%s). The HIP extension is the only implementation of the "
            "ffwm_amd ops -- build it with `python -m ffwm_amd.build`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _i
    lib.ffwm_conv3x3_winograd_workspace_bytes.argtypes = [_i64, _i64]
    lib.ffwm_conv3x3_winograd_workspace_bytes.restype = _i64
    lib.ffwm_conv3x3_winograd_splits.argtypes = [_i64, _i64, _i64, _i64, _i64, _i]
    lib.ffwm_conv3x3_winograd_splits.restype = _i
    lib.ffwm_conv2d_forward_workspace.argtypes = [_i64] * 5 + [_i] * 4
    lib.ffwm_conv2d_forward_workspace.restype = _i64
    lib.ffwm_last_error.argtypes = []
    lib.ffwm_last_error.restype = ctypes.c_char_p
    got = lib.ffwm_abi_version()
    if got != ABI_VERSION:
        raise FFWMError("libffwm_hip.so ABI version %d, expected %d: rebuild it" % (got, ABI_VERSION))
    _lib = lib
    # FFWM_OPTS="key=value,key=value": tuning / ablation switches (ffwm_set_option) from the environment, for experiments
    # with unmodified callers (bench.py, tools/)
    for kv in filter(None, os.environ.get("FFWM_OPTS", "").split(",")):
        k, _, v = kv.partition("=")
        if lib.ffwm_set_option(k.strip().encode(), int(v)) < 0:
            raise FFWMError("FFWM_OPTS: unknown option %r" % k)
        if k.strip() == "ablate" and int(v) != 0:
            import warnings
            warnings.warn("FFWM_OPTS: ablate=%s -- timing-only kernel variants are ON: forward / backward RESULTS ARE WRONG "
                          "(bench experiments only)" % v)
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ffwm_last_error()
        raise FFWMError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))


# ---- profiler / options -------------------------------------------------------------------
def prof_enable(on=True):
    return load().ffwm_prof_enable(1 if on else 0)


def prof_reset():
    load().ffwm_prof_reset()


def prof_collect():
    """-> {kernel_name: {"launches": n, "total_ms": ms, "avg_ms": ms, "bytes_per_launch": b, "flops_per_launch": f,
    "roofline_ms": the time the binding roofline of every launch allows, summed}}"""
    lib = load()
    n = lib.ffwm_prof_collect()
    rows = {}
    for i in range(n):
        name = ctypes.create_string_buffer(128)
        launches = _i64()
        ms = ctypes.c_double()
        nbytes = ctypes.c_double()
        check(lib.ffwm_prof_get(i, name, 128, ctypes.byref(launches), ctypes.byref(ms),
                                ctypes.byref(nbytes)), "ffwm_prof_get")
        flops = ctypes.c_double()
        check(lib.ffwm_prof_get_flops(i, ctypes.byref(flops)), "ffwm_prof_get_flops")
        bound = ctypes.c_double()
        check(lib.ffwm_prof_get_bound(i, ctypes.byref(bound)), "ffwm_prof_get_bound")
        k = max(launches.value, 1)
        rows[name.value.decode()] = {"launches": launches.value, "total_ms": ms.value,
                                     "avg_ms": ms.value / k, "bytes_per_launch": nbytes.value / k,
                                     "flops_per_launch": flops.value / k, "roofline_ms": bound.value}
    return rows


def set_option(key, value):
    rc = load().ffwm_set_option(key.encode(), int(value))
    if rc < 0 and key not in ("be_fwd_variant", "be_bwd_variant", "channel_slab", "xcd_remap", "ablate", "rows_per_thread", "scatter_variant", "be_bwd_halo", "be_bwd_rows", "warp_fwd_variant"):
        check(rc, "ffwm_set_option")
    return rc
