"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the CPU oracle (oracle/ffwm_oracle.c).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; nothing under ``ffwm_amd/`` does (tests/test_abi_cpu.py::test_product_path_never_touches_the_oracle
enforces it).

Each function takes contiguous CPU torch tensors (float32 or float64) and returns new
tensors, mirroring what the reference's CUDA ops compute:

* block_extractor  -- /root/reference/cuda/block_extractor/block_extractor_kernel.cu:21-170
* local_attn_reshape -- cuda/local_attn_reshape/local_attn_reshape_kernel.cu:21-108
* resample2d       -- cuda/resample2d_package/resample2d_kernel.cu:21-330
  all three pinned to golden vectors produced by the reference's own kernels on gfx950
  (oracle/build_ref.py -> oracle/_ref/, tests/golden/reference_ops_gfx950.pt, tests/test_oracle_ref_golden.py)
* warp             -- models/base_networks.py:168-173 (F.grid_sample bilinear/zeros/align_corners=False)
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libffwm_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/libffwm_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("ffwm_oracle.c", "ffwm_oracle_impl.inc")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "--no-print-directory"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _sfx(t):
    if t.dtype == torch.float32:
        return "_f32"
    if t.dtype == torch.float64:
        return "_f64"
    raise TypeError("oracle supports float32/float64 only, got %s" % t.dtype)


def _p(t):
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle wants contiguous CPU tensors"
    return ctypes.c_void_p(t.data_ptr())


_i64 = ctypes.c_int64
_i32 = ctypes.c_int


def _call(name, ref, *args):
    fn = getattr(lib(), name + _sfx(ref))
    fn.restype = None
    fn(*args)


# ---------------------------------------------------------------- block_extractor
def block_extractor_forward(source, flow, k):
    B, C, Hs, Ws = source.shape
    _, two, Hf, Wf = flow.shape
    assert two == 2
    out = source.new_empty(B, C, k * Hf, k * Wf)
    _call("oracle_block_extractor_forward", source, _p(source), _p(flow), _p(out),
          _i64(B), _i64(C), _i64(Hs), _i64(Ws), _i64(Hf), _i64(Wf), _i32(k))
    return out


def block_extractor_backward(source, flow, grad_output, k):
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow.shape
    g_src = torch.zeros_like(source)
    g_flow = torch.zeros_like(flow)
    go = grad_output.contiguous()
    _call("oracle_block_extractor_backward", source, _p(source), _p(flow), _p(go), _p(g_src),
          _p(g_flow), _i64(B), _i64(C), _i64(Hs), _i64(Ws), _i64(Hf), _i64(Wf), _i32(k))
    return g_src, g_flow


# ---------------------------------------------------------------- local_attn_reshape
def local_attn_reshape_forward(inputs, k):
    B, C, H, W = inputs.shape
    assert C == k * k
    out = inputs.new_empty(B, 1, k * H, k * W)
    _call("oracle_local_attn_reshape_forward", inputs, _p(inputs), _p(out), _i64(B), _i64(H),
          _i64(W), _i32(k))
    return out


def local_attn_reshape_backward(grad_output, k):
    B, one, Ho, Wo = grad_output.shape
    H, W = Ho // k, Wo // k
    go = grad_output.contiguous()
    g_in = go.new_zeros(B, k * k, H, W)
    _call("oracle_local_attn_reshape_backward", go, _p(go), _p(g_in), _i64(B), _i64(H), _i64(W),
          _i32(k))
    return g_in


# ---------------------------------------------------------------- resample2d
def resample2d_forward(input1, input2, kernel_size=2, dilation=1):
    _, C, Hi, Wi = input1.shape
    B, three, H, W = input2.shape
    assert three == 3
    out = input1.new_empty(B, C, H, W)
    _call("oracle_resample2d_forward", input1, _p(input1), _p(input2), _p(out), _i64(B), _i64(C),
          _i64(Hi), _i64(Wi), _i64(H), _i64(W), _i32(kernel_size), _i32(dilation))
    return out


def resample2d_backward(input1, input2, grad_output, kernel_size=2, dilation=1,
                        reference_quirk=True):
    _, C, Hi, Wi = input1.shape
    B, _, H, W = input2.shape
    go = grad_output.contiguous()
    g1 = torch.zeros_like(input1)
    g2 = torch.zeros_like(input2)
    _call("oracle_resample2d_backward_input1", input1, _p(input2), _p(go), _p(g1), _i64(B),
          _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W), _i32(kernel_size), _i32(dilation),
          _i32(1 if reference_quirk else 0))
    _call("oracle_resample2d_backward_input2", input1, _p(input1), _p(input2), _p(go), _p(g2),
          _i64(B), _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W), _i32(kernel_size),
          _i32(dilation))
    return g1, g2


# ---------------------------------------------------------------- warp (WarpNet / grid_sample)
def warp_forward(feat, flow, flipcat=False):
    B, C, Hi, Wi = feat.shape
    _, two, H, W = flow.shape
    assert two == 2
    out = feat.new_empty(B, 2 * C if flipcat else C, H, W)
    _call("oracle_warp_forward", feat, _p(feat), _p(flow), _p(out), _i64(B), _i64(C), _i64(Hi),
          _i64(Wi), _i64(H), _i64(W), _i32(1 if flipcat else 0))
    return out


def warp_backward(feat, flow, grad_output, flipcat=False):
    B, C, Hi, Wi = feat.shape
    _, _, H, W = flow.shape
    go = grad_output.contiguous()
    g_feat = torch.zeros_like(feat)
    g_flow = torch.zeros_like(flow)
    _call("oracle_warp_backward", feat, _p(feat), _p(flow), _p(go), _p(g_feat), _p(g_flow),
          _i64(B), _i64(C), _i64(Hi), _i64(Wi), _i64(H), _i64(W), _i32(1 if flipcat else 0))
    return g_feat, g_flow


# ---------------------------------------------------------------- autograd shells (tests only)
class BlockExtractorOracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, source, flow, k):
        ctx.save_for_backward(source, flow)
        ctx.k = k
        return block_extractor_forward(source.contiguous(), flow.contiguous(), k)

    @staticmethod
    def backward(ctx, go):
        source, flow = ctx.saved_tensors
        gs, gf = block_extractor_backward(source.contiguous(), flow.contiguous(), go, ctx.k)
        return gs, gf, None


class LocalAttnReshapeOracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, k):
        ctx.k = k
        return local_attn_reshape_forward(inputs.contiguous(), k)

    @staticmethod
    def backward(ctx, go):
        return local_attn_reshape_backward(go, ctx.k), None


class Resample2dOracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=2, dilation=1, reference_quirk=True):
        ctx.save_for_backward(input1, input2)
        ctx.cfg = (kernel_size, dilation, reference_quirk)
        return resample2d_forward(input1.contiguous(), input2.contiguous(), kernel_size, dilation)

    @staticmethod
    def backward(ctx, go):
        input1, input2 = ctx.saved_tensors
        ks, dil, quirk = ctx.cfg
        g1, g2 = resample2d_backward(input1.contiguous(), input2.contiguous(), go, ks, dil, quirk)
        return g1, g2, None, None, None


class WarpOracleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, flow, flipcat=False):
        ctx.save_for_backward(feat, flow)
        ctx.flipcat = flipcat
        return warp_forward(feat.contiguous(), flow.contiguous(), flipcat)

    @staticmethod
    def backward(ctx, go):
        feat, flow = ctx.saved_tensors
        gf, gfl = warp_backward(feat.contiguous(), flow.contiguous(), go, ctx.flipcat)
        return gf, gfl, None
