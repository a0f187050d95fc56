"""Tensor-level entry points of the HIP hot path: torch tensors in, C ABI underneath.

PyTorch is plumbing here (device memory, the current HIP stream); every function below is one
call into libffwm_hip.so through ``include/ffwm_hip.h``.  CPU tensors are refused, as the
reference's wrappers refuse them (/root/reference/models/external_function.py:37-38,84-85):
there is no CPU or eager fallback in this package.
"""
import ctypes

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("ffwm_amd ops support float32/float64 (AT_DISPATCH_FLOATING_TYPES), got %s" % t.dtype)


def _go_strides(grad_output):
    """(ctypes int64[4] or None, keep-alive): the element strides of a grad_output that is not contiguous -- handed to the
    *_backward_strided entry points, which read it the way the reference's kernels do (DIM3_INDEX with the tensor's strides,
    cuda/block_extractor/block_extractor_kernel.cu:8-15; the reference's Functions drop the result of .contiguous(),
    models/external_function.py:46-47).  Views with a negative stride or a stride-0 last dimension are copied instead."""
    if grad_output.is_contiguous():
        return grad_output, None
    st = grad_output.stride()
    if any(x < 0 for x in st) or st[3] == 0:
        return grad_output.contiguous(), None
    return grad_output, (ctypes.c_int64 * 4)(*st)


def _check(name, *tensors, strided_ok=None):
    ref = tensors[0]
    for t in tensors:
        if t is None:
            continue
        if t is strided_ok:
            if not t.is_cuda or t.device != ref.device or t.dtype != ref.dtype or t.dim() != 4:
                raise ValueError("%s: grad_output must be a 4-D tensor of the operands' device and dtype" % name)
            continue
        if not t.is_cuda:
            raise NotImplementedError("%s: ffwm_amd ops run on the GPU only (got a %s tensor)" % (name, t.device))
        if t.device != ref.device:
            raise ValueError("%s: tensors live on different devices (%s vs %s)" % (name, t.device, ref.device))
        if t.dtype != ref.dtype:
            raise TypeError("%s: mixed dtypes (%s vs %s)" % (name, t.dtype, ref.dtype))
        if t.dim() != 4:
            raise ValueError("%s: expected 4-D NCHW tensors, got %d-D" % (name, t.dim()))
        if not t.is_contiguous():
            raise ValueError("%s: tensors must be contiguous NCHW" % name)


def _ptr(t):
    return None if t is None else t.data_ptr()


class _on_device(object):
    """Launch on the tensors' device and its current stream (the reference has no device guard)."""

    __slots__ = ("dev", "prev")

    def __init__(self, t):
        self.dev = t.device.index
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.dev:
            self.prev = cur
            torch.cuda.set_device(self.dev)
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
        return False


# ---------------------------------------------------------------- block_extractor
def block_extractor_forward(source, flow_field, kernel_size, out=None):
    """-> out[B,C,k*Hf,k*Wf]; reference block_extractor_kernel.cu:21-85."""
    _check("block_extractor_forward", source, flow_field, out)
    B, C, Hs, Ws = source.shape
    Bf, two, Hf, Wf = flow_field.shape
    if two != 2:
        raise ValueError("block_extractor_forward: flow_field must have 2 channels")
    if Bf < B:
        raise ValueError("block_extractor_forward: flow_field batch %d < source batch %d" % (Bf, B))
    k = int(kernel_size)
    if out is None:
        out = source.new_empty((B, C, k * Hf, k * Wf))
    elif tuple(out.shape) != (B, C, k * Hf, k * Wf):
        raise ValueError("block_extractor_forward: output has the wrong shape")
    if out.numel() == 0:
        return out
    with _on_device(source) as stream:
        _lib.check(_lib.load().ffwm_block_extractor_forward(
            _ptr(source), _ptr(flow_field), _ptr(out), B, C, Hs, Ws, Hf, Wf, k, _dtype_code(source), stream),
            "ffwm_block_extractor_forward")
    return out


def block_extractor_backward(source, flow_field, grad_output, kernel_size, grad_source=None,
                             grad_flow_field=None):
    """Accumulates (+=) into the given, zero-filled gradient buffers; None skips one."""
    grad_output, gst = _go_strides(grad_output)
    _check("block_extractor_backward", source, flow_field, grad_output, grad_source, grad_flow_field, strided_ok=grad_output if gst is not None else None)
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow_field.shape
    k = int(kernel_size)
    if tuple(grad_output.shape) != (B, C, k * Hf, k * Wf):
        raise ValueError("block_extractor_backward: grad_output has the wrong shape")
    if grad_output.numel() == 0:
        return grad_source, grad_flow_field
    with _on_device(source) as stream:
        if gst is None:
            _lib.check(_lib.load().ffwm_block_extractor_backward(
                _ptr(source), _ptr(flow_field), _ptr(grad_output), _ptr(grad_source), _ptr(grad_flow_field),
                B, C, Hs, Ws, Hf, Wf, k, _dtype_code(source), stream), "ffwm_block_extractor_backward")
        else:              # a non-contiguous grad_output is read through its strides (no copy): the per-element kernel
            _lib.check(_lib.load().ffwm_block_extractor_backward_strided(
                _ptr(source), _ptr(flow_field), _ptr(grad_output), gst, _ptr(grad_source), _ptr(grad_flow_field),
                B, C, Hs, Ws, Hf, Wf, k, _dtype_code(source), stream), "ffwm_block_extractor_backward_strided")
    return grad_source, grad_flow_field


# ---------------------------------------------------------------- block attention (fused consumer)
def block_attention_forward(source, flow_field, weights, kernel_size, out=None):
    """out[B,C,Hf,Wf] = avg_pool2d(block_extractor(source, flow) * local_attn_reshape(weights), k) without the
    k^2-fold expanded tensors; weights[B,k*k,Hf,Wf]."""
    _check("block_attention_forward", source, flow_field, weights, out)
    B, C, Hs, Ws = source.shape
    Bf, two, Hf, Wf = flow_field.shape
    k = int(kernel_size)
    if two != 2 or Bf < B:
        raise ValueError("block_attention_forward: flow_field must be [>=B, 2, Hf, Wf]")
    if tuple(weights.shape) != (B, k * k, Hf, Wf):
        raise ValueError("block_attention_forward: weights must be [B, k*k, Hf, Wf], got %s" % (tuple(weights.shape),))
    if out is None:
        out = source.new_empty((B, C, Hf, Wf))
    elif tuple(out.shape) != (B, C, Hf, Wf):
        raise ValueError("block_attention_forward: output has the wrong shape")
    if out.numel() == 0:
        return out
    with _on_device(source) as stream:
        _lib.check(_lib.load().ffwm_block_attention_forward(
            _ptr(source), _ptr(flow_field), _ptr(weights), _ptr(out), B, C, Hs, Ws, Hf, Wf, k, _dtype_code(source),
            stream), "ffwm_block_attention_forward")
    return out


def block_attention_backward(source, flow_field, weights, grad_output, kernel_size, grad_source=None,
                             grad_flow_field=None, grad_weights=None):
    """Accumulates (+=) into the given, zero-filled gradient buffers; None skips one."""
    _check("block_attention_backward", source, flow_field, weights, grad_output, grad_source, grad_flow_field,
           grad_weights)
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow_field.shape
    k = int(kernel_size)
    if tuple(grad_output.shape) != (B, C, Hf, Wf) or tuple(weights.shape) != (B, k * k, Hf, Wf):
        raise ValueError("block_attention_backward: grad_output / weights have the wrong shape")
    if grad_output.numel() == 0:
        return grad_source, grad_flow_field, grad_weights
    with _on_device(source) as stream:
        _lib.check(_lib.load().ffwm_block_attention_backward(
            _ptr(source), _ptr(flow_field), _ptr(weights), _ptr(grad_output), _ptr(grad_source),
            _ptr(grad_flow_field), _ptr(grad_weights), B, C, Hs, Ws, Hf, Wf, k, _dtype_code(source), stream),
            "ffwm_block_attention_backward")
    return grad_source, grad_flow_field, grad_weights


# ---------------------------------------------------------------- local_attn_reshape
def local_attn_reshape_forward(inputs, kernel_size, out=None):
    """-> out[B,1,k*H,k*W]; reference local_attn_reshape_kernel.cu:21-61."""
    _check("local_attn_reshape_forward", inputs, out)
    B, C, H, W = inputs.shape
    k = int(kernel_size)
    if C != k * k:
        raise ValueError("local_attn_reshape_forward: need C == k*k (C=%d, k=%d)" % (C, k))
    if out is None:
        out = inputs.new_empty((B, 1, k * H, k * W))
    elif tuple(out.shape) != (B, 1, k * H, k * W):
        raise ValueError("local_attn_reshape_forward: output has the wrong shape")
    if out.numel() == 0:
        return out
    with _on_device(inputs) as stream:
        _lib.check(_lib.load().ffwm_local_attn_reshape_forward(
            _ptr(inputs), _ptr(out), B, H, W, k, _dtype_code(inputs), stream),
            "ffwm_local_attn_reshape_forward")
    return out


def local_attn_reshape_backward(grad_output, kernel_size, grad_inputs=None, accumulate=False):
    """grad_inputs[B,k*k,H,W]; accumulate=True is the reference's += into a zero-filled buffer."""
    grad_output, gst = _go_strides(grad_output)
    _check("local_attn_reshape_backward", grad_inputs if grad_inputs is not None else grad_output, grad_output, grad_inputs,
           strided_ok=grad_output if gst is not None else None)
    B, one, Ho, Wo = grad_output.shape
    k = int(kernel_size)
    if one != 1 or Ho % k or Wo % k:
        raise ValueError("local_attn_reshape_backward: grad_output must be [B,1,k*H,k*W]")
    H, W = Ho // k, Wo // k
    if grad_inputs is None:
        if accumulate:
            raise ValueError("local_attn_reshape_backward: accumulate=True needs a grad_inputs buffer")
        grad_inputs = grad_output.new_empty((B, k * k, H, W))
    elif tuple(grad_inputs.shape) != (B, k * k, H, W):
        raise ValueError("local_attn_reshape_backward: grad_inputs has the wrong shape")
    if grad_output.numel() == 0:
        return grad_inputs
    with _on_device(grad_output) as stream:
        if gst is None:
            _lib.check(_lib.load().ffwm_local_attn_reshape_backward(
                _ptr(grad_output), _ptr(grad_inputs), B, H, W, k, 1 if accumulate else 0,
                _dtype_code(grad_output), stream), "ffwm_local_attn_reshape_backward")
        else:
            _lib.check(_lib.load().ffwm_local_attn_reshape_backward_strided(
                _ptr(grad_output), gst, _ptr(grad_inputs), B, H, W, k, 1 if accumulate else 0,
                _dtype_code(grad_output), stream), "ffwm_local_attn_reshape_backward_strided")
    return grad_inputs


# ---------------------------------------------------------------- resample2d
def resample2d_forward(input1, input2, kernel_size=2, dilation=1, out=None):
    """-> out[B,C,H,W]; reference resample2d_kernel.cu:21-95."""
    _check("resample2d_forward", input1, input2, out)
    B1, C, Hi, Wi = input1.shape
    B, three, H, W = input2.shape
    if three != 3:
        raise ValueError("resample2d_forward: input2 must be [B,3,H,W] = (dx, dy, sigma)")
    if B1 < B:
        raise ValueError("resample2d_forward: input1 batch %d < input2 batch %d" % (B1, B))
    if out is None:
        out = input1.new_empty((B, C, H, W))
    elif tuple(out.shape) != (B, C, H, W):
        raise ValueError("resample2d_forward: output has the wrong shape")
    if out.numel() == 0:
        return out
    with _on_device(input1) as stream:
        _lib.check(_lib.load().ffwm_resample2d_forward(
            _ptr(input1), _ptr(input2), _ptr(out), B, C, Hi, Wi, H, W, int(kernel_size), int(dilation),
            _dtype_code(input1), stream), "ffwm_resample2d_forward")
    return out


def resample2d_backward(input1, input2, grad_output, kernel_size=2, dilation=1, grad_input1=None,
                        grad_input2=None, reference_quirk=True, overwrite_input1=False):
    """grad_input1 += (zero-fill it first) -- or, with overwrite_input1, grad_input1[:B] = (the buffer may be uninitialised: the owned-tile
    kernel stores every cell once, any other path clears the buffer itself) --, grad_input2 is overwritten; None skips one."""
    grad_output, gst = _go_strides(grad_output)
    _check("resample2d_backward", input1, input2, grad_output, grad_input1, grad_input2, strided_ok=grad_output if gst is not None else None)
    _, C, Hi, Wi = input1.shape
    B, _, H, W = input2.shape
    if tuple(grad_output.shape) != (B, C, H, W):
        raise ValueError("resample2d_backward: grad_output has the wrong shape")
    if grad_output.numel() == 0:
        return grad_input1, grad_input2
    flags = (1 if reference_quirk else 0) | (2 if overwrite_input1 else 0)
    with _on_device(input1) as stream:
        if gst is None:
            _lib.check(_lib.load().ffwm_resample2d_backward(
                _ptr(input1), _ptr(input2), _ptr(grad_output), _ptr(grad_input1), _ptr(grad_input2), B, C, Hi,
                Wi, H, W, int(kernel_size), int(dilation), flags, _dtype_code(input1), stream), "ffwm_resample2d_backward")
        else:
            _lib.check(_lib.load().ffwm_resample2d_backward_strided(
                _ptr(input1), _ptr(input2), _ptr(grad_output), gst, _ptr(grad_input1), _ptr(grad_input2), B, C, Hi,
                Wi, H, W, int(kernel_size), int(dilation), flags, _dtype_code(input1), stream), "ffwm_resample2d_backward_strided")
    return grad_input1, grad_input2


# ---------------------------------------------------------------- warp (WarpNet)
def warp_forward(feat, flow, flipcat=False, out=None):
    """grid_sample(bilinear, zeros, align_corners=False) [+ flip + cat]; base_networks.py:168-173,326-329."""
    _check("warp_forward", feat, flow, out)
    B, C, Hi, Wi = feat.shape
    Bf, two, H, W = flow.shape
    if two != 2 or Bf != B:
        raise ValueError("warp_forward: flow must be [B,2,H,W] with the features' batch size")
    Co = 2 * C if flipcat else C
    if out is None:
        out = feat.new_empty((B, Co, H, W))
    elif tuple(out.shape) != (B, Co, H, W):
        raise ValueError("warp_forward: output has the wrong shape")
    if out.numel() == 0:
        return out
    with _on_device(feat) as stream:
        _lib.check(_lib.load().ffwm_warp_forward(
            _ptr(feat), _ptr(flow), _ptr(out), B, C, Hi, Wi, H, W, 1 if flipcat else 0, _dtype_code(feat),
            stream), "ffwm_warp_forward")
    return out


def warp_backward(feat, flow, grad_output, flipcat=False, grad_feat=None, grad_flow=None, overwrite_feat=False):
    """Both gradients accumulate (+=) into zero-filled buffers; None skips one.  overwrite_feat=True: grad_feat may be UNINITIALISED --
    the library produces it whole (the owned-tile kernel stores instead of adding and the zero-fill is saved; every other path zero-fills
    it itself: flipcat bit 1 of ffwm_warp_backward)."""
    _check("warp_backward", feat, flow, grad_output, grad_feat, grad_flow)
    B, C, Hi, Wi = feat.shape
    _, _, H, W = flow.shape
    if tuple(grad_output.shape) != (B, 2 * C if flipcat else C, H, W):
        raise ValueError("warp_backward: grad_output has the wrong shape")
    if grad_output.numel() == 0:
        return grad_feat, grad_flow
    with _on_device(feat) as stream:
        _lib.check(_lib.load().ffwm_warp_backward(
            _ptr(feat), _ptr(flow), _ptr(grad_output), _ptr(grad_feat), _ptr(grad_flow), B, C, Hi, Wi, H, W,
            (1 if flipcat else 0) | (2 if overwrite_feat and grad_feat is not None else 0), _dtype_code(feat), stream), "ffwm_warp_backward")
    return grad_feat, grad_flow


class _WarpProblem(ctypes.Structure):          # include/ffwm_hip.h: ffwm_warp_problem
    _fields_ = [("feat", ctypes.c_void_p), ("flow", ctypes.c_void_p), ("output", ctypes.c_void_p),
                ("grad_output", ctypes.c_void_p), ("grad_feat", ctypes.c_void_p), ("grad_flow", ctypes.c_void_p),
                ("B", ctypes.c_int64), ("C", ctypes.c_int64), ("Hi", ctypes.c_int64), ("Wi", ctypes.c_int64),
                ("H", ctypes.c_int64), ("W", ctypes.c_int64)]


def _warp_table(feats, flows, outs=None, grad_outputs=None, grad_feats=None, grad_flows=None):
    n = len(feats)
    arr = (_WarpProblem * n)()
    for i in range(n):
        B, C, Hi, Wi = feats[i].shape
        Bf, two, H, W = flows[i].shape
        if two != 2 or Bf != B:
            raise ValueError("warp_multi: flow %d must be [B,2,H,W] with the feature's batch" % i)
        a = arr[i]
        a.feat, a.flow = _ptr(feats[i]), _ptr(flows[i])
        a.output = _ptr(outs[i]) if outs is not None else None
        a.grad_output = _ptr(grad_outputs[i]) if grad_outputs is not None else None
        a.grad_feat = _ptr(grad_feats[i]) if grad_feats is not None and grad_feats[i] is not None else None
        a.grad_flow = _ptr(grad_flows[i]) if grad_flows is not None and grad_flows[i] is not None else None
        a.B, a.C, a.Hi, a.Wi, a.H, a.W = B, C, Hi, Wi, H, W
    return arr


def warp_multi_forward(feats, flows, flipcat=False):
    """Several independent warps (lists of equal length) in one or two launches; -> list of outputs."""
    _check("warp_multi_forward", *(list(feats) + list(flows)))
    outs = [f.new_empty((f.size(0), (2 if flipcat else 1) * f.size(1), fl.size(2), fl.size(3))) for f, fl in zip(feats, flows)]
    arr = _warp_table(feats, flows, outs=outs)
    with _on_device(feats[0]) as stream:
        _lib.check(_lib.load().ffwm_warp_multi_forward(ctypes.cast(arr, ctypes.c_void_p), len(feats), 1 if flipcat else 0,
                                                       _dtype_code(feats[0]), stream), "ffwm_warp_multi_forward")
    return outs


def warp_multi_backward(feats, flows, grad_outputs, flipcat, grad_feats, grad_flows):
    """grad_feats[i] / grad_flows[i]: zero-filled tensors to accumulate into, or None."""
    _check("warp_multi_backward", *(list(feats) + list(flows) + list(grad_outputs)))
    arr = _warp_table(feats, flows, grad_outputs=grad_outputs, grad_feats=grad_feats, grad_flows=grad_flows)
    with _on_device(feats[0]) as stream:
        _lib.check(_lib.load().ffwm_warp_multi_backward(ctypes.cast(arr, ctypes.c_void_p), len(feats), 1 if flipcat else 0,
                                                        _dtype_code(feats[0]), stream), "ffwm_warp_multi_backward")


# ---------------------------------------------------------------- guided filter
def guided_filter_forward(x, y, r, eps=1e-8):
    """-> (out, saved); GuidedFilter(r, eps)(x, y), reference external_function.py:239-277."""
    _check("guided_filter_forward", x, y)
    if x.shape != y.shape:
        raise ValueError("guided_filter_forward: x and y must have the same shape (c_x == c_y), got %s vs %s"
                         % (tuple(x.shape), tuple(y.shape)))
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    saved = x.new_empty((5, B, C, H, W))
    if out.numel() == 0:
        return out, saved
    with _on_device(x) as stream:
        _lib.check(_lib.load().ffwm_guided_filter_forward(
            _ptr(x), _ptr(y), _ptr(out), _ptr(saved), B * C, H, W, int(r), float(eps), _dtype_code(x), stream),
            "ffwm_guided_filter_forward")
    return out, saved


def guided_filter_backward(x, y, saved, grad_output, r):
    """-> grad_x (y is data and gets no gradient)."""
    _check("guided_filter_backward", x, y, grad_output)
    B, C, H, W = x.shape
    gx = torch.empty_like(x)
    if gx.numel() == 0:
        return gx
    ws = x.new_empty((2, B, C, H, W))
    with _on_device(x) as stream:
        _lib.check(_lib.load().ffwm_guided_filter_backward(
            _ptr(x), _ptr(y), _ptr(saved), _ptr(grad_output), _ptr(gx), _ptr(ws), B * C, H, W, int(r), _dtype_code(x), stream),
            "ffwm_guided_filter_backward")
    return gx


# ---------------------------------------------------------------- fused affine regularisation
def affine_regularization(flow, ktk, kernel_size, want_grad=True):
    """-> (loss, grad_flow or None): AffineRegularizationLoss(kz)(flow) of the reference (losses.py:200-219)
    and its gradient, one launch.  ktk = the module's K^T K matrix [kz^2, kz^2] in flow's dtype."""
    if flow.dim() != 4 or flow.size(1) != 2:
        raise ValueError("affine_regularization: flow must be [B,2,h,w]")
    if not flow.is_cuda:
        raise NotImplementedError("affine_regularization: ffwm_amd ops run on the GPU only (got a %s tensor)" % flow.device)
    flow = flow.contiguous()
    ktk = ktk.to(device=flow.device, dtype=flow.dtype).contiguous()
    B, _, h, w = flow.shape
    k = int(kernel_size)
    if ktk.numel() != k ** 4:
        raise ValueError("affine_regularization: ktk must hold kz^2 x kz^2 values")
    loss = torch.zeros((), device=flow.device, dtype=flow.dtype)
    grad = torch.zeros_like(flow) if want_grad else None
    scale = 1.0 / (B * (h - k + 1) * (w - k + 1))
    with _on_device(flow) as stream:
        _lib.check(_lib.load().ffwm_affine_regularization(
            _ptr(flow), _ptr(ktk), _ptr(loss), _ptr(grad), B, h, w, k, scale, _dtype_code(flow), stream),
            "ffwm_affine_regularization")
    return loss * scale, grad


# ---------------------------------------------------------------- correlation column maximum (MFMA)
def correlation_colmax(source, target):
    """torch.bmm(source[B,N,C], target[B,C,N]).max(dim=1)[0] without the [B,N,N] matrix (losses.py:347-353)."""
    if source.dim() != 3 or target.dim() != 3 or source.size(0) != target.size(0) or source.size(1) != target.size(2) \
            or source.size(2) != target.size(1):
        raise ValueError("correlation_colmax: need source [B,N,C] and target [B,C,N]")
    if not source.is_cuda:
        raise NotImplementedError("correlation_colmax: ffwm_amd ops run on the GPU only (got a %s tensor)" % source.device)
    if source.dtype != torch.float32 or target.dtype != torch.float32:
        raise TypeError("correlation_colmax: float32 only")
    source, target = source.contiguous(), target.contiguous()
    B, N, C = source.shape
    out = source.new_empty((B, N))
    with _on_device(source) as stream:
        _lib.check(_lib.load().ffwm_correlation_colmax(_ptr(source), _ptr(target), _ptr(out), B, N, C, _lib.F32, stream),
                   "ffwm_correlation_colmax")
    return out


# ---------------------------------------------------------------- conv weight gradient (MFMA)
def conv3x3_wgrad_supported(input, grad_output):
    """Shapes the MFMA weight-gradient kernel takes: float32 NCHW, W a multiple of 64, below 4 GiB."""
    return (input.dtype == torch.float32 and grad_output.dtype == torch.float32 and input.is_cuda
            and input.shape[3] % 64 == 0 and input.numel() * 4 < (1 << 32) - 64 and grad_output.numel() * 4 < (1 << 32) - 64)


def conv3x3_wgrad(input, grad_output, grad_weight=None, grad_bias=None):
    """grad_weight[K,C,3,3] of a 3x3 / stride 1 / pad 1 convolution (+= when given, zero-filled otherwise); a given
    (zero-filled) grad_bias[K] receives the bias gradient from the same pass."""
    _check("conv3x3_wgrad", input, grad_output, grad_weight)
    if grad_bias is not None and not (grad_bias.is_cuda and grad_bias.device == input.device and grad_bias.dtype == input.dtype
                                      and grad_bias.is_contiguous()):
        raise ValueError("conv3x3_wgrad: grad_bias must be a contiguous tensor on the input's device, same dtype")
    B, C, H, W = input.shape
    Bg, K, Hg, Wg = grad_output.shape
    if (Bg, Hg, Wg) != (B, H, W):
        raise ValueError("conv3x3_wgrad: grad_output %s does not match input %s" % (tuple(grad_output.shape), tuple(input.shape)))
    if grad_weight is None:
        grad_weight = input.new_zeros((K, C, 3, 3))
    elif tuple(grad_weight.shape) != (K, C, 3, 3):
        raise ValueError("conv3x3_wgrad: grad_weight has the wrong shape")
    if grad_bias is not None and tuple(grad_bias.shape) != (K,):
        raise ValueError("conv3x3_wgrad: grad_bias has the wrong shape")
    with _on_device(input) as stream:
        _lib.check(_lib.load().ffwm_conv3x3_wgrad(_ptr(input), _ptr(grad_output), _ptr(grad_weight), _ptr(grad_bias), B, C, K,
                                                  H, W, _dtype_code(input), stream), "ffwm_conv3x3_wgrad")
    return grad_weight


def conv2d_wgrad(rows, gathered, kernel, stride, pad, grad_weight=None):
    """grad_weight[K, C, k, k] of Conv2d (rows = grad_output, gathered = input) or [Ci, Co, 4, 4] of ConvTranspose2d(4, 2, 1)
    (rows = input, gathered = grad_output) on csrc/conv_bwd.hip; += into `grad_weight` when given, else a fresh tensor."""
    _check("conv2d_wgrad", rows, gathered, grad_weight)
    B, K, Ho, Wo = rows.shape
    Bg, C, H, W = gathered.shape
    if Bg != B:
        raise ValueError("conv2d_wgrad: batch sizes differ")
    if grad_weight is None:
        grad_weight = rows.new_zeros((K, C, kernel, kernel))
    elif tuple(grad_weight.shape) != (K, C, kernel, kernel):
        raise ValueError("conv2d_wgrad: grad_weight has the wrong shape")
    with _on_device(rows) as stream:
        _lib.check(_lib.load().ffwm_conv2d_wgrad(_ptr(rows), _ptr(gathered), _ptr(grad_weight), B, K, Ho, Wo, C, H, W, int(kernel),
                                                 int(stride), int(pad), _dtype_code(rows), stream), "ffwm_conv2d_wgrad")
    return grad_weight


def zero_fill(t):
    """Clear a contiguous CUDA tensor with the library's own kernel (ffwm_zero_fill; never a memset node: DESIGN 5.1)."""
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError("zero_fill: a contiguous CUDA tensor")
    with _on_device(t) as stream:
        _lib.check(_lib.load().ffwm_zero_fill(_ptr(t), t.numel() * t.element_size(), stream), "ffwm_zero_fill")
    return t


def conv2d_wgrad_tiled_ok(rows):
    """Shapes the tiled weight-gradient kernel takes: a plane of a multiple of 4 pixels, 16-byte aligned float32 rows."""
    return (rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 4 and (rows.shape[2] * rows.shape[3]) % 4 == 0
            and rows.data_ptr() % 16 == 0 and rows.numel() < (1 << 29))


def conv2d_wgrad_tiled(rows, gathered, kernel, stride, pad, want_bias=False, zeroed=None):
    """-> (grad_weight, grad_bias or None) of Conv2d (rows = grad_output, gathered = input: [K, C, k, k]; grad_bias = the row sums)
    or grad_weight [Ci, Co, 4, 4] of ConvTranspose2d(4, 2, 1) (rows = input, gathered = grad_output; no bias) on the tiled kernel of
    csrc/conv_bwd.hip.  Both results are slices of ONE buffer: a fresh one (the library clears it with one fill launch when it has
    to), or `zeroed` -- K C k k (+ K) ZERO floats the caller owns (a slice of the trainer's gradient arena): no fill launch at all."""
    _check("conv2d_wgrad_tiled", rows, gathered)
    B, K, Ho, Wo = rows.shape
    Bg, C, H, W = gathered.shape
    if Bg != B:
        raise ValueError("conv2d_wgrad_tiled: batch sizes differ")
    n = K * C * kernel * kernel
    total = n + (K if want_bias else 0)
    if zeroed is not None:
        if not (zeroed.is_cuda and zeroed.device == rows.device and zeroed.dtype == rows.dtype and zeroed.is_contiguous() and zeroed.numel() == total):
            raise ValueError("conv2d_wgrad_tiled: `zeroed` must be %d contiguous zero elements of the operands' device and dtype" % total)
        buf = zeroed.view(-1)
    else:
        buf = rows.new_empty((total,))
    gw = buf[:n].view(K, C, kernel, kernel)
    gb = buf[n:] if want_bias else None
    lib = _lib.load()
    with _on_device(rows) as stream:
        # (`prezeroed` is an ARGUMENT since ABI 5: rounds 4-5 toggled a process-global library option around the call -- ADVICE r5)
        _lib.check(lib.ffwm_conv2d_wgrad_tiled(_ptr(rows), _ptr(gathered), _ptr(gw), _ptr(gb), B, K, Ho, Wo, C, H, W, int(kernel),
                                               int(stride), int(pad), 1 if zeroed is not None else 0, _dtype_code(rows), stream),
                   "ffwm_conv2d_wgrad_tiled")
    return gw, gb


def conv3x3_winograd(x, weight, bias=None, data_gradient=False, act=0, slope=0.0, out=None, frozen=None, pre=None):
    """Conv2d(C, K, 3, 1, 1) forward (weight [K, C, 3, 3]) or its data gradient (x = grad_output [B, K_layer, H, W], weight =
    the layer's own [K_layer, C_layer, 3, 3]; returns [B, C_layer, H, W]) by fp32 Winograd F(2x2, 3x3) on the MFMA units
    (csrc/conv_winograd.hip).  act = 1 applies LeakyReLU(slope) after the bias (slope 0 = ReLU).  frozen: a dict OWNED BY THE
    LAYER (it must die with the weight tensor) in which the transformed weights of a frozen layer are kept between calls,
    keyed by direction, layout and the weight's version counter."""
    _check("conv3x3_winograd", x, weight, out)
    if bias is not None and (not bias.is_cuda or bias.dtype != x.dtype or not bias.is_contiguous()):
        raise ValueError("conv3x3_winograd: bias must be a contiguous tensor of the input's device and dtype")
    B, C, H, W = x.shape
    if data_gradient:
        if weight.shape[0] != C or tuple(weight.shape[2:]) != (3, 3):
            raise ValueError("conv3x3_winograd: weight %s does not belong to a grad_output with %d channels" % (tuple(weight.shape), C))
        K = weight.shape[1]
    else:
        if weight.shape[1] != C or tuple(weight.shape[2:]) != (3, 3):
            raise ValueError("conv3x3_winograd: weight %s does not fit an input with %d channels" % (tuple(weight.shape), C))
        K = weight.shape[0]
    if bias is not None and tuple(bias.shape) != (K,):
        raise ValueError("conv3x3_winograd: bias must have %d elements" % K)
    if out is None:
        out = x.new_empty((B, K, H, W))
    elif tuple(out.shape) != (B, K, H, W):
        raise ValueError("conv3x3_winograd: out has the wrong shape")
    lib = _lib.load()
    _sync_winograd_split_mode()
    mode = int(bool(data_gradient))
    ws = None
    if pre is not None:
        # transformed weights prepared beforehand for exactly this weight tensor, direction and width class (a dict the producer of
        # the weights attached to them: conv3x3_winograd_weights_multi)
        ws = pre.get((mode, W % 4 == 0))
        if ws is not None:
            mode |= 2
            frozen = None
    if ws is None and frozen is not None and frozen is not False:
        key = (mode, W % 4 == 0, weight._version, weight.data_ptr())
        ws = frozen.get(key)
        if ws is not None:
            mode |= 2
        else:
            for k in [k for k in frozen if k[2:] != key[2:]]:
                del frozen[k]         # transforms of an older version of the weights
    if ws is None:
        ws = x.new_empty((lib.ffwm_conv3x3_winograd_workspace_bytes(K, C) // 4,))
        if frozen is not None and frozen is not False:
            frozen[key] = ws
    with _on_device(x) as stream:
        _lib.check(lib.ffwm_conv3x3_winograd_forward(_ptr(x), _ptr(weight), _ptr(bias) if bias is not None else None, _ptr(out), _ptr(ws),
                                                     B, C, H, W, K, mode, int(act), float(slope), _dtype_code(x),
                                                     stream), "ffwm_conv3x3_winograd_forward")
    if frozen is not None and frozen is not False and not (mode & 2):
        # a transform that was just written into a layer's cache may be picked up by a call on ANOTHER stream (the trainer runs the
        # loss networks' passes on side streams): make it visible to every stream, once per frozen layer
        torch.cuda.current_stream(x.device).synchronize()
    return out


class _WinoWeightsItem(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("K", ctypes.c_int64), ("C", ctypes.c_int64),
                ("data_gradient", ctypes.c_int), ("width_multiple_of_4", ctypes.c_int)]


def conv3x3_winograd_weights_multi(items):
    """items: [(weight [Kl, Cl, 3, 3], data_gradient, width_multiple_of_4), ...] -> the transformed weights of every item (one
    tensor each, what conv3x3_winograd(..., pre=...) takes), all from ONE launch per 24 items (ffwm_conv3x3_winograd_weights_multi)."""
    if not items:
        return []
    lib = _lib.load()
    w0 = items[0][0]
    arr = (_WinoWeightsItem * len(items))()
    out = []
    for a, (w, dg, wm4) in zip(arr, items):
        if not (w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.device == w0.device):
            raise ValueError("conv3x3_winograd_weights_multi: contiguous float32 [K, C, 3, 3] weights on one GPU")
        K, C = (w.shape[1], w.shape[0]) if dg else (w.shape[0], w.shape[1])
        ws = w.new_empty((lib.ffwm_conv3x3_winograd_workspace_bytes(K, C) // 4,))
        a.weight, a.workspace, a.K, a.C, a.data_gradient, a.width_multiple_of_4 = w.data_ptr(), ws.data_ptr(), K, C, int(bool(dg)), int(bool(wm4))
        out.append(ws)
    with _on_device(w0) as stream:
        _lib.check(lib.ffwm_conv3x3_winograd_weights_multi(ctypes.cast(arr, ctypes.c_void_p), len(items), _lib.F32, stream),
                   "ffwm_conv3x3_winograd_weights_multi")
    return out


_WINO_SPLIT_CAPPED = False
_WINO_SPLIT_BEFORE_CAP = 1


def _sync_winograd_split_mode():
    """torch.use_deterministic_algorithms(True) caps the Winograd kernel's reduction split at TWO pieces: two partial sums meet in a
    zero-filled output by float atomics, and a two-term sum does not depend on the order -- four terms do.  The library
    option conv_wino_split (1 = up to four pieces, the default; 2 = capped; 0 = never split) is RAISED to the cap while the switch is on
    and put back to what it was when it goes off; with the switch off a value set by hand is never touched (the first call
    used to overwrite a hand-set 2 with 1).  A hipGraph captured before the switch changed keeps the split it was captured with:
    re-capture (FFWMTrainer.release_graphs()) after changing the switch."""
    global _WINO_SPLIT_CAPPED, _WINO_SPLIT_BEFORE_CAP
    want = bool(torch.are_deterministic_algorithms_enabled())
    if want == _WINO_SPLIT_CAPPED:
        return
    if want:
        prev = _lib.set_option("conv_wino_split", 2)
        _WINO_SPLIT_BEFORE_CAP = prev
        if prev == 0:
            _lib.set_option("conv_wino_split", 0)          # never split: already deterministic
    else:
        _lib.set_option("conv_wino_split", _WINO_SPLIT_BEFORE_CAP)
    _WINO_SPLIT_CAPPED = want


def conv3x3_winograd_splits(B, C, H, W, K, act=0):
    """In how many pieces conv3x3_winograd will cut the reduction of this call (1, 2 or 4): ffwm_conv3x3_winograd_splits."""
    _sync_winograd_split_mode()
    return int(_lib.load().ffwm_conv3x3_winograd_splits(int(B), int(C), int(H), int(W), int(K), int(act)))


# ---------------------------------------------------------------- LightCNN max-feature-map
def _mfm_dims(x):
    if x.dim() < 2 or x.shape[1] % 2:
        raise ValueError("mfm: need [B, 2C, ...], got %s" % (tuple(x.shape),))
    hw = 1
    for d in x.shape[2:]:
        hw *= d
    return x.shape[0], x.shape[1] // 2, hw


def _bias_ok(x, bias, n):
    if bias is None:
        return
    if not (bias.is_cuda and bias.dtype == x.dtype and bias.is_contiguous() and bias.numel() == n):
        raise ValueError("bias must be a contiguous float32 GPU vector of %d elements" % n)


# ---------------------------------------------------------------- residual-block tails / warp-attention gate
_ACT_CODES = {"lrelu": 1, "sigmoid": 3}


def _same_f32(name, *ts):
    t0 = ts[0]
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == t0.shape and t.device == t0.device):
            raise NotImplementedError("%s: contiguous float32 GPU tensors of one shape only" % name)


def add_act_forward(a, b, act, negative_slope=0.2):
    """act(a + b): the tail of ResidualBlock.forward (base_networks.py:207-233) as one pass; act in {"lrelu", "sigmoid"}."""
    _same_f32("add_act_forward", a, b)
    y = torch.empty_like(a)
    if y.numel():
        with _on_device(a) as stream:
            _lib.check(_lib.load().ffwm_add_act_forward(_ptr(a), _ptr(b), _ptr(y), a.numel(), _ACT_CODES[act], float(negative_slope),
                                                        _lib.F32, stream), "ffwm_add_act_forward")
    return y


def add_act_backward(y, grad_y, act, negative_slope=0.2):
    """grad_y * act'(a + b) from y = act(a + b): the gradient of a and of b."""
    _same_f32("add_act_backward", y, grad_y)
    dz = torch.empty_like(y)
    if dz.numel():
        with _on_device(y) as stream:
            _lib.check(_lib.load().ffwm_add_act_backward(_ptr(y), _ptr(grad_y), _ptr(dz), y.numel(), _ACT_CODES[act],
                                                         float(negative_slope), _lib.F32, stream), "ffwm_add_act_backward")
    return dz


def sigmoid_gate_forward(a, b, x):
    """-> (y, att): att = sigmoid(a + b), y = x * att (FFWM.forward's `skip * att_i(skip)`, base_networks.py:330-333)."""
    _same_f32("sigmoid_gate_forward", a, b, x)
    att, y = torch.empty_like(a), torch.empty_like(a)
    if y.numel():
        with _on_device(a) as stream:
            _lib.check(_lib.load().ffwm_sigmoid_gate_forward(_ptr(a), _ptr(b), _ptr(x), _ptr(att), _ptr(y), a.numel(), _lib.F32, stream),
                       "ffwm_sigmoid_gate_forward")
    return y, att


def sigmoid_gate_backward(x, att, grad_y):
    """-> (grad_z, grad_x): grad_z = the gradient of a and of b."""
    _same_f32("sigmoid_gate_backward", x, att, grad_y)
    dz, dx = torch.empty_like(x), torch.empty_like(x)
    if dz.numel():
        with _on_device(x) as stream:
            _lib.check(_lib.load().ffwm_sigmoid_gate_backward(_ptr(x), _ptr(att), _ptr(grad_y), _ptr(dz), _ptr(dx), x.numel(), _lib.F32,
                                                              stream), "ffwm_sigmoid_gate_backward")
    return dz, dx


def mfm_forward(x, bias=None):
    """max(x[:, :C] + bias[:C], x[:, C:] + bias[C:]) of a contiguous float32 [B, 2C, ...] tensor."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise NotImplementedError("mfm_forward: contiguous float32 GPU tensors only")
    B, C, HW = _mfm_dims(x)
    _bias_ok(x, bias, 2 * C)
    y = x.new_empty((B, C) + tuple(x.shape[2:]))
    if y.numel():
        with _on_device(x) as stream:
            _lib.check(_lib.load().ffwm_mfm_forward(_ptr(x), _ptr(bias), _ptr(y), B, C, HW, _lib.F32, stream), "ffwm_mfm_forward")
    return y


def mfm_backward(x, grad_y, bias=None):
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and grad_y.is_contiguous() and grad_y.dtype == x.dtype):
        raise NotImplementedError("mfm_backward: contiguous float32 GPU tensors only")
    B, C, HW = _mfm_dims(x)
    _bias_ok(x, bias, 2 * C)
    dx = torch.empty_like(x)
    if dx.numel():
        with _on_device(x) as stream:
            _lib.check(_lib.load().ffwm_mfm_backward(_ptr(x), _ptr(bias), _ptr(grad_y), _ptr(dx), B, C, HW, _lib.F32, stream),
                       "ffwm_mfm_backward")
    return dx


def bias_relu_forward(h, bias, out=None):
    """relu(h + bias[None, :, None, None]) of a contiguous float32 [B, C, ...] tensor; out may be h itself."""
    if not (h.is_cuda and h.dtype == torch.float32 and h.is_contiguous()):
        raise NotImplementedError("bias_relu_forward: contiguous float32 GPU tensors only")
    B, C = h.shape[0], h.shape[1]
    HW = h.numel() // max(B * C, 1)
    _bias_ok(h, bias, C)
    y = torch.empty_like(h) if out is None else out
    if y.numel():
        with _on_device(h) as stream:
            _lib.check(_lib.load().ffwm_bias_relu_forward(_ptr(h), _ptr(bias), _ptr(y), B, C, HW, _lib.F32, stream),
                       "ffwm_bias_relu_forward")
    return y


# ---------------------------------------------------------------- fused L1 terms (csrc/l1_loss.hip)
class _L1Problem(ctypes.Structure):            # include/ffwm_hip.h: ffwm_l1_problem
    _fields_ = [("x", ctypes.c_void_p), ("y", ctypes.c_void_p), ("mask", ctypes.c_void_p), ("grad_x", ctypes.c_void_p),
                ("n", ctypes.c_int64), ("chw", ctypes.c_int64), ("hw", ctypes.c_int64), ("scale", ctypes.c_double),
                ("slot", ctypes.c_int)]


def _l1_table(terms, grads=None):
    """terms: [(x, y, mask or None, segments)], segments = [(x_row0, y_row0, rows, scale, slot)]: rows x_row0 .. x_row0 + rows of x
    against rows y_row0 .. of y (and of the mask, indexed like x).  One kernel problem per segment."""
    segs = []
    for i, (x, y, m, segments) in enumerate(terms):
        if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous() and y.dtype == x.dtype
                and y.shape[1:] == x.shape[1:]):
            raise ValueError("l1_multi: term %d needs contiguous float32 GPU tensors with equal trailing shapes" % i)
        row = x.numel() // max(x.shape[0], 1)
        if m is not None:
            if not (m.is_contiguous() and m.dtype == x.dtype and m.dim() == x.dim() and m.shape[0] == x.shape[0]
                    and m.shape[2:] == x.shape[2:] and m.shape[1] in (1, x.shape[1])):
                raise ValueError("l1_multi: term %d: the mask must be [B, 1 or C, ...] over x" % i)
            mrow = m.numel() // m.shape[0]
        for (x0, y0, rows, scale, slot) in segments:
            if x0 < 0 or y0 < 0 or x0 + rows > x.shape[0] or y0 + rows > y.shape[0]:
                raise ValueError("l1_multi: term %d: segment outside the tensors" % i)
            segs.append((x.data_ptr() + 4 * x0 * row, y.data_ptr() + 4 * y0 * row, None if m is None else m.data_ptr() + 4 * x0 * mrow,
                         None if grads is None or grads[i] is None else grads[i].data_ptr() + 4 * x0 * row,
                         rows * row, row if m is not None else 1, mrow if m is not None else 1, float(scale), int(slot)))
    arr = (_L1Problem * len(segs))()
    for a, sg in zip(arr, segs):
        a.x, a.y, a.mask, a.grad_x, a.n, a.chw, a.hw, a.scale, a.slot = sg
    return arr, len(segs)


def l1_multi_forward(terms, n_slots):
    """terms as in _l1_table -> out[n_slots] with out[slot] = sum over the segments of scale * sum |x m - y m| (one launch)."""
    out = torch.zeros(n_slots, device=terms[0][0].device, dtype=torch.float32)
    arr, n = _l1_table(terms)
    with _on_device(terms[0][0]) as stream:
        _lib.check(_lib.load().ffwm_l1_multi(ctypes.cast(arr, ctypes.c_void_p), n, _ptr(out), None, n_slots, _lib.F32, stream),
                   "ffwm_l1_multi")
    return out


def l1_multi_backward(terms, grad_out, need):
    """-> [grad_x or None per term]: grad_out[slot] * scale * sign(x m - y m) * m (one launch for all terms)."""
    sel = [i for i, nd in enumerate(need) if nd]
    if not sel:
        return [None] * len(terms)
    # rows no segment covers get no gradient: such a tensor starts from zeros
    grads = []
    for i in sel:
        x, segments = terms[i][0], terms[i][3]
        # the kernel WRITES a segment's rows (it does not accumulate): segments of one tensor must not overlap; rows none of them covers
        # get no gradient, so unless the segments tile the tensor exactly it starts from zeros
        spans = sorted((s0, s0 + r) for (s0, _, r, _, _) in segments)
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            if b0 < a1:
                raise ValueError("l1_multi: term %d: segments [%d, %d) and [%d, %d) overlap (a row would keep only one gradient)"
                                 % (i, a0, a1, b0, b1))
        covered = bool(spans) and spans[0][0] == 0 and spans[-1][1] == x.shape[0] and all(a1 == b0 for (_, a1), (b0, _) in zip(spans, spans[1:]))
        grads.append(torch.empty_like(x) if covered else torch.zeros_like(x))
    arr, n = _l1_table([terms[i] for i in sel], grads)
    go = grad_out.contiguous()
    with _on_device(go) as stream:
        _lib.check(_lib.load().ffwm_l1_multi(ctypes.cast(arr, ctypes.c_void_p), n, None, _ptr(go), go.numel(), _lib.F32, stream),
                   "ffwm_l1_multi")
    out = [None] * len(terms)
    for i, g in zip(sel, grads):
        out[i] = g
    return out
