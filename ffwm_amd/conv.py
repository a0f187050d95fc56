"""Routing of conv weight gradients to the hand-written MFMA kernel (csrc/conv_wgrad.hip).

The conv stacks of the reference (models/base_networks.py:59-165 FlowNet, :274-347 FFWM) are plain
``nn.Conv2d`` layers; their forward and data gradient stay with the vendor library (Winograd fp32, ~100
TFLOP/s effective on MI355X), but the weight gradient of the large-image 3x3 layers -- above all dres2's
195 -> 195 channels at 128 x 128, 1.7 ms per call in the vendor library -- is computed by
``ffwm_conv3x3_wgrad``.  ``route_conv_wgrad(net)`` re-classes the eligible layers in place, so parameter
names, state dicts and spectral-norm hooks are untouched.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext, ops


class _Conv3x3MfmaWgrad(Function):
    """F.conv2d(x, w, b, stride 1, padding 1) whose backward takes grad_weight from the MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.ops.aten.convolution(x, weight, bias, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = gb = gw = None
        if need_w:
            # the bias gradient is a row sum of the operand the MFMA kernel streams anyway
            if need_b:
                gb = torch.zeros(weight.shape[0], device=go.device, dtype=go.dtype)
            gw = ops.conv3x3_wgrad(x if x.is_contiguous() else x.contiguous(), go, None, gb)
        if need_x or (need_b and gb is None):
            gx, _, gb2 = torch.ops.aten.convolution_backward(
                go, x, weight, [weight.shape[0]] if (need_b and gb is None) else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                [bool(need_x), False, bool(need_b and gb is None)])
            if gb is None:
                gb = gb2
        return gx, gw, gb


def wgrad_route_ok(x, weight):
    """Layers whose weight gradient the MFMA kernel computes faster than the vendor library on MI355X
    (tools/exp_conv_micro.py): fp32, >= 64 channels on both sides, image rows a multiple of 64 pixels, and
    either a 128-pixel-wide image or a channel count the vendor kernels tile badly (not a multiple of 64); plus the
    layers between an RGB image and >= 64 feature channels at 128 x 128 (3 channels x 9 taps fit one MFMA tile)."""
    K, C = weight.shape[0], weight.shape[1]
    W = x.shape[3]
    wide = min(C, K) >= 64 and (W >= 128 or C % 64 != 0 or K % 64 != 0)
    rgb = min(C, K) <= 3 and max(C, K) >= 64 and W >= 128          # image <-> features: the packed-tap variant alone
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and W % 64 == 0 and (wide or rgb) and ops.conv3x3_wgrad_supported(x, x))


class MfmaWgradConv2d(nn.Conv2d):
    """nn.Conv2d (3x3, stride 1, padding 1, no dilation / groups) with the routed weight gradient."""

    def _conv_forward(self, input, weight, bias):
        if torch.is_grad_enabled() and weight.requires_grad and wgrad_route_ok(input, weight):
            ext = _ext.get()
            if ext is not None:
                return ext.conv3x3_mfma_wgrad(input, weight, bias)
            return _Conv3x3MfmaWgrad.apply(input, weight, bias)
        return super()._conv_forward(input, weight, bias)


def eligible(m):
    return (type(m) is nn.Conv2d and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"
            and (min(m.in_channels, m.out_channels) >= 64
                 or (min(m.in_channels, m.out_channels) <= 3 and max(m.in_channels, m.out_channels) >= 64)))


def route_conv_wgrad(net):
    """Re-class every eligible nn.Conv2d of `net` in place; returns the number of layers routed."""
    n = 0
    for m in net.modules():
        if eligible(m):
            m.__class__ = MfmaWgradConv2d
            n += 1
    return n


# ================================================================================= MFMA forward (csrc/conv_fwd.hip)
# The layers MIOpen serves through NHWC implicit-GEMM kernels wrapped in layout transposes and zero-fills -- stride-2
# convolutions, 4x4 / stride-2 transposed convolutions, small-plane 3x3 layers with awkward channel counts (FlowNet's
# 1026 / 770 / 386-channel concatenations: 125-160 us through the vendor path, 26-30 us here) -- run their FORWARD on the
# hand-written fp32 MFMA kernel, in training too.  The 4x4 / stride-2 family is closed under differentiation with
# respect to the input (the data gradient of the convolution is the transposed convolution with the same weight tensor
# and vice versa), so those data gradients run on the same kernel; everything else in backward stays with ATen.
def _conv_fwd_call(x, weight, bias, stride, pad, transposed):
    from .flownet_eval import conv_mfma, NONE
    return conv_mfma(x, weight, bias, stride, pad, transposed, NONE)


class _MfmaConv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, bias is not None)
        return _conv_fwd_call(x.contiguous(), weight.contiguous(), bias, stride, pad, False)

    @staticmethod
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        gx = None
        k = weight.size(2)
        if need_x and k == 4 and stride == 2 and pad == 1 and x.size(2) == 2 * go.size(2) and x.size(3) == 2 * go.size(3):
            # d(input) of Conv2d(C, K, 4, 2, 1) = ConvTranspose2d(K, C, 4, 2, 1) with the SAME weight tensor [K, C, 4, 4]
            gx = _conv_fwd_call(go, weight.contiguous(), None, 2, 1, True)
            need_x = False
        gxa, gw, gb = torch.ops.aten.convolution_backward(
            go, x, weight, [weight.size(0)] if has_bias else None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
            [need_x, need_w, need_b])
        return (gx if gx is not None else gxa), gw, gb, None, None


class _MfmaConvTranspose2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return _conv_fwd_call(x.contiguous(), weight.contiguous(), bias, 2, 1, True)

    @staticmethod
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = None
        if need_x:
            # d(input) of ConvTranspose2d(C, K, 4, 2, 1) = Conv2d(K, C, 4, 2, 1) with the same weight tensor [C, K, 4, 4]
            gx = _conv_fwd_call(go, weight.contiguous(), None, 2, 1, False)
        _, gw, gb = torch.ops.aten.convolution_backward(
            go, x, weight, [weight.size(1)] if ctx.has_bias else None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1,
            [False, need_w, need_b])
        return gx, gw, gb


def fwd_route_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and x.numel() < (1 << 29) and weight.numel() < (1 << 29))


class MfmaFwdConv2d(nn.Conv2d):
    """nn.Conv2d (3x3 or 4x4, stride 1 or 2) whose forward runs on csrc/conv_fwd.hip."""

    def _conv_forward(self, input, weight, bias):
        if fwd_route_ok(input, weight) and (self.stride[0] == 2 or input.size(2) <= 32):
            ext = _ext.get()
            if ext is not None:
                return ext.conv2d(input, weight, bias, self.stride[0], self.padding[0])
            return _MfmaConv2d.apply(input, weight, bias, self.stride[0], self.padding[0])
        return super()._conv_forward(input, weight, bias)


class MfmaFwdConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(C, K, 4, 2, 1) whose forward and data gradient run on csrc/conv_fwd.hip."""

    def forward(self, input, output_size=None):
        if output_size is None and fwd_route_ok(input, self.weight):
            ext = _ext.get()
            if ext is not None:
                return ext.conv_transpose2d(input, self.weight, self.bias)
            return _MfmaConvTranspose2d.apply(input, self.weight, self.bias)
        return super().forward(input, output_size)


import os as _os
_POLICY = _os.environ.get("FFWM_FWD_ROUTE", "all")


def fwd_eligible(m):
    if _POLICY == "none":
        return False
    if type(m) is nn.Conv2d and _POLICY in ("odd", "oddT") and m.in_channels % 64 == 0:
        return False
    if type(m) is nn.ConvTranspose2d and _POLICY == "odd":
        return False
    if type(m) is nn.Conv2d and _POLICY == "T":
        return False
    if type(m) is nn.Conv2d:
        return (m.kernel_size in ((3, 3), (4, 4)) and m.stride in ((1, 1), (2, 2)) and m.padding[0] == m.padding[1]
                and m.padding[0] < m.kernel_size[0] and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"
                and m.in_channels >= 32 and m.out_channels >= 32
                and (m.stride == (2, 2) or m.kernel_size == (3, 3)))
    if type(m) is nn.ConvTranspose2d:
        return (m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1) and m.output_padding == (0, 0)
                and m.dilation == (1, 1) and m.groups == 1 and m.in_channels >= 32 and m.out_channels >= 32)
    return False


def route_conv_fwd(net):
    """Re-class every eligible convolution of `net` in place (layers already routed for their weight gradient keep that
    class: they are the large stride-1 layers the vendor's Winograd kernels serve well); returns the number routed."""
    n = 0
    for m in net.modules():
        if fwd_eligible(m):
            m.__class__ = MfmaFwdConv2d if isinstance(m, nn.Conv2d) else MfmaFwdConvTranspose2d
            n += 1
    return n
