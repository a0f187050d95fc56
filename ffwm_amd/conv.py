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

from . import ops


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
