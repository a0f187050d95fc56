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


import os as _os0
_WGRAD_W64 = _os0.environ.get("FFWM_WGRAD_W64", "1") == "1"      # round 6: on -- the 64-pixel layers with aligned channel counts (att convs of the 64 x 64 level) reach the Winograd-domain kernel: warp + attention sub-path 1902 -> 1940 img/s, train step unchanged


class _Conv3x3MfmaWgrad(Function):
    """F.conv2d(x, w, b, stride 1, padding 1) whose backward takes grad_weight from the MFMA kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.ops.aten.convolution(x, weight, bias, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = gb = gw = None
        if need_w:
            gw, gb = conv_weight_grad(x if x.is_contiguous() else x.contiguous(), go, weight, 1, 1, need_b)
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
    wide = min(C, K) >= 64 and (W >= 128 or C % 64 != 0 or K % 64 != 0 or _WGRAD_W64)
    rgb = min(C, K) <= 3 and max(C, K) >= 64 and W >= 128          # image <-> features: the packed-tap variant alone
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled()
            and W % 64 == 0 and (wide or rgb) and ops.conv3x3_wgrad_supported(x, x))


class MfmaWgradConv2d(nn.Conv2d):
    """nn.Conv2d (3x3, stride 1, padding 1, no dilation / groups) with the routed weight gradient."""

    def _conv_forward(self, input, weight, bias):
        if torch.is_grad_enabled() and weight.requires_grad and wgrad_route_ok(input, weight):
            return _Conv3x3MfmaWgrad.apply(input, weight, bias)
        if (torch.is_grad_enabled() and weight.requires_grad and _tiled_wgrad_layer_ok(self)
                and input.is_cuda and input.dtype == torch.float32 and not torch.is_autocast_enabled()):
            return _MfmaConv2d.apply(input, weight, bias, self.stride[0], self.padding[0], False)      # vendor forward, own weight gradient
        return super()._conv_forward(input, weight, bias)


def _tiled_wgrad_layer_ok(m):
    if m.kernel_size == (1, 1):
        # the 1x1 shortcut of a residual block: through the vendor library its weight gradient is an NHWC implicit GEMM behind two
        # layout transposes, a fill and a separate bias-gradient reduction (296 us at 195 -> 195, 128 x 128, batch 8); one launch here
        return (_TILED_WGRAD and _TILED_WGRAD_1X1 and m.stride == (1, 1) and m.padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1
                and min(m.in_channels, m.out_channels) >= 32)
    return (_TILED_WGRAD and m.kernel_size in ((3, 3), (4, 4)) and m.stride in ((1, 1), (2, 2)) and m.padding[0] == m.padding[1]
            and m.padding[0] < m.kernel_size[0] and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros")


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


# ================================================================================= weight gradients: one chooser for every routed layer
# Measured per layer of the FFWM train step on an MI355X (tools/bwd_layers.py, profiles/r03_bwd_layers.txt), the whole aten op --
# NHWC implicit GEMM + its layout transposes and fills -- against the hand-written kernels:
#   * csrc/conv_wgrad.hip (3x3 / stride 1, 64-pixel rows): 2-5 x the vendor on the 128-pixel layers (wgrad_route_ok);
#   * csrc/conv_bwd.hip, tiled variant: faster than or level with the vendor on every other 3x3 / 4x4 layer of the step (2-3 x on
#     FlowNet's 2 x 2 ... 8 x 8 tail and on the thin heads, level on the 256-384 channel layers at 32 x 32), one launch + at most
#     one memset instead of five to eight launches, and the bias gradient comes out of the same pass;
#   * left with the vendor: image heads (<= 4 output channels) on large planes, where a 64-row MFMA tile is mostly empty.
import os as _os1
_TILED_WGRAD = _os1.environ.get("FFWM_TILED_WGRAD", "1") != "0"
_TILED_WGRAD_1X1 = _os1.environ.get("FFWM_TILED_WGRAD_1X1", "1") != "0"


def _tiled_wgrad_wins(rows, gathered, kernel):
    K, P = rows.shape[1], rows.shape[2] * rows.shape[3]
    if not (_TILED_WGRAD and kernel in (1, 3, 4) and ops.conv2d_wgrad_tiled_ok(rows) and gathered.numel() < (1 << 29)):
        return False
    if kernel == 1 and (min(K, gathered.shape[1]) < 32 or not _TILED_WGRAD_1X1):
        return False             # thin 1x1 layers stay with the vendor's GEMM
    if K <= 4 and (P >= 16384 or (P >= 4096 and gathered.shape[1] >= 128)):
        return False
    return True


class _GradArena:
    """Weight-gradient staging of ONE train step: every hand-written weight-gradient kernel of the step adds into zeroed memory (atomics
    of pixel slices / k tiles), and until round 5 each call cleared its own buffer -- ~270 fill launches per step (torch.zeros here, the
    library's zero-fill in the tiled kernel), 4-8 us each for a few hundred KiB.  Inside `begin()` .. `end()` the buffers are consecutive
    slices of one tensor that ONE launch clears at the start of the step; the slices are handed to autograd as the layers' gradients.
    Who may keep a slice: with the trainers' flat gradient buffers AccumulateGrad ADDS a slice into the parameter's own view and drops
    it; in a reducer's gather mode (param.grad is None during backward) autograd INSTALLS the slice as param.grad until the bucket is
    packed -- which always happens before the next begin(), the only moment a slice's memory is reused (the trainers wrap the
    step in try / finally so that end() also runs when a step raises).  The first step measures
    the demand with the old per-call fills; a step that asks for more than the arena holds falls back per call, and the arena grows
    behind it (never while a stream is capturing; retired buffers stay alive: a captured graph may still clear and use them)."""

    def __init__(self):
        self.buf, self.off, self.need, self.active, self.retired = None, 0, 0, False, []

    def begin(self, device):
        self.off = self.need = 0
        if self.buf is not None and self.buf.device != device:
            self.retired.append(self.buf)
            self.buf = None
        if self.buf is not None:
            ops.zero_fill(self.buf)
        self.active = True
        self.device = device

    def take(self, n, like):
        if not self.active or like.device != self.device or like.dtype != torch.float32:
            return None
        n_al = (n + 63) & ~63
        self.need += n_al
        if self.buf is None or self.off + n_al > self.buf.numel():
            return None
        s = self.buf[self.off:self.off + n]
        self.off += n_al
        return s

    def end(self):
        if not self.active:
            return
        self.active = False
        if (self.buf is None or self.need > self.buf.numel()) and self.need > 0 and not torch.cuda.is_current_stream_capturing():
            if self.buf is not None:
                self.retired.append(self.buf)
            self.buf = torch.zeros(self.need + self.need // 8, device=self.device, dtype=torch.float32)


GRAD_ARENA = _GradArena()


def conv_weight_grad(x, go, weight, stride, pad, need_b):
    """-> (grad_weight, grad_bias or None) of Conv2d(C, K, k, stride, pad) from its input and grad_output."""
    k = weight.shape[2]
    if k == 3 and stride == 1 and pad == 1 and wgrad_route_ok(x, weight):
        # grad_weight and grad_bias (a row sum of the operand the kernel streams anyway) as slices of ONE zero-filled buffer: one fill,
        # or none inside a trainer's step (GRAD_ARENA)
        n = weight.numel()
        total = n + (weight.shape[0] if need_b else 0)
        buf = GRAD_ARENA.take(total, go)
        if buf is None:
            buf = torch.zeros(total, device=go.device, dtype=go.dtype)
        gw, gb = buf[:n].view_as(weight), (buf[n:] if need_b else None)
        ops.conv3x3_wgrad(x, go, gw, gb)
        return gw, gb
    if _tiled_wgrad_wins(go, x, k) and weight.shape[2] == weight.shape[3]:
        return ops.conv2d_wgrad_tiled(go, x, k, stride, pad, want_bias=bool(need_b),
                                      zeroed=GRAD_ARENA.take(weight.numel() + (weight.shape[0] if need_b else 0), go))
    _, gw, gb = torch.ops.aten.convolution_backward(go, x, weight, [weight.shape[0]] if need_b else None, [stride, stride], [pad, pad],
                                                    [1, 1], False, [0, 0], 1, [False, True, bool(need_b)])
    return gw, gb


def conv_transpose_weight_grad(x, go, weight, need_b):
    """-> (grad_weight [Ci, Co, 4, 4], grad_bias or None) of ConvTranspose2d(Ci, Co, 4, 2, 1): the same pixel sum as a Conv2d weight
    gradient with the two tensors' roles swapped."""
    if _tiled_wgrad_wins(x, go, 4):
        gw, _ = ops.conv2d_wgrad_tiled(x, go, 4, 2, 1, zeroed=GRAD_ARENA.take(weight.numel(), go))
        return gw, (_bias_grad(go) if need_b else None)
    _, gw, gb = torch.ops.aten.convolution_backward(go, x, weight, [weight.shape[1]] if need_b else None, [2, 2], [1, 1], [1, 1], True,
                                                    [0, 0], 1, [False, True, bool(need_b)])
    return gw, gb


# ================================================================================= Winograd on MFMA (csrc/conv_winograd.hip)
# Forward and data gradient of the 3x3 / stride-1 layers by fp32 Winograd F(2x2, 3x3) with the 16 per-position GEMMs on the
# MFMA units: 1.5-1.9 x the vendor's VALU Winograd on the layers that dominate the train step (netG's 195 -> 195 residual
# blocks at 128^2 / 64^2: 0.62 ms against 0.93 ms; 256 channels at 128^2: 0.71 against 1.32 ms).  Small planes stay with the
# vendor: below ~2000 tiles the 64 x 64-tile workgroups do not fill the chip.
import os as _os
_WINOGRAD = _os.environ.get("FFWM_WINOGRAD", "1") != "0"
WINOGRAD_MIN_TILES = int(_os.environ.get("FFWM_WINOGRAD_MIN_TILES", 2048))
# (strips of 64 tiles) x (tiles of 64 output channels) a call must offer the 256 persistent workgroups: measured per shape of
# the train step (tools/wino_layers.py), below ~160 pairs the vendor's kernel is as fast or faster in isolation (128 -> 128 @32²: 52 vs
# 30 us); inside the captured multi-stream step 100 pairs measure 0.2-0.3 ms per step better (42.33 vs 42.60 ms) by moving 49 more
# small launches from the vendor's kernel to this one at 0.2-0.3 of its peak -- within the noise of the step in round 4.  Round 5 (late):
# the kernel's launches are 8-13 % shorter (epilogue, chunk loop) and 100 pairs measure 37.42 against 37.54 ms in four alternating runs on
# one box (tools/ab/abenv.sh) -- 0.3 % of the step, bought with 44 more launches per step at 0.2-0.3 of the kernel's peak, which drag the
# step-wide average of the Winograd scope (bench.py's roofline_mfma row, the number the rounds are compared on) from 0.58 to 0.53.  The
# threshold stays at 160; FFWM_WINOGRAD_MIN_PAIRS=100 is the faster step.
WINOGRAD_MIN_PAIRS = int(_os.environ.get("FFWM_WINOGRAD_MIN_PAIRS", 160))


def _winograd_dir_ok(x, c_red, k_out, act=0):
    tiles = x.shape[0] * ((x.shape[2] + 1) // 2) * ((x.shape[3] + 1) // 2)
    if not (min(c_red, k_out) >= 32 and tiles >= WINOGRAD_MIN_TILES and x.shape[0] * max(c_red, k_out) * x.shape[2] * x.shape[3] < (1 << 29)):
        return False
    pairs = ((tiles + 63) // 64) * ((k_out + 63) // 64)
    if pairs < WINOGRAD_MIN_PAIRS and act == 0:
        # few pairs: the library cuts the reduction over 2 / 4 workgroups (256 -> 256 at 32 x 32, batch 8: 128 pairs x 2)
        pairs *= ops.conv3x3_winograd_splits(x.shape[0], c_red, x.shape[2], x.shape[3], k_out, act)
    return pairs >= WINOGRAD_MIN_PAIRS


def winograd_dirs(x, weight, act=0):
    """(forward, data gradient): which directions of Conv2d(C, K, 3, 1, 1) on this input run on the Winograd kernel.  act: the
    activation fused into the FORWARD call (the data gradient never has one)."""
    if not (_WINOGRAD and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and not torch.is_autocast_enabled()):          # (under autocast the fp32 kernels stand aside: the vendor path casts)
        return False, False
    K, C = weight.shape[0], weight.shape[1]
    if K <= 4 and C >= 32:
        # an image head (netG's 195 -> 3 output layer): the forward is the thin direct kernel of conv_winograd.hip alone (a lane
        # owns 4 pixels x K channels), measured 50 us against the vendor's 137 at 128 x 128, batch 8; its data gradient stays
        return x.shape[3] % 4 == 0 and x.shape[0] * x.shape[2] * x.shape[3] >= 65536 and x.numel() < (1 << 29), False
    return _winograd_dir_ok(x, C, K, act), _winograd_dir_ok(x, K, C)


def winograd_ok(x, weight, act=0):
    return winograd_dirs(x, weight, act)[0]


# Weight transforms prepared ahead of the calls.  A producer that knows every weight of a network at once (the batched spectral norm)
# tags each product with its layer (`_ffwm_wino_owner`) and, from the second forward on, with the transforms the layer's calls used
# last time (`_ffwm_wino`: {(data_gradient, W % 4 == 0): workspace}, all of a network's in one launch:
# ops.conv3x3_winograd_weights_multi).  A call notes what it uses on the owner and takes its transform from the dict.
_WINO_BATCH = _os.environ.get("FFWM_WINO_BATCH", "1") != "0"


def _note_winograd_use(weight, x, dirs, need_dgrad):
    owner = getattr(weight, "_ffwm_wino_owner", None)
    if owner is None or not _WINO_BATCH:
        return None
    keys = owner.__dict__.setdefault("_wino_keys", set())
    wm4 = x.shape[3] % 4 == 0
    if dirs[0]:
        keys.add((0, wm4))
    if dirs[1] and need_dgrad:
        keys.add((1, wm4))
    return getattr(weight, "_ffwm_wino", None)


class _WinogradConv3x3(Function):
    """Conv2d(C, K, 3, 1, 1): forward and d(input) on csrc/conv_winograd.hip, d(weight) on csrc/conv_wgrad.hip where that
    kernel serves the shape (wgrad_route_ok), else the vendor's."""

    @staticmethod
    def forward(ctx, x, weight, bias, frozen=None, dirs=(True, True)):
        pre = _note_winograd_use(weight, x, dirs, ctx.needs_input_grad[0])          # transforms prepared with the weights (spectral_norm.SpectralNormGroup), or None
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.frozen = frozen
        ctx.own_dgrad = dirs[1]
        ctx.wino_pre = pre
        if dirs[0]:
            return ops.conv3x3_winograd(x, weight, bias, frozen=frozen, pre=pre)
        return torch.ops.aten.convolution(x, weight, bias, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = None
        if need_x and ctx.own_dgrad:
            gx = ops.conv3x3_winograd(go, weight, None, data_gradient=True, frozen=ctx.frozen, pre=ctx.wino_pre)
        elif need_x:
            gx = torch.ops.aten.convolution_backward(go, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if need_w:
            gw, gb = conv_weight_grad(x, go, weight, 1, 1, need_b)
        if need_b and gb is None:
            gb = _bias_grad(go)
        return gx, gw, gb, None, None


def frozen_cache(owner, weight):
    """The dict in which a layer with frozen weights keeps its transformed weights (None for a trainable layer): it lives on
    the module, so it dies with it -- a cache keyed by the storage address alone would outlive the tensor.  Only a frozen
    nn.Parameter qualifies: a spectral-norm product is a fresh plain tensor per forward (requires_grad False under no_grad,
    version counter 0, an address the caching allocator hands out again), so its transform must never be kept."""
    if (not isinstance(weight, nn.Parameter)) or weight.requires_grad or hasattr(owner, "weight_orig"):
        return None
    if getattr(weight, "_ffwm_flat_adam", False):
        return None      # a parameter of a flat optimizer (netD while the G step freezes it): updated without a version bump
    if torch.cuda.is_current_stream_capturing():
        return None      # a transform kernel that was only captured, not run, must not be recorded as valid
    return owner.__dict__.setdefault("_winograd_frozen", {})


class _WinogradConvBiasReLU(Function):
    """relu(Conv2d(C, K, 3, 1, 1)(x) + bias) of a FROZEN layer (VGG19: models/losses.py:398-519) as one launch: the bias and the
    activation are the Winograd kernel's epilogue, the transformed weights are kept between calls; backward = the mask from
    the saved output, then the data gradient on the same kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias, frozen, own_dgrad):
        x = x.contiguous()
        y = ops.conv3x3_winograd(x, weight, bias, act=1, slope=0.0, frozen=frozen)
        ctx.save_for_backward(weight, y, x if not own_dgrad else None)
        ctx.frozen = frozen
        ctx.own_dgrad = own_dgrad
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_y):
        weight, y, x = ctx.saved_tensors
        gh = torch.ops.aten.threshold_backward(grad_y.contiguous(), y, 0)
        if ctx.own_dgrad:
            return ops.conv3x3_winograd(gh, weight, None, data_gradient=True, frozen=ctx.frozen), None, None, None, None
        gx = torch.ops.aten.convolution_backward(gh, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        return gx, None, None, None, None


def winograd_bias_relu(x, layer):
    """relu(layer(x)) for a FROZEN nn.Conv2d(C, K, 3, 1, 1) with a bias; the caller has checked winograd_ok(x, layer.weight)."""
    cache = frozen_cache(layer, layer.weight)
    if torch.is_grad_enabled() and x.requires_grad:
        return _WinogradConvBiasReLU.apply(x, layer.weight, layer.bias, cache, winograd_dirs(x, layer.weight, 1)[1])
    return ops.conv3x3_winograd(x.contiguous(), layer.weight, layer.bias, act=1, slope=0.0, frozen=cache)


def winograd_conv(x, layer, bias=None):
    """conv3x3(x, layer.weight) + bias on the Winograd kernel with autograd; the caller has checked winograd_ok(x, layer.weight)."""
    weight = layer.weight
    cache = frozen_cache(layer, weight)
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return _WinogradConv3x3.apply(x, weight, bias, cache, winograd_dirs(x, weight))
    return ops.conv3x3_winograd(x.contiguous(), weight.contiguous(), bias, frozen=cache)


class WinogradConv2d(MfmaWgradConv2d):
    """nn.Conv2d (3x3, stride 1, padding 1) on the Winograd MFMA kernel when the plane is large enough (winograd_ok), else
    whatever MfmaWgradConv2d does with it."""

    def _conv_forward(self, input, weight, bias):
        dirs = winograd_dirs(input, weight)
        grad = torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad)
        if dirs[0] or (grad and dirs[1] and input.requires_grad):
            # (weight may be a spectral-norm product, not self.weight: only a frozen PARAMETER keeps its transform)
            cache = frozen_cache(self, weight) if weight is self._parameters.get("weight") else None
            if grad:
                return _WinogradConv3x3.apply(input, weight, bias, cache, dirs)
            return ops.conv3x3_winograd(input.contiguous(), weight.contiguous(), bias, frozen=cache,
                                        pre=_note_winograd_use(weight, input, (True, False), False))
        if self.__dict__.get("_mfma_fwd_small") and fwd_route_ok(input, weight) and input.size(2) <= 32:
            return _MfmaConv2d.apply(input, weight, bias, 1, 1)          # small planes: the direct MFMA kernel (route_conv_fwd)
        return super()._conv_forward(input, weight, bias)


def winograd_eligible(m):
    return (type(m) in (nn.Conv2d, MfmaWgradConv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
            and m.dilation == (1, 1) and m.groups == 1 and m.padding_mode == "zeros"
            and (min(m.in_channels, m.out_channels) >= 32 or (m.out_channels <= 4 and m.in_channels >= 32)))


def route_conv_winograd(net):
    """Re-class every eligible 3x3 / stride-1 convolution of `net` in place; returns the number of layers routed."""
    n = 0
    if not _WINOGRAD:
        return n
    for m in net.modules():
        if winograd_eligible(m):
            m.__class__ = WinogradConv2d
            n += 1
    return n


# ================================================================================= MFMA forward (csrc/conv_fwd.hip)
# The layers MIOpen serves through NHWC implicit-GEMM kernels wrapped in layout transposes and zero-fills -- stride-2
# convolutions, 4x4 / stride-2 transposed convolutions, small-plane 3x3 layers with awkward channel counts (FlowNet's
# 1026 / 770 / 386-channel concatenations: 125-160 us through the vendor path, 26-30 us here) -- run their FORWARD on the
# hand-written fp32 MFMA kernel, in training too, and so does their whole backward: the data gradient of a 4x4 / stride-2
# convolution is the transposed convolution with the same weight tensor and vice versa, the data gradient of the 3x3
# layers is the same kernel with the weight read transposed / as parity classes (csrc/conv_fwd.hip modes 1-3), the
# weight gradients are csrc/conv_bwd.hip.  No NHWC implicit GEMM, no layout transposes, no zero-fills for these layers.
def _conv_fwd_call(x, weight, bias, stride, pad, transposed):
    from .flownet_eval import conv_mfma, NONE
    return conv_mfma(x, weight, bias, stride, pad, transposed, NONE)


def _bias_grad(go):
    return go.sum((0, 2, 3))


# The data / weight gradients of the layers routed to csrc/conv_fwd.hip: the kernels exist and are parity-tested (conv_fwd.hip
# modes 1-3, conv_bwd.hip), but inside the train step they lose ~1 ms each against the vendor's (A/B in DESIGN.md section 6), so
# they are opt-in: FFWM_CONV_DGRAD=1 / FFWM_CONV_WGRAD=1.
_OWN_DGRAD = _os.environ.get("FFWM_CONV_DGRAD", "0") == "1"            # stride-2 / every 3x3 stride-1 data gradient on conv_fwd.hip: opt-in (slower)
_OWN_DGRAD_T = _os.environ.get("FFWM_CONVT_DGRAD", "1") != "0"        # transposed convolutions' data gradient: measured faster, on


class _MfmaConv2d(Function):
    """Conv2d (3x3 / 4x4, stride 1 / 2): forward on csrc/conv_fwd.hip (own_fwd) or the vendor library; backward per direction by
    measurement (tools/bwd_layers.py): weight gradient through conv_weight_grad, data gradient on csrc/conv_fwd.hip for the
    3x3 / stride-1 layers on planes of <= 64 pixels (FlowNet's conv5_1 / conv6_1 / inter_conv5 / inter_conv4: 24-48 us against the
    vendor's 32-53), with the vendor for the stride-2 layers (its NHWC kernel + transposes is still 5-20 us faster than mode 1 / 2)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, own_fwd=True):
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, pad, bias is not None)
        if own_fwd:
            return _conv_fwd_call(x, weight, bias, stride, pad, False)
        return torch.ops.aten.convolution(x, weight, bias, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        stride, pad, has_bias = ctx.cfg
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = None
        k = weight.size(2)
        even = x.size(2) == 2 * go.size(2) and x.size(3) == 2 * go.size(3)
        if need_x and _OWN_DGRAD and stride == 2 and pad == 1 and even and min(weight.shape[0], weight.shape[1]) >= 32:
            # d(input) of Conv2d(C, K, 4, 2, 1) = ConvTranspose2d(K, C, 4, 2, 1) with the SAME weight tensor [K, C, 4, 4]; of
            # Conv2d(C, K, 3, 2, 1) = the transposed 3x3 (output padding 1): parity classes with 1 or 2 taps per axis
            gx = _conv_fwd_call(go, weight, None, 2, 1, 1 if k == 4 else 2)
            need_x = False
        elif (need_x and k == 3 and stride == 1 and pad == 1 and min(weight.shape[0], weight.shape[1]) >= 32
              and (_OWN_DGRAD or x.size(2) * x.size(3) <= 64)):
            gx = _conv_fwd_call(go, weight, None, 1, 1, 3)         # 3x3 / stride-1 convolution of grad_output, weight read rotated
            need_x = False
        if need_w:
            gw, gb = conv_weight_grad(x, go, weight, stride, pad, need_b)
        if need_x:
            gx = torch.ops.aten.convolution_backward(go, x, weight, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if need_b and gb is None:
            gb = _bias_grad(go)
        return gx, gw, gb, None, None, None


class _MfmaConvTranspose2d(Function):
    """ConvTranspose2d(Ci, Co, 4, 2, 1): forward on csrc/conv_fwd.hip (own_fwd) or the vendor library; data gradient = Conv2d(Co, Ci,
    4, 2, 1) with the same weight tensor on csrc/conv_fwd.hip (20-38 us against the vendor's 25-58 on every deconv of FlowNet);
    weight gradient through conv_transpose_weight_grad."""

    @staticmethod
    def forward(ctx, x, weight, bias, own_fwd=True):
        x, weight = x.contiguous(), weight.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if own_fwd:
            return _conv_fwd_call(x, weight, bias, 2, 1, True)
        return torch.ops.aten.convolution(x, weight, bias, [2, 2], [1, 1], [1, 1], True, [0, 0], 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        x, weight = ctx.saved_tensors
        go = grad_output.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        gx = gw = gb = None
        if need_x and _OWN_DGRAD_T and min(weight.shape[0], weight.shape[1]) >= 16:
            # d(input) of ConvTranspose2d(C, K, 4, 2, 1) = Conv2d(K, C, 4, 2, 1) with the same weight tensor [C, K, 4, 4]
            gx = _conv_fwd_call(go, weight, None, 2, 1, False)
            need_x = False
        if need_w:
            gw, gb = conv_transpose_weight_grad(x, go, weight, need_b)
        if need_x:
            gx = torch.ops.aten.convolution_backward(go, x, weight, None, [2, 2], [1, 1], [1, 1], True, [0, 0], 1, [True, False, False])[0]
        if need_b and gb is None:
            gb = _bias_grad(go)
        return gx, gw, gb, None


def fwd_route_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled()
            and x.numel() < (1 << 29) and weight.numel() < (1 << 29))


class MfmaFwdConv2d(nn.Conv2d):
    """nn.Conv2d (3x3 or 4x4, stride 1 or 2) whose forward runs on csrc/conv_fwd.hip."""

    def _conv_forward(self, input, weight, bias):
        if fwd_route_ok(input, weight) and (self.stride[0] == 2 or input.size(2) <= 32):
            return _MfmaConv2d.apply(input, weight, bias, self.stride[0], self.padding[0])
        return super()._conv_forward(input, weight, bias)


class MfmaFwdConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(C, K, 4, 2, 1) whose forward and data gradient run on csrc/conv_fwd.hip."""

    def forward(self, input, output_size=None):
        if output_size is None and fwd_route_ok(input, self.weight):
            return _MfmaConvTranspose2d.apply(input, self.weight, self.bias)
        return super().forward(input, output_size)


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
    if type(m) in (nn.Conv2d, WinogradConv2d):
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
            if type(m) is WinogradConv2d:
                m._mfma_fwd_small = True          # its small-plane calls (winograd_ok false)
            else:
                m.__class__ = MfmaFwdConv2d if isinstance(m, nn.Conv2d) else MfmaFwdConvTranspose2d
            n += 1
    return n


# ================================================================================= the rest: vendor forward, own backward
# What none of the routes above takes -- FlowNet's thin decoder layers (18 / 34 channels), its seven two-channel flow heads and six
# 2 -> 2 flow upsamplers, netD's image layers -- keeps the vendor's forward (launch-bound either way) but gets its weight gradient
# from the tiled kernel: the vendor's path is an NHWC implicit GEMM behind two or three layout transposes and a fill per call.
class OwnBwdConv2d(nn.Conv2d):
    def _conv_forward(self, input, weight, bias):
        if (torch.is_grad_enabled() and weight.requires_grad and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
                and not torch.is_autocast_enabled()):
            return _MfmaConv2d.apply(input, weight, bias, self.stride[0], self.padding[0], False)
        return super()._conv_forward(input, weight, bias)


class OwnBwdConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, input, output_size=None):
        if (output_size is None and torch.is_grad_enabled() and self.weight.requires_grad and input.is_cuda
                and input.dtype == torch.float32 and input.dim() == 4 and not torch.is_autocast_enabled()):
            return _MfmaConvTranspose2d.apply(input, self.weight, self.bias, False)
        return super().forward(input, output_size)


def route_conv_bwd(net):
    """Re-class every convolution no other route has taken whose weight gradient the tiled kernel serves; returns the number."""
    n = 0
    if not _TILED_WGRAD:
        return n
    for m in net.modules():
        if type(m) is nn.Conv2d and _tiled_wgrad_layer_ok(m):
            m.__class__ = OwnBwdConv2d
            n += 1
        elif (type(m) is nn.ConvTranspose2d and m.kernel_size == (4, 4) and m.stride == (2, 2) and m.padding == (1, 1)
              and m.output_padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1):
            m.__class__ = OwnBwdConvTranspose2d
            n += 1
    return n


# ================================================================================= FlowNet's two-channel layers in training
# predict_flow* = Conv2d(C, 2, 3, 1, 1) + Tanh and upsampled_flow* = ConvTranspose2d(2, 2, 4, 2, 1) (models/base_networks.py:45-49,
# 104-109): forward on the launch-lean kernels of the eval path (csrc/flownet_ops.hip: conv + bias + tanh in one launch), data
# gradient on their new backward twins, weight (and bias) gradient on the tiled kernel.  Through the vendor library each of the
# 26 layers of the two flow nets costs ~10 launches per step, most of them layout transposes around an implicit GEMM that uses
# two of its 64 rows.
class _FlowHead(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        from .flownet_eval import flow_head
        x, weight = x.contiguous(), weight.contiguous()
        y = flow_head(x, weight, bias)
        ctx.save_for_backward(x, weight, y)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_y):
        from . import _lib
        x, weight, y = ctx.saved_tensors
        go = grad_y.contiguous()
        B, C, H, W = x.shape
        gz = torch.empty_like(y)
        gx = torch.empty_like(x)
        with ops._on_device(x) as stream:          # the tensors' device, not the thread's current one
            _lib.check(_lib.load().ffwm_flow_head_backward(y.data_ptr(), go.data_ptr(), weight.data_ptr(), gz.data_ptr(), gx.data_ptr(), B, C, H, W,
                                                           _lib.F32, stream), "ffwm_flow_head_backward")
        gw = gb = None
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb = conv_weight_grad(x, gz, weight, 1, 1, need_b)
        if need_b and gb is None:
            gb = _bias_grad(gz)
        return (gx if ctx.needs_input_grad[0] else None), gw, gb


class FlowHead(nn.Sequential):
    """nn.Sequential(Conv2d(C, 2, 3, 1, 1), Tanh) as ONE launch per direction on the GPU (same parameters, same state-dict keys)."""

    def forward(self, x):
        conv = self[0]
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.weight.dtype == torch.float32 and conv.bias is not None
                and not torch.is_autocast_enabled() and x.shape[2] * x.shape[3] < (1 << 28)):
            return _FlowHead.apply(x, conv.weight, conv.bias)
        return super().forward(x)


class _FlowUp(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        from .flownet_eval import flow_up
        x, weight = x.contiguous(), weight.contiguous()
        B, _, H, W = x.shape
        out = torch.empty(B, 2, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
        flow_up(x, weight, bias, out)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import _lib
        x, weight = ctx.saved_tensors
        B, _, H, W = x.shape
        go = grad_out
        # a channel slice of the decoder concatenation's gradient is read in place (batch stride), anything else is made contiguous
        if not (go.stride(3) == 1 and go.stride(2) == 2 * W and go.stride(1) == 4 * H * W and go.stride(0) >= 8 * H * W):
            go = go.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            with ops._on_device(x) as stream:
                _lib.check(_lib.load().ffwm_flow_up_backward(go.data_ptr(), weight.data_ptr(), gx.data_ptr(), B, H, W, go.stride(0), _lib.F32,
                                                             stream), "ffwm_flow_up_backward")
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb = conv_transpose_weight_grad(x, go.contiguous(), weight, need_b)
        elif need_b:
            gb = _bias_grad(go)
        return gx, gw, gb


class FlowUpConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d(2, 2, 4, 2, 1) on the direct kernels of csrc/flownet_ops.hip, forward and backward."""

    def forward(self, input, output_size=None):
        if (output_size is None and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and self.bias is not None
                and not torch.is_autocast_enabled() and input.shape[2] * input.shape[3] < (1 << 26)):
            return _FlowUp.apply(input, self.weight, self.bias)
        return super().forward(input, output_size)


def route_flow_heads(net):
    """Re-class the flow heads (Sequential(Conv2d(C, 2, 3, 1, 1), Tanh)) and the 2 -> 2 flow upsamplers of a FlowNet; -> count."""
    n = 0
    for m in net.modules():
        if (type(m) is nn.Sequential and len(m) == 2 and isinstance(m[0], nn.Conv2d) and isinstance(m[1], nn.Tanh)
                and m[0].out_channels == 2 and m[0].kernel_size == (3, 3) and m[0].stride == (1, 1) and m[0].padding == (1, 1)
                and m[0].dilation == (1, 1) and m[0].groups == 1 and m[0].bias is not None):
            m.__class__ = FlowHead
            n += 1
        elif (isinstance(m, nn.ConvTranspose2d) and m.in_channels == 2 and m.out_channels == 2 and m.kernel_size == (4, 4)
              and m.stride == (2, 2) and m.padding == (1, 1) and m.output_padding == (0, 0) and m.dilation == (1, 1) and m.groups == 1
              and m.bias is not None):
            m.__class__ = FlowUpConvTranspose2d
            n += 1
    return n


# ================================================================================= every route of a training network, in the trainer's order
def route_training_kernels(net, convs=True, wgrad=True):
    """Apply to `net` what FFWMTrainer applies to netG / netD (trainer.py:160-205), in the same order, and return the counts: the
    3x3 weight gradients (conv_wgrad.hip / conv_wgrad_wino.hip), Winograd forward + data gradient (conv_winograd.hip), the direct
    MFMA kernel for stride-2 / transposed / small-plane layers (conv_fwd.hip), the tiled weight gradient for what is left
    (conv_bwd.hip), BatchNorm + LeakyReLU pairs, residual tails / the warp-attention gate, batched spectral norm.  For a module that
    is trained outside the trainer (bench.py's warp + attention sub-path); convs=False keeps the vendor's convolutions (A/B)."""
    from .norm import fuse_bn_lrelu
    from .residual import fuse_residual
    from .spectral_norm import fuse_spectral_norm
    n = {}
    if convs:
        if wgrad:
            n["mfma_wgrad"] = route_conv_wgrad(net)
        n["winograd"] = route_conv_winograd(net)
        n["mfma_fwd"] = route_conv_fwd(net)
        n["own_bwd"] = route_conv_bwd(net)
    n["bn_lrelu"] = fuse_bn_lrelu(net)
    n["residual"] = fuse_residual(net)
    fuse_spectral_norm(net)
    return n
