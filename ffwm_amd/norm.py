"""Training-mode BatchNorm2d fused with the LeakyReLU that follows it (csrc/bn_lrelu.hip).

The conv blocks of the reference (models/base_networks.py:12-31 FlowNet, :207-264 FFWM, :381-413 discriminators) are
``conv -> nn.BatchNorm2d -> nn.LeakyReLU(0.2)``.  ``fuse_bn_lrelu(net)`` re-classes each such BatchNorm2d in place
(parameters, buffers and state-dict keys untouched) and replaces the LeakyReLU that follows it in the same
``nn.Sequential`` by ``nn.Identity`` (no parameters: the container's indices, hence every key, stay the same).
In eval mode, on the CPU, or for inputs the kernel does not take, the module is exactly BatchNorm2d + leaky_relu.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _ext, _lib


def _launch(fn_name, x, *args):
    dev = x.device.index
    cur = torch.cuda.current_device()
    if cur != dev:
        torch.cuda.set_device(dev)
    try:
        fn = getattr(_lib.load(), fn_name)
        _lib.check(fn(*args, _lib.F32, torch.cuda.current_stream(dev).cuda_stream), fn_name)
    finally:
        if cur != dev:
            torch.cuda.set_device(cur)


def _p(t):
    return None if t is None else t.data_ptr()


_SCRATCH = {}


def _scratch(x):
    """Zero-filled scratch (2 C sums + C arrival counters) when a channel is worth splitting over several workgroups (few channels,
    large planes).  ONE buffer per (device, C), filled once: the kernels hand it back zero-filled, and the calls that share it
    are ordered on their stream.  (Other streams get their own buffer.)"""
    C = x.shape[1]
    if C < 512 and x.numel() // C >= 32768:
        key = (x.device, C, torch.cuda.current_stream(x.device).cuda_stream)
        buf = _SCRATCH.get(key)
        if buf is None:
            buf = _SCRATCH[key] = torch.zeros(2 * C + (C + 1) // 2, device=x.device, dtype=torch.float64)
        return buf
    return None


def reset_scratch():
    """Drop every cached scratch buffer (here and in the C++ binding).  trainer.capture() calls it before a capture starts: a buffer
    first made inside a capture lives in that graph's private pool and its zero-fill is a node of that graph alone -- a later capture
    (capture -> release_graphs -> capture) that found it in the cache would replay on memory nobody filled."""
    _SCRATCH.clear()
    ext = _ext.get()
    if ext is not None and hasattr(ext, "bn_scratch_reset"):
        ext.bn_scratch_reset()


class _BnLreluFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, slope):
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(C, device=x.device, dtype=torch.float32)
        save_invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        scratch = _scratch(x)          # (freed after the launch: the caching allocator reuses it in stream order only)
        _launch("ffwm_bn_lrelu_forward", x, _p(x), _p(weight), _p(bias), _p(running_mean), _p(running_var), _p(y),
                _p(save_mean), _p(save_invstd), _p(scratch), B, C, H * W, float(eps), float(momentum), float(slope))
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.slope = float(slope)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        B, C, H, W = x.shape
        go = grad_y.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty(C, device=x.device, dtype=torch.float32) if (need_w and weight is not None) else None
        db = torch.empty(C, device=x.device, dtype=torch.float32) if (need_b and bias is not None) else None
        scratch = _scratch(x)
        _launch("ffwm_bn_lrelu_backward", x, _p(x), _p(go), _p(weight), _p(bias), _p(save_mean), _p(save_invstd), _p(dx),
                _p(dw), _p(db), _p(scratch), B, C, H * W, ctx.slope)
        return dx, dw, db, None, None, None, None, None


def _launch_act(fn_name, x, act, *args):
    dev = x.device.index
    cur = torch.cuda.current_device()
    if cur != dev:
        torch.cuda.set_device(dev)
    try:
        fn = getattr(_lib.load(), fn_name)
        _lib.check(fn(*args, int(act), _lib.F32, torch.cuda.current_stream(dev).cuda_stream), fn_name)
    finally:
        if cur != dev:
            torch.cuda.set_device(cur)


class _BnResActFunction(Function):
    """y = act(BatchNorm_train(x) + res + rbias[c]): the tail of a residual block (csrc/bn_lrelu.hip, RES variant)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, res, rbias, eps, momentum, slope, act):
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(C, device=x.device, dtype=torch.float32)
        save_invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        scratch = _scratch(x)
        _launch_act("ffwm_bn_res_act_forward", x, act, _p(x), _p(weight), _p(bias), _p(running_mean), _p(running_var), _p(res), _p(rbias),
                    _p(y), _p(save_mean), _p(save_invstd), _p(scratch), B, C, H * W, float(eps), float(momentum), float(slope))
        ctx.save_for_backward(x, weight, save_mean, save_invstd, y)
        ctx.slope, ctx.act = float(slope), int(act)
        ctx.has_bias, ctx.has_rbias = bias is not None, rbias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_y):
        x, weight, save_mean, save_invstd, y = ctx.saved_tensors
        B, C, H, W = x.shape
        go = grad_y.contiguous()
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        dres = torch.empty_like(x) if need[5] else None
        dw = torch.empty(C, device=x.device, dtype=torch.float32) if (need[1] and weight is not None) else None
        want_db = (need[2] and ctx.has_bias) or (need[6] and ctx.has_rbias)
        db = torch.empty(C, device=x.device, dtype=torch.float32) if want_db else None
        scratch = _scratch(x)
        _launch_act("ffwm_bn_res_act_backward", x, ctx.act, _p(x), _p(y), _p(go), _p(weight), _p(save_mean), _p(save_invstd), _p(dx), _p(dres),
                    _p(dw), _p(db), _p(scratch), B, C, H * W, ctx.slope)
        return (dx, dw, db if (need[2] and ctx.has_bias) else None, None, None, dres, db if (need[6] and ctx.has_rbias) else None,
                None, None, None, None)


RES_ACTS = {"lrelu": 0, "sigmoid": 1}
_BN_RES = os.environ.get("FFWM_BN_RES_ACT", "1") != "0"


def bn_res_act_ok(bn, x, act_code):
    """The fused tail serves: a training-mode BatchNorm2d with a fixed momentum on a contiguous float32 GPU tensor, LeakyReLU / sigmoid."""
    # (exactly a plain BatchNorm2d or its host-counted twin: a BatchNormLeakyReLU2d IS-A BatchNorm2d too, and the fused tail would
    # silently drop its activation)
    return (_BN_RES and act_code is not None and type(bn) in (nn.BatchNorm2d, HostCountBatchNorm2d) and bn.training and bn.momentum is not None and torch.is_tensor(x)
            and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() // x.shape[1] > 1 and not torch.is_autocast_enabled()
            and (bn.weight is None or bn.weight.dtype == torch.float32))


def bn_res_act(x, bn, res, rbias, act_code, slope):
    """act(bn(x) + res + rbias) with bn in training mode: one kernel per direction; the batch counter is kept on the host like the
    other fused BatchNorm modules do (norm._bn_host_counted)."""
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if hasattr(bn, "flush_batch_counter"):
            bn._pending_batches += 1
        else:
            bn.num_batches_tracked.add_(1)
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    return _BnResActFunction.apply(x.contiguous(), bn.weight, bn.bias, rm, rv, res.contiguous(), rbias, bn.eps, bn.momentum, slope, act_code)


# Below this many elements a BatchNorm + LeakyReLU pair is launch-bound either way and the two native ops cost
# less host time than one Python autograd Function; above it the saved read-modify-write passes win.
MIN_FUSED_NUMEL = 1 << 20


# ... with the C++ binding (ffwm_amd/_ext.py) the host cost no longer decides; measured on the train step, floors of
# 0 / 256 K / 1 M elements give 60.1 / 59.1 / 59.0 ms: tiny planes are better off in the vendor's two kernels
MIN_FUSED_NUMEL_EXT = int(os.environ.get("FFWM_BN_MIN_NUMEL", str(1 << 20)))


def _kernel_ok(x):
    floor = MIN_FUSED_NUMEL if (MIN_FUSED_NUMEL == 0 or _ext.get() is None) else MIN_FUSED_NUMEL_EXT
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.numel() >= floor and x.is_contiguous()
            and x.data_ptr() % 16 == 0)


class BatchNormLeakyReLU2d(nn.BatchNorm2d):
    """nn.BatchNorm2d followed by leaky_relu(negative_slope); one kernel per direction in training mode on the GPU."""

    negative_slope = 0.2
    _pending_batches = 0

    def forward(self, x):
        if (self.training and self.momentum is not None and _kernel_ok(x)
                and (self.weight is None or self.weight.dtype == torch.float32)):
            if self.track_running_stats and self.num_batches_tracked is not None:
                # the counter only matters for momentum=None and for checkpoints: count on the host, add when a
                # state dict is taken (one tiny kernel per BatchNorm and step otherwise)
                self._pending_batches += 1
            rm = self.running_mean if self.track_running_stats else None
            rv = self.running_var if self.track_running_stats else None
            ext = _ext.get()
            if ext is not None:                # C++ autograd binding: same kernels, a fraction of the host cost
                return ext.bn_lrelu(x, self.weight, self.bias, rm, rv, self.eps, self.momentum, self.negative_slope)
            return _BnLreluFunction.apply(x, self.weight, self.bias, rm, rv, self.eps, self.momentum, self.negative_slope)
        return F.leaky_relu(_bn_host_counted(self, x), self.negative_slope)

    def flush_batch_counter(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
        self._pending_batches = 0


def _bn_host_counted(m, x):
    """nn.BatchNorm2d.forward with the batch counter on the host: in training mode with a fixed momentum `num_batches_tracked` takes no
    part in the arithmetic, yet nn.BatchNorm2d spends one tiny kernel per call on `+= 1` (68 launches per FFWM train step for the
    BatchNorms the fused kernel does not take).  Counted in `_pending_batches`, added when a state dict is taken."""
    if m.training and m.momentum is not None and m.track_running_stats and m.num_batches_tracked is not None and x.is_cuda:
        m._pending_batches += 1
        return F.batch_norm(x, m.running_mean, m.running_var, m.weight, m.bias, True, m.momentum, m.eps)
    return nn.BatchNorm2d.forward(m, x)


class HostCountBatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d whose batch counter lives on the host between state-dict snapshots (same results, one launch less per call)."""

    _pending_batches = 0

    def forward(self, x):
        return _bn_host_counted(self, x)

    def flush_batch_counter(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
        self._pending_batches = 0


def _flush_hook(module, *args, **kwargs):
    module.flush_batch_counter()


def _reset_hook(module, *args, **kwargs):
    module._pending_batches = 0            # a loaded num_batches_tracked replaces whatever was counted before the load


def fuse_bn_lrelu(net):
    """Re-class every BatchNorm2d that is directly followed by a LeakyReLU in an nn.Sequential; returns the count."""
    n = 0
    for seq in [m for m in net.modules() if isinstance(m, nn.Sequential)]:
        keys = list(seq._modules.keys())
        for ka, kb in zip(keys, keys[1:]):
            a, b = seq._modules[ka], seq._modules[kb]
            if type(a) is nn.BatchNorm2d and type(b) is nn.LeakyReLU:
                a.__class__ = BatchNormLeakyReLU2d
                a.negative_slope = b.negative_slope
                a._pending_batches = 0
                a.register_state_dict_pre_hook(_flush_hook)
                a.register_load_state_dict_pre_hook(_reset_hook)
                seq._modules[kb] = nn.Identity()
                n += 1
    for m in net.modules():                 # the BatchNorms without a LeakyReLU behind them: host-side batch counter only
        if type(m) is nn.BatchNorm2d:
            m.__class__ = HostCountBatchNorm2d
            m._pending_batches = 0
            m.register_state_dict_pre_hook(_flush_hook)
            m.register_load_state_dict_pre_hook(_reset_hook)
    return n
