"""The element-wise tails of netG's residual blocks and its warp-attention gate on csrc/residual.hip.

Reference: models/base_networks.py:207-233 (`ResidualBlock.forward`: activ(blocks(x) + input(x))) and :326-333
(`FFWM.forward`: skip = skip * att_i(skip), att_i ending in a sigmoid ResidualBlock).  PyTorch runs an add and an activation
per block and a multiply behind the gate; here each is one pass forward and one backward, same values (ATen's order of
operations).  `fuse_residual(net)` re-classes the blocks in place (state-dict names untouched)."""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


class _AddAct(Function):
    @staticmethod
    def forward(ctx, a, b, act, slope):
        y = ops.add_act_forward(a.contiguous(), b.contiguous(), act, slope)
        ctx.save_for_backward(y)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        dz = ops.add_act_backward(y, g.contiguous(), ctx.act, ctx.slope)
        return dz, dz, None, None


class _SigmoidGate(Function):
    @staticmethod
    def forward(ctx, a, b, x):
        x = x.contiguous()
        y, att = ops.sigmoid_gate_forward(a.contiguous(), b.contiguous(), x)
        ctx.save_for_backward(x, att)
        ctx.mark_non_differentiable(att)
        return y, att

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, _gatt):
        x, att = ctx.saved_tensors
        dz, dx = ops.sigmoid_gate_backward(x, att, gy.contiguous())
        return dz, dz, dx


def _act_of(module):
    if isinstance(module, nn.LeakyReLU) and module.negative_slope > 0:
        return "lrelu", float(module.negative_slope)
    if isinstance(module, nn.Sigmoid):
        return "sigmoid", 0.0
    return None, 0.0


def _kernel_ok(*ts):
    t0 = ts[0]
    return (t0.is_cuda and not torch.is_autocast_enabled()
            and all(t.dtype == torch.float32 and t.shape == t0.shape and t.device == t0.device for t in ts))


def add_act(a, b, activ):
    """activ(a + b) for an nn.LeakyReLU / nn.Sigmoid module `activ`; the PyTorch composition when the kernel does not apply."""
    act, slope = _act_of(activ)
    if act is not None and _kernel_ok(a, b):
        return _AddAct.apply(a, b, act, slope)
    return activ(a + b)


def gated(att_module, skip, fuse=True):
    """-> (skip * att, att) with att = att_module(skip); when `att_module` is netG's att_i (conv block + sigmoid ResidualBlock) the
    residual add, the sigmoid and the product are one kernel."""
    rb = att_module[-1] if isinstance(att_module, nn.Sequential) and len(att_module) >= 2 else None
    if (fuse and rb is not None and hasattr(rb, "blocks") and hasattr(rb, "input") and isinstance(getattr(rb, "activ", None), nn.Sigmoid)
            and _kernel_ok(skip)):
        h = skip
        for m in list(att_module)[:-1]:
            h = m(h)
        a, b = rb.blocks(h), rb.input(h)
        if _kernel_ok(a, b, skip):
            return _SigmoidGate.apply(a, b, skip)
        att = rb.activ(a + b)
        return skip * att, att
    att = att_module(skip)
    return skip * att, att


def fuse_residual(net):
    """Re-class every ResidualBlock of `net` whose activation the kernel serves; returns the number."""
    from .nets import FusedResidualBlock, ResidualBlock
    n = 0
    for m in net.modules():
        if type(m) is ResidualBlock and _act_of(m.activ)[0] is not None:
            m.__class__ = FusedResidualBlock
            n += 1
    if hasattr(net, "fuse_gate"):
        net.fuse_gate = True
    return n
