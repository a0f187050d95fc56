"""Batched spectral normalisation: every spectrally-normalised conv of a network in ONE launch.

The reference wraps each convolution of FFWM (netG) and MSDiscriminator (netD) in
``torch.nn.utils.spectral_norm`` (/root/reference/models/base_networks.py:5,218-264,381-413), whose
forward-pre-hook runs ~12 tiny kernels per layer per forward call and ~6 per backward: ~1700 of the
train step's ~4600 launches.  ``fuse_spectral_norm(net)`` keeps the modules, parameters and buffers
exactly as they are (``weight_orig`` / ``weight_u`` / ``weight_v``: state dicts stay interchangeable
with the reference's), removes the per-layer hooks and installs ONE pre-forward hook on the network
that calls ``ffwm_spectral_norm_forward`` (csrc/spectral_norm.hip) for all layers at once; the
backward of all layers is one ``ffwm_spectral_norm_backward`` launch.

Semantics kept: one power iteration per forward call in training mode, none in eval mode, u / v
updated in place, gradients treat u and v as constants (the hook runs the iteration under no_grad).
GPU only, like every op of this package.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.nn.utils.spectral_norm import SpectralNorm

from . import _lib


class _SnLayer(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("wv", ctypes.c_void_p),
                ("weight_sn", ctypes.c_void_p), ("sigma", ctypes.c_void_p), ("u_saved", ctypes.c_void_p),
                ("v_saved", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


class _SnGradLayer(ctypes.Structure):
    _fields_ = [("weight", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p), ("sigma", ctypes.c_void_p),
                ("grad_weight_sn", ctypes.c_void_p), ("grad_weight", ctypes.c_void_p), ("partials", ctypes.c_void_p),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


def _bind():
    return _lib.load()


_DT = {torch.float32: _lib.F32, torch.float64: _lib.F64}


class _SnGroupFunction(Function):
    @staticmethod
    def forward(ctx, group, training, *weights):
        lib = _bind()
        n = len(weights)
        w0 = weights[0]
        dev, dt = w0.device, w0.dtype
        outs = [torch.empty_like(w) for w in weights]
        saved = torch.empty(group.uv_total + n, device=dev, dtype=dt)      # [u_0 v_0 u_1 v_1 ... | sigma_0..n-1]
        esz = w0.element_size()
        base = saved.data_ptr()
        arr = group.fwd_array
        for i, (w, o) in enumerate(zip(weights, outs)):
            a = arr[i]
            a.weight = w.data_ptr()
            a.weight_sn = o.data_ptr()
            a.u_saved = base + group.u_off[i] * esz
            a.v_saved = base + group.v_off[i] * esz
            a.sigma = base + (group.uv_total + i) * esz
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.ffwm_spectral_norm_forward(ctypes.cast(arr, ctypes.c_void_p), n, group.n_power_iterations if training else 0, group.eps,
                                                  _DT[dt], stream), "ffwm_spectral_norm_forward")
        ctx.save_for_backward(*weights)
        ctx.group, ctx.saved = group, saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _bind()
        group, saved = ctx.group, ctx.saved
        weights = ctx.saved_tensors
        esz = saved.element_size()
        base = saved.data_ptr()
        arr = group.bwd_array
        partials = torch.empty(group.partials_total, device=saved.device, dtype=saved.dtype)
        pbase = partials.data_ptr()
        gws, keep, k = [], [], 0
        for i, (w, g) in enumerate(zip(weights, grads)):
            if g is None or not ctx.needs_input_grad[2 + i]:
                gws.append(None)
                continue
            g = g.contiguous()
            keep.append(g)
            gw = torch.empty_like(w)
            a = arr[k]
            a.weight = w.data_ptr()
            a.u = base + group.u_off[i] * esz
            a.v = base + group.v_off[i] * esz
            a.sigma = base + (group.uv_total + i) * esz
            a.grad_weight_sn = g.data_ptr()
            a.grad_weight = gw.data_ptr()
            a.partials = pbase + group.p_off[i] * esz
            a.rows, a.cols = group.rows[i], group.cols[i]
            gws.append(gw)
            k += 1
        if k:
            stream = torch.cuda.current_stream(saved.device).cuda_stream
            _lib.check(lib.ffwm_spectral_norm_backward(ctypes.cast(arr, ctypes.c_void_p), k, _DT[saved.dtype], stream), "ffwm_spectral_norm_backward")
        return (None, None) + tuple(gws)


class SpectralNormGroup(object):
    """All spectrally-normalised layers of one network, normalised together."""

    def __init__(self, net):
        self.layers = []            # (module, name)
        for m in net.modules():
            for key, hook in list(m._forward_pre_hooks.items()):
                if isinstance(hook, SpectralNorm):
                    if hook.dim != 0:
                        raise NotImplementedError("fused spectral norm: only dim=0 layers (Conv2d / Linear)")
                    self.layers.append((m, hook.name))
                    self.n_power_iterations, self.eps = hook.n_power_iterations, hook.eps
                    del m._forward_pre_hooks[key]
        if not self.layers:
            raise ValueError("no spectral_norm layer found")
        self.rows, self.cols, self.u_off, self.v_off = [], [], [], []
        off = 0
        for m, name in self.layers:
            w = getattr(m, name + "_orig")
            r = w.size(0)
            c = w.numel() // r
            self.rows.append(r)
            self.cols.append(c)
            self.u_off.append(off)
            self.v_off.append(off + r)
            off += r + c
        self.uv_total = off
        self.p_off, poff = [], 0
        for r, c in zip(self.rows, self.cols):
            self.p_off.append(poff)
            poff += (r * c + 8191) // 8192
        self.partials_total = poff
        n = len(self.layers)
        self.fwd_array = (_SnLayer * n)()
        self.bwd_array = (_SnGradLayer * n)()
        w0 = getattr(self.layers[0][0], self.layers[0][1] + "_orig")
        self.wv = torch.empty(sum(self.rows), device=w0.device, dtype=w0.dtype)
        self._bind_static()

    def _bind_static(self):
        """Pointers that do not change between calls (u / v buffers, scratch, sizes)."""
        esz = self.wv.element_size()
        roff = 0
        for i, (m, name) in enumerate(self.layers):
            a = self.fwd_array[i]
            a.u = getattr(m, name + "_u").data_ptr()
            a.v = getattr(m, name + "_v").data_ptr()
            a.wv = self.wv.data_ptr() + roff * esz
            a.rows, a.cols = self.rows[i], self.cols[i]
            roff += self.rows[i]

    def __call__(self, training):
        weights = [getattr(m, name + "_orig") for m, name in self.layers]
        w0 = weights[0]
        if not w0.is_cuda:
            raise NotImplementedError("fused spectral norm runs on the GPU only")
        if w0.device != self.wv.device or w0.dtype != self.wv.dtype:       # the network was moved / cast
            self.wv = torch.empty(sum(self.rows), device=w0.device, dtype=w0.dtype)
        self._bind_static()
        outs = _SnGroupFunction.apply(self, training, *weights)
        for (m, name), w in zip(self.layers, outs):
            setattr(m, name, w)            # what the per-layer hook does: a plain attribute, not a Parameter
        self._prepare_winograd(outs)

    def _prepare_winograd(self, outs):
        """Every weight of the network is known here: the Winograd transforms its 3x3 layers used in the last pass (forward, and the data
        gradient when gradients are on) go out as ONE launch per 24 instead of one small launch in front of every convolution."""
        from . import conv, ops
        if not conv._WINO_BATCH or outs[0].dtype != torch.float32:
            return
        items, slots = [], []
        grad = torch.is_grad_enabled()
        for (m, name), w in zip(self.layers, outs):
            if name != "weight" or not isinstance(m, conv.WinogradConv2d) or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3):
                continue
            w._ffwm_wino_owner = m
            w._ffwm_wino = {}
            for key in sorted(m.__dict__.get("_wino_keys", ())):
                if key[0] == 1 and not grad:
                    continue
                items.append((w, key[0], key[1]))
                slots.append((w, key))
        if items:
            for (w, key), ws in zip(slots, ops.conv3x3_winograd_weights_multi(items)):
                w._ffwm_wino[key] = ws


def fuse_spectral_norm(net):
    """Replace the per-layer spectral-norm hooks of ``net`` by one batched launch per forward call."""
    group = SpectralNormGroup(net)
    net._ffwm_sn_group = group
    net.register_forward_pre_hook(lambda mod, inputs: group(mod.training))
    return group
