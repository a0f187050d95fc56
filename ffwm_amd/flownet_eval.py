"""FlowNet eval forward as a launch-lean path (BASELINE configs[1]: FlowNetF forward, batch 6).

`nets.FlowNet.forward` (= /root/reference/models/base_networks.py:116-165) through the stock library path is ~215 kernel
launches at batch 6, 80 % of them under 10 us: the step is bound by launches and by tiny helper kernels, not by the
2.18 GMAC per image.  `FoldedFlowNet` computes the SAME function with

* eval-mode BatchNorm folded into the weights and bias of the convolution in front of it (exact up to rounding:
  w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma / sqrt(var + eps) + beta) -- :12-31 is conv -> BN -> LeakyReLU;
* one epilogue kernel per conv block (`ffwm_bias_act_forward`: bias + LeakyReLU) that also writes the result into its
  channel slice of the decoder's concatenation buffer, so `torch.cat` (:133-151) launches nothing;
* the seven `predict_flow*` heads (:45-49, 3x3 conv to TWO channels + tanh) and the six 2 -> 2 channel flow upsamplers
  (:104-109) as direct HIP kernels instead of implicit-GEMM launches wrapped in layout transposes;
* the whole forward replayed from ONE captured hipGraph (`graph=True`), removing the host launch cost of the remaining
  ~110 kernels.

Since round 6 no vendor convolution is left in this path: stride-2 / transposed / small-plane layers run on csrc/conv_fwd.hip (fp32 MFMA
implicit GEMM, reduction splits through workspace slots), the 64 x 64+ stride-1 layers on the library's Winograd kernel, conv0 / inter_conv0
on a direct thin-channel kernel, and the decoder's transposed convolutions on a side stream beside the flow chain.  Weights are
snapshotted at construction: build it from a network in `.eval()` and rebuild after the weights change.
"""
import ctypes
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops

LRELU, TANH, NONE = 1, 2, 0


def _fold(conv, bn):
    """(weight, bias) of conv followed by eval-mode bn.  Conv2d weight [Co,Ci,k,k]; ConvTranspose2d weight [Ci,Co,k,k]."""
    w = conv.weight.detach()
    b = conv.bias.detach() if conv.bias is not None else torch.zeros(bn.num_features, device=w.device, dtype=w.dtype)
    scale = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
    if isinstance(conv, nn.ConvTranspose2d):
        w = w * scale.view(1, -1, 1, 1)
    else:
        w = w * scale.view(-1, 1, 1, 1)
    b = (b - bn.running_mean.detach()) * scale + bn.bias.detach()
    return w.contiguous(), b.contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def bias_act(h, bias, act, y=None, y2=None, slope=0.2):
    """y / y2: None, or a 4-D view [B, C, H, W] whose samples are contiguous (a channel slice of a contiguous buffer)."""
    B, C, H, W = h.shape
    hw = H * W
    if y is None and y2 is None:
        y = h
    for d in (y, y2):
        if d is not None:
            assert d.shape == h.shape and d.stride(1) == hw and d.stride(3) == 1 and d.stride(2) == W
    _lib.check(_lib.load().ffwm_bias_act_forward(
        h.data_ptr(), None if bias is None else bias.data_ptr(), None if y is None else y.data_ptr(),
        None if y2 is None else y2.data_ptr(), B, C, hw, 0 if y is None else y.stride(0), 0 if y2 is None else y2.stride(0),
        act, float(slope), _lib.F32, _stream(h)), "ffwm_bias_act_forward")
    return y if y is not None else y2


def flow_head(x, weight, bias):
    B, C, H, W = x.shape
    y = torch.empty(B, 2, H, W, device=x.device, dtype=x.dtype)
    _lib.check(_lib.load().ffwm_flow_head_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), B, C, H, W,
                                                  _lib.F32, _stream(x)), "ffwm_flow_head_forward")
    return y


def thin_weights(weight):
    """A thin 3 x 3 layer's weights in the direct kernel's [C][3][3][K] arrangement (from Conv2d's [K, C, 3, 3])."""
    return weight.detach().permute(1, 2, 3, 0).contiguous()


def conv_thin(x, weight_ctk, bias, act, slope=0.2):
    """Conv2d(C, K, 3, 1, 1) + bias + LeakyReLU on the direct thin-channel kernel (csrc/flownet_ops.hip)."""
    B, C, H, W = x.shape
    K = weight_ctk.shape[-1]
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and weight_ctk.is_contiguous() and K % 8 == 0
            and tuple(weight_ctk.shape[:3]) == (C, 3, 3)):
        raise ValueError("conv_thin: float32 contiguous GPU tensors, weights [C, 3, 3, K] with K % 8 == 0")
    y = torch.empty(B, K, H, W, device=x.device, dtype=x.dtype)
    _lib.check(_lib.load().ffwm_conv_thin_forward(x.data_ptr(), weight_ctk.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                                  B, C, H, W, K, int(act), float(slope), _lib.F32, _stream(x)), "ffwm_conv_thin_forward")
    return y


def flow_up(flow, weight, bias, out):
    """out: [B, 2, 2H, 2W] view whose samples are contiguous (the last two channels of a concatenation buffer)."""
    B, _, H, W = flow.shape
    assert out.shape == (B, 2, 2 * H, 2 * W) and out.stride(1) == 4 * H * W
    _lib.check(_lib.load().ffwm_flow_up_forward(flow.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W,
                                                out.stride(0), _lib.F32, _stream(flow)), "ffwm_flow_up_forward")
    return out


class Arena(object):
    """The split-launch workspaces of one forward pass as slices of ONE tensor: a layer with so few output pixels that the convolution
    kernel cuts its reduction stores the slices' partial sums into a workspace (csrc/conv_fwd.hip: every slot is written before it is
    read, nothing is zero-filled).  The first pass measures, later passes carve; a slice is valid until the next begin().  (Rounds 2-5
    zero-filled accumulation buffers here for the atomics of the split: 17 fills, later one, per forward.)"""

    def __init__(self):
        self.buf, self.need, self.pos, self.used = None, 0, 0, 0

    def begin(self, device):
        self.need = max(self.need, self.used)
        if self.need and (self.buf is None or self.buf.numel() < self.need or self.buf.device != device):
            self.buf = torch.empty(self.need, device=device, dtype=torch.float32)
        self.pos = self.used = 0

    def take(self, n, device):
        padded = (n + 63) // 64 * 64            # 256-byte aligned slices
        self.used += padded
        if self.buf is not None and self.buf.device == device and self.pos + padded <= self.buf.numel():
            v = self.buf[self.pos:self.pos + n]
            self.pos += padded
            return v
        return torch.empty(n, device=device, dtype=torch.float32)


ZeroArena = Arena          # the name rounds 2-5 used


def conv_mfma(x, weight, bias, stride, pad, transposed, act, slope=0.2, dst=None, dst2=None, arena=None, split=True):
    """The hand-written fp32 MFMA convolution (csrc/conv_fwd.hip) with its bias + activation epilogue: fused in the kernel, or -- when
    the layer has so few output pixels that the launch is cut along the reduction -- applied by the fixed-order reduction of the
    slices' workspace slots.  dst: destination instead of a fresh tensor, dst2: a second destination for the same values; both
    4-D views whose samples are contiguous (channel slices of concatenation buffers)."""
    B, C, H, W = x.shape
    k = weight.size(2)
    mode = int(transposed)      # 0 conv, 1 ConvTranspose2d(4, 2, 1), 2 / 3: d(input) of Conv2d(3, 2, 1) / Conv2d(3, 1, 1) (x = grad_output)
    if mode in (1, 2):
        K, Ho, Wo = weight.size(1), 2 * H, 2 * W
    elif mode == 3:
        K, Ho, Wo = weight.size(1), H, W
    else:
        K = weight.size(0)
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    lib = _lib.load()
    y = dst if dst is not None else torch.empty(B, K, Ho, Wo, device=x.device, dtype=x.dtype)
    for d in (y, dst2):
        if d is not None:
            assert d.shape == (B, K, Ho, Wo) and d.stride(1) == Ho * Wo and d.stride(3) == 1 and d.stride(2) == Wo
    ws = None
    if split:
        need = lib.ffwm_conv2d_forward_workspace(B, C, H, W, K, k, stride, pad, mode)
        if need > 0:
            ws = arena.take(need // 4, x.device) if arena is not None else torch.empty(need // 4, device=x.device, dtype=torch.float32)
    _lib.check(lib.ffwm_conv2d_forward(x.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(), y.data_ptr(),
                                       None if dst2 is None else dst2.data_ptr(), B, C, H, W, K, k, stride, pad, mode, y.stride(0),
                                       0 if dst2 is None else dst2.stride(0), act, float(slope),
                                       None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel() * 4,
                                       _lib.F32, _stream(x)), "ffwm_conv2d_forward")
    return y


class FoldedFlowNet(object):
    def __init__(self, net, graph=False, mfma_conv=True, mfma_min_channels=32):
        self.mfma_conv = bool(mfma_conv)
        # layers thinner than this stay with the vendor kernel (measured slower on csrc/conv_fwd.hip); the fixture test lowers it
        # to 1 so that EVERY stride-2 / transposed / small-plane layer of a narrow FlowNet runs on the hand-written kernel
        self.mfma_min_channels = int(mfma_min_channels)
        self.arena = Arena()
        self.own_winograd = os.environ.get("FFWM_FLOWNET_OWN_WINOGRAD", "1") != "0"
        self.thin_direct = os.environ.get("FFWM_FLOWNET_THIN_DIRECT", "1") != "0"
        self.fork = os.environ.get("FFWM_FLOWNET_FORK", "1") != "0"
        self._side = None
        self._thin = {}
        self._wino = {}
        if net.training:
            raise ValueError("FoldedFlowNet folds eval-mode BatchNorm statistics: call net.eval() first")
        p = next(net.parameters())
        if not p.is_cuda or p.dtype != torch.float32:
            raise NotImplementedError("FoldedFlowNet: float32 GPU networks only")
        self.device = p.device
        self.blocks = {}
        for name, mod in net.named_children():
            if isinstance(mod, nn.Sequential) and len(mod) >= 2 and isinstance(mod[1], nn.BatchNorm2d):
                conv = mod[0]
                w, b = _fold(conv, mod[1])
                slope = mod[2].negative_slope if len(mod) > 2 and isinstance(mod[2], nn.LeakyReLU) else 0.2
                self.blocks[name] = (isinstance(conv, nn.ConvTranspose2d), w, b, conv.stride, conv.padding, slope)
        self.heads = {L: (getattr(net, "predict_flow%d" % L)[0].weight.detach().contiguous(),
                          getattr(net, "predict_flow%d" % L)[0].bias.detach().contiguous()) for L in range(7)}
        self.ups = {L: (getattr(net, "upsampled_flow%d_to_%d" % (L + 1, L)).weight.detach().contiguous(),
                        getattr(net, "upsampled_flow%d_to_%d" % (L + 1, L)).bias.detach().contiguous()) for L in range(6)}
        self.use_graph = bool(graph)
        self._graph = None
        self._static_in = None
        self._static_out = None

    # conv (no bias: it is added by the epilogue) -> bias + LeakyReLU, in place and / or into a cat slice
    def _block(self, name, x, dst=None, dst2=None):
        transposed, w, b, stride, padding, slope = self.blocks[name]
        if (self.thin_direct and dst is None and dst2 is None and not transposed and tuple(w.shape[2:]) == (3, 3) and stride[0] == 1
                and padding[0] == 1 and x.size(1) <= 18 and w.size(0) % 8 == 0 and x.size(2) * x.size(3) >= 4096 and x.is_contiguous()):
            # the thin full-resolution layers (conv0 6 -> 64, inter_conv0 18 -> 16): direct kernel, a pixel and 16 output channels per lane
            wt = self._thin.get(name)
            if wt is None:
                wt = self._thin[name] = thin_weights(w)
            return conv_thin(x, wt, b, LRELU, slope)
        if self.mfma_conv and (transposed or stride[0] == 2 or x.size(2) <= 32) and x.size(1) >= self.mfma_min_channels:
            # the layers MIOpen wraps in layout transposes (stride-2 convolutions, transposed convolutions), the
            # weight-streaming 2 x 2 ... 8 x 8 tail and the 16 x 16 / 32 x 32 stride-1 layers (measured per layer,
            # tools/conv_layers.py: faster than the vendor kernel + epilogue everywhere except the >= 64 x 64 stride-1
            # layers and the thin 18 / 34-channel ones): hand-written MFMA kernel with the epilogue fused
            return conv_mfma(x, w, b, stride[0], padding[0], transposed, LRELU, slope, dst=dst, dst2=dst2, arena=self.arena)
        if (self.own_winograd and not transposed and stride[0] == 1 and padding[0] == 1 and tuple(w.shape[2:]) == (3, 3)
                and dst is None and dst2 is None):
            # the >= 64 x 64 stride-1 layers (conv0, conv1_1, inter_conv1, inter_conv0): the library's fp32 Winograd F(2x2, 3x3) kernel
            # with bias + LeakyReLU in its output transform; the transformed (folded) weights are kept per layer
            return ops.conv3x3_winograd(x, w, b, act=LRELU, slope=slope, frozen=self._wino.setdefault(name, {}))
        h = F.conv_transpose2d(x, w, None, stride, padding) if transposed else F.conv2d(x, w, None, stride, padding)
        if dst is None and dst2 is None:
            return bias_act(h, b, LRELU, slope=slope)
        bias_act(h, b, LRELU, y=dst if dst is not None else h, y2=dst2, slope=slope)
        return dst if dst is not None else h

    def _forward(self, x):
        B = x.size(0)
        self.arena.begin(x.device)
        f = self._block("conv0", x)
        skips = {}
        cats = {}
        for L in range(1, 7):
            f = self._block("conv%d" % L, f)
            if 3 <= L <= 5:
                # the level's concatenation buffer (skip, deconv output, upsampled flow): the encoder epilogue writes the
                # skip in place AND into its slice
                cs, hs = self.blocks["conv%d_1" % L][1].size(0), f.size(2)
                cd = self.blocks["deconv%d" % L][1].size(1)
                cats[L] = torch.empty(B, cs + cd + 2, hs, hs, device=x.device, dtype=x.dtype)
                f = self._block("conv%d_1" % L, f, dst=None, dst2=cats[L][:, :cs])
            else:
                f = self._block("conv%d_1" % L, f)
            skips[L] = f
        # Decoder.  deconv_L reads the concatenation of level L + 1 and so does inter_conv_{L+1} -> predict_flow_{L+1} -> upsampled_flow: two
        # independent chains per level that meet in level L's concatenation buffer.  Round 6: the transposed convolution runs on a side stream
        # (a parallel branch of the captured graph) next to the flow chain -- each kernel alone is latency-bound at these sizes.
        flows = {}
        cur = torch.cuda.current_stream(x.device)
        side = None
        if self.fork:
            if self._side is None:
                self._side = torch.cuda.Stream(device=x.device)
            side = self._side
        feat, cat = f, None
        for L in range(5, -1, -1):
            cd = self.blocks["deconv%d" % L][1].size(1)
            hs = feat.size(2) * 2
            if L >= 3:
                buf = cats[L]
                cs = buf.size(1) - cd - 2
            else:
                cs = 0
                buf = torch.empty(B, cd + 2, hs, hs, device=x.device, dtype=x.dtype)
            if side is not None:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    self._block("deconv%d" % L, feat, dst=buf[:, cs:cs + cd])
            else:
                self._block("deconv%d" % L, feat, dst=buf[:, cs:cs + cd])
            if L == 5:
                flow = flow_head(f, *self.heads[6])
            else:
                flow = flows[L + 1] = flow_head(self._block("inter_conv%d" % (L + 1), cat), *self.heads[L + 1])
            flow_up(flow, *self.ups[L], out=buf[:, cs + cd:])
            if side is not None:
                cur.wait_stream(side)
            feat = cat = buf
        flows[0] = flow_head(self._block("inter_conv0", cat), *self.heads[0])
        return flows[0], flows[1], flows[2]

    @torch.no_grad()
    def __call__(self, x):
        if not self.use_graph:
            return self._forward(x)
        if self._graph is None or self._static_in.shape != x.shape:
            self._static_in = x.clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):                     # warm-up outside the capture (MIOpen solver selection, allocations)
                    self._forward(self._static_in)
            torch.cuda.current_stream().wait_stream(s)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_out = self._forward(self._static_in)
        if x.data_ptr() != self._static_in.data_ptr():
            self._static_in.copy_(x, non_blocking=True)
        self._graph.replay()
        return self._static_out
