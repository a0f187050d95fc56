"""Networks on either side of the warp hot path (SURVEY section 8 rows a7/a8 and the train-step harness).

Architectures follow /root/reference/models/base_networks.py (FlowNet :59-165, FFWM netG :274-347,
MSDiscriminator :354-437), /root/reference/lightcnn/light_cnn.py:82-129 (LightCNN-29) and the VGG19
slicing of /root/reference/models/losses.py:398-519, with the SAME parameter/buffer names, so a reference
``state_dict`` loads unchanged (``FlowNet``: 243 keys, ``FFWM(sn=True)``: 383 keys).  The code itself is
table-driven and new; dense conv stacks stay on PyTorch-ROCm (MIOpen / hipBLASLt, i.e. MFMA), the
flow-guided warp inside netG's warp-attention module goes through the hand-written HIP kernel
(``WarpFlipCat``: grid_sample + flip + concat in one pass).
"""
import math

import os

import os as _os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import spectral_norm

from . import _ext
from . import conv as _conv
from .external_function import WarpFlipCat, WarpNet, warp_many

LRELU = 0.2


def _msra(mods):
    """kaiming_normal_ on conv / deconv weights, zero bias (base_networks.py:8-24)."""
    for m in mods:
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            nn.init.kaiming_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


# =============================================================================== FlowNet
def _cbl(cin, cout, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, stride, 1, bias=True), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(LRELU, inplace=True))


def _dbl(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(LRELU, inplace=True))


def _flow_head(cin):
    return nn.Sequential(nn.Conv2d(cin, 2, 3, 1, 1, bias=True), nn.Tanh())


class FlowNet(nn.Module):
    """7-level conv encoder + deconv decoder predicting dense 2-channel flow at 128/64/32
    (tanh -> absolute normalised sampling coordinates).  forward(x[B,3,128,128]) ->
    (flow128, flow64, flow32).  The reference's never-used ``inter_conv_occ*`` layers are kept
    (``keep_unused=True``) only for state-dict compatibility; they take no part in forward."""

    def __init__(self, ngf=64, in_ch=3, keep_unused=True):
        super().__init__()
        g = ngf
        # encoder: name -> (cin, cout, stride)
        enc = [("conv0", in_ch, g, 1), ("conv1", g, g, 2), ("conv1_1", g, 2 * g, 1),
               ("conv2", 2 * g, 2 * g, 2), ("conv2_1", 2 * g, 2 * g, 1), ("conv3", 2 * g, 4 * g, 2),
               ("conv3_1", 4 * g, 4 * g, 1), ("conv4", 4 * g, 8 * g, 2), ("conv4_1", 8 * g, 8 * g, 1),
               ("conv5", 8 * g, 8 * g, 2), ("conv5_1", 8 * g, 8 * g, 1), ("conv6", 8 * g, 16 * g, 2),
               ("conv6_1", 16 * g, 16 * g, 1)]
        for name, cin, cout, s in enc:
            setattr(self, name, _cbl(cin, cout, s))
        # decoder level L = 5..0: concat width, deconv output width
        cat = {5: 16 * g + 2, 4: 12 * g + 2, 3: 6 * g + 2, 2: g + 2, 1: g // 2 + 2, 0: g // 4 + 2}
        dec_out = {5: 8 * g, 4: 4 * g, 3: 2 * g, 2: g, 1: g // 2, 0: g // 4}
        dec_in = {5: 16 * g, 4: cat[5], 3: cat[4], 2: cat[3], 1: cat[2], 0: cat[1]}
        for L in range(5, -1, -1):
            setattr(self, "deconv%d" % L, _dbl(dec_in[L], dec_out[L]))
        for L in range(5, -1, -1):
            setattr(self, "inter_conv%d" % L, _cbl(cat[L], dec_out[L]))
        if keep_unused:
            for L in range(5, -1, -1):
                setattr(self, "inter_conv_occ%d" % L, _cbl(cat[L] - 1, dec_out[L]))
        heads = {6: 16 * g, 5: 8 * g, 4: 4 * g, 3: 2 * g, 2: g, 1: g // 2, 0: g // 4}
        for L in range(6, -1, -1):
            setattr(self, "predict_flow%d" % L, _flow_head(heads[L]))
        for L in range(6, 0, -1):
            setattr(self, "upsampled_flow%d_to_%d" % (L, L - 1), nn.ConvTranspose2d(2, 2, 4, 2, 1))
        _msra(self.modules())

    def unused_parameters(self):
        """Parameters that never receive a gradient (excluded from the DP gradient buckets)."""
        return [p for n, p in self.named_parameters() if n.startswith("inter_conv_occ")]

    def forward(self, x):
        f = self.conv0(x)
        skips = {}
        for L in range(1, 7):
            f = getattr(self, "conv%d_1" % L)(getattr(self, "conv%d" % L)(f))
            skips[L] = f
        flow = self.predict_flow6(f)
        flows = {}
        cat = None
        for L in range(5, -1, -1):
            up = getattr(self, "upsampled_flow%d_to_%d" % (L + 1, L))(flow)
            d = getattr(self, "deconv%d" % L)(f if L == 5 else cat)
            # levels 5..3 also concatenate the encoder skip (base_networks.py:129-151)
            cat = torch.cat((skips[L], d, up) if L >= 3 else (d, up), 1)
            flow = getattr(self, "predict_flow%d" % L)(getattr(self, "inter_conv%d" % L)(cat))
            flows[L] = flow
        return flows[0], flows[1], flows[2]


# =============================================================================== FFWM generator
class Tanh2(nn.Module):
    def forward(self, x):
        return (torch.tanh(x) + 1) / 2


def _activ(name):
    table = {"relu": nn.ReLU, "lrelu": lambda: nn.LeakyReLU(LRELU), "sigmoid": nn.Sigmoid, "tanh": nn.Tanh,
             "tanh2": Tanh2}
    if name not in table:
        raise NotImplementedError("Activation %s not implemented" % name)
    return table[name]()


def _sn(m, on):
    return spectral_norm(m) if on else m


class ResidualBlock(nn.Module):
    """activ(blocks(x) + input(x)); attribute names (activ / input / blocks) match the reference
    (base_networks.py:207-233)."""

    def __init__(self, inc, outc=None, kernel=3, activ="lrelu", sn=True):
        super().__init__()
        outc = inc if outc is None else outc
        pad = kernel // 2 if sn else kernel   # the reference's non-SN branch pads by `kernel`
        self.activ = _activ(activ)
        self.input = _sn(nn.Conv2d(inc, outc, 1, 1, 0), sn)
        self.blocks = nn.Sequential(_sn(nn.Conv2d(inc, outc, kernel, 1, pad), sn), nn.BatchNorm2d(outc),
                                    nn.LeakyReLU(LRELU), _sn(nn.Conv2d(outc, outc, kernel, 1, pad), sn),
                                    nn.BatchNorm2d(outc))

    def forward(self, x):
        return self.activ(self.blocks(x) + self.input(x))


class FusedResidualBlock(ResidualBlock):
    """The same block with its tail as one kernel (ffwm_amd/residual.py re-classes instances in place): in training mode on the GPU
    the last BatchNorm2d of `blocks`, the shortcut's bias, the add and the activation are ONE pass per direction (norm.bn_res_act,
    csrc/bn_lrelu.hip) -- through PyTorch-ROCm that tail is a BatchNorm kernel, the GEMM path's separate bias add and the add +
    activation: 142 us forward at 195 channels x 128 x 128 x 8 against the block's 970 us of convolutions; otherwise add + activation
    alone (csrc/residual.hip), or the plain composition."""

    def forward(self, x):
        from .norm import RES_ACTS, bn_res_act, bn_res_act_ok
        from .residual import _act_of, add_act
        blocks, inp = self.blocks, self.input
        bn = blocks[-1] if isinstance(blocks, nn.Sequential) and len(blocks) >= 2 else None
        act, slope = _act_of(self.activ)
        # (a per-layer spectral-norm hook on the shortcut would be bypassed by calling _conv_forward: only without hooks, i.e. plain
        # layers or the batched spectral norm of spectral_norm.fuse_spectral_norm, which sets `weight` before the network runs)
        if (bn is not None and isinstance(inp, nn.Conv2d) and not inp._forward_pre_hooks and not inp._forward_hooks
                and bn_res_act_ok(bn, x, RES_ACTS.get(act)) and torch.is_grad_enabled()):
            h = x
            for m in list(blocks)[:-1]:
                h = m(h)
            # the shortcut WITHOUT its bias (added inside the fused kernel); `inp.weight` is the spectral-norm product of this forward
            s = inp._conv_forward(x, inp.weight, None)
            if h.shape == s.shape and h.shape[1] == bn.num_features:
                return bn_res_act(h, bn, s, inp.bias, RES_ACTS[act], slope)
            return add_act(bn(h), s if inp.bias is None else s + inp.bias.view(1, -1, 1, 1), self.activ)
        return add_act(blocks(x), inp(x), self.activ)


def _conv_block(inc, outc, ks, s, p, activ="lrelu", res=0, bn=True, sn=True):
    layers = [_sn(nn.Conv2d(inc, outc, ks, s, p), sn)]
    if bn:
        layers.append(nn.BatchNorm2d(outc))
    if activ is not None:
        layers.append(_activ(activ))
    layers += [ResidualBlock(outc, activ=activ, sn=sn) for _ in range(res)]
    return nn.Sequential(*layers)


def _shuffle_block(inc, outc, sn=True):
    return nn.Sequential(_sn(nn.Conv2d(inc, outc * 4, 3, 1, 1), sn), nn.PixelShuffle(2), nn.BatchNorm2d(outc),
                         _activ("lrelu"))


class FFWM(nn.Module):
    """netG: e0-e3 encoder, PixelShuffle decoder, per-level warp-attention (warp encoder features
    with the flow, flip, concat, attention, multiply), residual refinement, three sigmoid
    reconstructions (32/64/128).  forward(x, flow=[flow32, flow64, flow128]).

    ``warp_flipcat`` is the hot-path hook: by default the fused HIP kernel (one pass for
    grid_sample + flip + cat); tests inject a torch reference to run the conv stacks on CPU."""

    def __init__(self, num_layers=3, isflip=True, sn=True, warp_flipcat=None):
        super().__init__()
        ch = [64, 64, 128, 256]
        dch = [256, 128, 64, 64]
        self.isflip = isflip
        dm = 3 if isflip else 2
        am = dm - 1
        self.layers = num_layers
        self.e0 = _conv_block(3, ch[0], 7, 1, 3, res=1, bn=False, sn=sn)
        for i in (1, 2, 3):
            setattr(self, "e%d" % i, _conv_block(ch[i - 1], ch[i], 4, 2, 1, res=1, sn=sn))
        d_in = [dch[0], dch[1] * dm, dch[2] * dm + 3]
        r_ch = [dch[1] * dm, dch[2] * dm + 3, dch[3] * dm + 3]
        for i in range(3):
            setattr(self, "d%d" % i, _shuffle_block(d_in[i], dch[i + 1], sn=sn))
        for i in range(3):
            setattr(self, "dres%d" % i, nn.Sequential(*[ResidualBlock(r_ch[i], activ="lrelu", sn=sn) for _ in range(2)]))
        for i in range(3):
            setattr(self, "rec%d" % i, _conv_block(r_ch[i], 3, 3, 1, 1, bn=False, activ="sigmoid", sn=sn))
        for i, c in enumerate((ch[2] * am, ch[1] * am, ch[0] * am)):
            setattr(self, "att%d" % i, nn.Sequential(_conv_block(c, c, 3, 1, 1, sn=sn),
                                                     ResidualBlock(c, c, activ="sigmoid", sn=sn)))
        self.warpNet = WarpNet()
        self._fused = warp_flipcat if warp_flipcat is not None else WarpFlipCat()
        # the HIP path: all levels' warps in one multi-problem launch (FFWM_WARP_MULTI=0: a launch per level, issued where the decoder needs
        # it -- its backward then runs right behind the attention convs' that produced its grad_output, on warm caches)
        self._multi = warp_flipcat is None and _os.environ.get("FFWM_WARP_MULTI", "1") != "0"
        self.fuse_gate = False                       # residual.fuse_residual: `skip * att_i(skip)` with its sigmoid tail as one kernel

    def _skip(self, feat, flow):
        if self.isflip:
            return self._fused(feat, flow)            # cat(w, flip(w, 3)) in one kernel
        return self.warpNet(feat, flow)

    def _skips(self, enc, flow):
        """warp + flip + cat of every level (base_networks.py:326-329).  The encoder features and the flows are all known
        once the encoder has run, so on the GPU the levels go out as ONE launch (external_function.warp_many) instead
        of one launch-bound kernel per level; the backward runs once, after the decoder's."""
        feats = [enc[self.layers - 1 - i] for i in range(self.layers)]
        if self._multi and feats[0].is_cuda:
            return warp_many(feats, [flow[i] for i in range(self.layers)], self.isflip)
        return [self._skip(f, flow[i]) for i, f in enumerate(feats)]

    def forward(self, x, flow=None, return_att=False):
        enc = [self.e0(x)]
        for i in range(1, self.layers + 1):
            enc.append(getattr(self, "e%d" % i)(enc[-1]))
        fdec = enc[-1]
        recons, att = [], None
        if callable(flow):
            # the encoder does not need the flows: a caller that computes them on another HIP stream passes a function that joins
            # that stream and returns them (trainer.FFWMTrainer.forward: flowNetF beside the encoder)
            flow = flow()
        skips = self._skips(enc, flow) if self._multi else None
        for i in range(self.layers):
            dec = getattr(self, "d%d" % i)(fdec)
            skip = skips[i] if skips is not None else self._skip(enc[self.layers - 1 - i], flow[i])
            if self.fuse_gate:
                from .residual import gated
                skip, att = gated(getattr(self, "att%d" % i), skip)
            else:
                att = getattr(self, "att%d" % i)(skip)
                skip = skip * att
            parts = [skip, dec]
            if recons:   # TP-GAN style: feed the lower-resolution reconstruction to the next level
                parts.append(F.interpolate(recons[-1], scale_factor=2, mode="bilinear"))
            fdec = getattr(self, "dres%d" % i)(torch.cat(parts, 1))
            recons.append(getattr(self, "rec%d" % i)(fdec))
        out = (recons[-3], recons[-2], recons[-1])
        return out + (att,) if return_att else out


class WarpAttention(nn.Module):
    """The warp-attention module of netG by itself (base_networks.py:323-333), all three levels: warp the encoder
    feature with the flow, flip, concatenate (one HIP kernel), run the attention convs `att_i` on it (conv block +
    sigmoid residual block, spectrally normalised: :283-291), multiply.  This is the "warp+attention path" of
    BASELINE.json's 3000 img/s target; bench.py times it forward + backward (`subpaths.warp_attention_path`).
    Same submodule names (`att0`..`att2`) and widths as FFWM, so its weights are a slice of a netG state dict."""

    LEVELS = ((128, 32), (64, 64), (64, 128))          # (encoder channels, plane size) of e2 / e1 / e0

    def __init__(self, sn=True, warp_flipcat=None):
        super().__init__()
        for i, (c, _) in enumerate(self.LEVELS):
            setattr(self, "att%d" % i, nn.Sequential(_conv_block(2 * c, 2 * c, 3, 1, 1, sn=sn),
                                                     ResidualBlock(2 * c, 2 * c, activ="sigmoid", sn=sn)))
        self._fused = warp_flipcat if warp_flipcat is not None else WarpFlipCat()
        self._multi = warp_flipcat is None
        self.fuse_gate = False                       # residual.fuse_residual

    def forward(self, feats, flows):
        if self._multi and feats[0].is_cuda:
            skips = warp_many(list(feats), list(flows), True)
        else:
            skips = [self._fused(feat, flow) for feat, flow in zip(feats, flows)]
        if self.fuse_gate:
            from .residual import gated
            return [gated(getattr(self, "att%d" % i), skip)[0] for i, skip in enumerate(skips)]
        return [skip * getattr(self, "att%d" % i)(skip) for i, skip in enumerate(skips)]


# =============================================================================== discriminator
class MSDiscriminator(nn.Module):
    """Multi-scale spectral-norm patch discriminator (base_networks.py:354-437): one 3-conv net per
    scale (factor-2 pyramid down to 16 px), score maps bilinearly resized to the finest and summed."""

    def __init__(self, real_crop_size=128, inc=3, max_n_scales=9, scale_factor=2, base_channels=64, sigmoid=False):
        super().__init__()
        self.scale_factor = scale_factor
        n = int(math.ceil(math.log(real_crop_size / 16.0) / math.log(scale_factor)))
        self.max_n_scales = min(n, max_n_scales)
        self.nets = nn.ModuleList(self._make(inc, base_channels, sigmoid) for _ in range(self.max_n_scales))

    @staticmethod
    def _make(inc, c, sigmoid):
        layers = []
        for cin, cout in ((inc, c), (c, 2 * c), (2 * c, 4 * c)):
            layers += [spectral_norm(nn.Conv2d(cin, cout, 3, 2, 1)), nn.BatchNorm2d(cout), nn.LeakyReLU(LRELU, True)]
        if sigmoid:
            layers += [spectral_norm(nn.Conv2d(4 * c, 1, 1)), nn.Sigmoid()]
        else:
            layers += [nn.Conv2d(4 * c, 1, 1)]
        return nn.Sequential(*layers)

    def forward(self, x):
        total = self.nets[0](x)
        size = total.shape[2:]
        for i, net in enumerate(self.nets[1:5], start=1):
            small = F.interpolate(x, scale_factor=self.scale_factor ** (-i), mode="bilinear")
            total = total + F.interpolate(net(small), size=size, mode="bilinear")
        return total


# bias + activation of the frozen feature networks as one kernel (csrc/mfm.hip); FFWM_FUSED_ACT=0 keeps the module path
FUSED_ACTIVATIONS = os.environ.get("FFWM_FUSED_ACT", "1") != "0"


# =============================================================================== LightCNN-29
_WINOGRAD_MFM = os.environ.get("FFWM_WINOGRAD_MFM", "0") == "1"      # LightCNN's 48 / 96-channel layers: measured 0.4 ms per step slower than the vendor


class mfm(nn.Module):
    """max-feature-map: conv/linear to 2*out channels, elementwise max of the halves."""

    def __init__(self, cin, cout, k=3, s=1, p=1, type=1):
        super().__init__()
        self.out_channels = cout
        self.filter = nn.Conv2d(cin, 2 * cout, k, s, p) if type == 1 else nn.Linear(cin, 2 * cout)

    def forward(self, x):
        f = self.filter
        if FUSED_ACTIVATIONS and x.is_cuda and x.dtype == torch.float32:
            # the layer without its bias, then bias + max-feature-map as one kernel per direction (csrc/mfm.hip)
            from .external_function import MaxFeatureMapFunction
            if isinstance(f, nn.Conv2d):
                if (f.kernel_size == (3, 3) and f.stride == (1, 1) and f.padding == (1, 1) and f.dilation == (1, 1) and f.groups == 1
                        and _WINOGRAD_MFM and _conv.winograd_ok(x, f.weight)):
                    h = _conv.winograd_conv(x, f)          # large planes: csrc/conv_winograd.hip
                else:
                    h = F.conv2d(x, f.weight, None, f.stride, f.padding, f.dilation, f.groups)
            else:
                h = F.linear(x, f.weight, None)
            ext = _ext.get()
            if ext is not None:
                return ext.mfm(h.contiguous(), f.bias)
            return MaxFeatureMapFunction.apply(h.contiguous(), f.bias)
        a, b = torch.split(f(x), self.out_channels, 1)
        return torch.max(a, b)


class group(nn.Module):
    def __init__(self, cin, cout, k, s, p):
        super().__init__()
        self.conv_a = mfm(cin, cin, 1, 1, 0)
        self.conv = mfm(cin, cout, k, s, p)

    def forward(self, x):
        return self.conv(self.conv_a(x))


class resblock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv1 = mfm(cin, cout)
        self.conv2 = mfm(cin, cout)

    def forward(self, x):
        return self.conv2(self.conv1(x)) + x


class LightCNN29(nn.Module):
    """LightCNN-29 identity-feature extractor (lightcnn/light_cnn.py:82-129): forward(gray[B,1,128,128])
    -> (logits, fc[B,256], pool[B,128,8,8])."""

    def __init__(self, num_classes=79077):
        super().__init__()
        self.conv1 = mfm(1, 48, 5, 1, 2)
        widths = [(48, 96, 1), (96, 192, 2), (192, 128, 3), (128, 128, 4)]
        for i, (cin, cout, nblk) in enumerate(widths, start=1):
            setattr(self, "block%d" % i, nn.Sequential(*[resblock(cin, cin) for _ in range(nblk)]))
            setattr(self, "group%d" % i, group(cin, cout, 3, 1, 1))
        for i in (1, 2, 3, 4):
            setattr(self, "pool%d" % i, nn.MaxPool2d(2, 2, ceil_mode=True))
        self.fc = mfm(8 * 8 * 128, 256, type=0)
        self.fc2 = nn.Linear(256, num_classes)

    def forward(self, x):
        x = self.pool1(self.conv1(x))
        x = self.pool2(self.group1(self.block1(x)))
        x = self.pool3(self.group2(self.block2(x)))
        x = self.group3(self.block3(x))
        p = self.pool4(self.group4(self.block4(x)))
        fc = F.dropout(self.fc(p.flatten(1)), training=self.training)
        return self.fc2(fc), fc, p


# =============================================================================== VGG19 features
_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512]
_VGG_SLICES = [("relu1_1", 0, 2), ("relu1_2", 2, 4), ("relu2_1", 4, 7), ("relu2_2", 7, 9), ("relu3_1", 9, 12),
               ("relu3_2", 12, 14), ("relu3_3", 14, 16), ("relu3_4", 16, 18), ("relu4_1", 18, 21),
               ("relu4_2", 21, 23), ("relu4_3", 23, 25), ("relu4_4", 25, 27), ("relu5_1", 27, 30),
               ("relu5_2", 30, 32), ("relu5_3", 32, 34), ("relu5_4", 34, 36)]


class VGG19(nn.Module):
    """VGG19 ``features`` sliced at every ReLU (losses.py:398-519), frozen.  torchvision and its
    pretrained weights are not available offline: weights are seeded random (same architecture, same
    FLOPs and memory traffic; the loss VALUES differ from a pretrained VGG, which the benchmark
    states).  ``upto`` stops the stack at the last slice a caller needs."""

    def __init__(self, upto="relu5_1"):
        super().__init__()
        feats, cin = [], 3
        for v in _VGG19_CFG:
            if v == "M":
                feats.append(nn.MaxPool2d(2, 2))
            else:
                feats += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
                cin = v
        self.slices = []
        for name, a, b in _VGG_SLICES:
            seq = nn.Sequential()
            for idx in range(a, b):
                seq.add_module(str(idx), feats[idx])
            setattr(self, name, seq)
            self.slices.append(name)
            if name == upto:
                break
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        out = {}
        fused = FUSED_ACTIVATIONS and x.is_cuda and x.dtype == torch.float32
        for name in self.slices:
            if fused:
                # conv without its bias, then bias + ReLU as one pass (csrc/mfm.hip: bias_relu) instead of the vendor
                # library's separate bias add followed by the activation's own read-modify-write
                from .external_function import BiasReLUFunction
                mods = list(getattr(self, name).children())
                i = 0
                while i < len(mods):
                    m = mods[i]
                    if isinstance(m, nn.Conv2d) and i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and m.bias is not None:
                        if not m.weight.requires_grad and _conv.winograd_ok(x, m.weight, 1):
                            # large planes: conv + bias + ReLU as ONE launch of the Winograd MFMA kernel (csrc/conv_winograd.hip)
                            x = _conv.winograd_bias_relu(x, m)
                            i += 2
                            continue
                        h = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)
                        ext = _ext.get()
                        x = ext.bias_relu(h, m.bias) if ext is not None else BiasReLUFunction.apply(h, m.bias)
                        i += 2
                    else:
                        x = m(x)
                        i += 1
            else:
                x = getattr(self, name)(x)
            out[name] = x
        return out


# =============================================================================== guided filter
class GuidedFilter(nn.Module):
    """Guided filter with box radius r (external_function.py:239-277): box sums by cumsum
    differences, all in PyTorch (HBM-bound; SURVEY section 8(f) rank 3 lists a fused kernel as next)."""

    def __init__(self, r, eps=1e-8):
        super().__init__()
        self.r, self.eps = r, eps

    @staticmethod
    def _diff(c, r, dim):
        n = c.size(dim)
        head = c.narrow(dim, r, r + 1)
        mid = c.narrow(dim, 2 * r + 1, n - 2 * r - 1) - c.narrow(dim, 0, n - 2 * r - 1)
        tail = c.narrow(dim, n - 1, 1) - c.narrow(dim, n - 2 * r - 1, r)
        return torch.cat((head, mid, tail), dim)

    def box(self, x):
        return self._diff(self._diff(x.cumsum(2), self.r, 2).cumsum(3), self.r, 3)

    def forward(self, x, y):
        h, w = x.shape[2:]
        assert h > 2 * self.r + 1 and w > 2 * self.r + 1
        n = self.box(x.new_ones((1, 1, h, w)))
        mean_x, mean_y = self.box(x) / n, self.box(y) / n
        cov_xy = self.box(x * y) / n - mean_x * mean_y
        var_x = self.box(x * x) / n - mean_x * mean_x
        a = cov_xy / (var_x + self.eps)
        b = mean_y - a * mean_x
        return (self.box(a) / n) * x + self.box(b) / n
