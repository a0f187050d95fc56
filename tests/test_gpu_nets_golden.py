"""Rows a7 / a8 on the MI355X against the REFERENCE's modules (tests/golden/reference_modules.pt, produced by importing
/root/reference/models/base_networks.py -- tests/golden/make_golden.py) -- run with ``-m gpu``.

* `nets.FlowNet(4)` (base_networks.py:59-165) eval and train mode on the GPU with the fused BatchNorm + LeakyReLU kernel
  FORCED for every pair (the production size gate of 1 M elements is lifted), against `flownet4_eval` / `flownet4_train`.
* `nets.FFWM(sn=True)` (base_networks.py:274-347) eval on the GPU with the HIP `WarpFlipCat` kernel on the warp-attention
  path (:323-333) and the batched spectral-norm kernels, against `ffwm_eval`.
* full-size property runs: FlowNet(64) forward at batch 6 (BASELINE configs[1]) and one FFWM train step at batch 8
  (configs[2]): finite, and equal to the unfused PyTorch paths.

Tolerance: 1e-4 max abs diff (the north_star's fp32 bound) on tanh / sigmoid outputs in [-1, 1]; the CPU fixture test
(tests/test_nets_golden.py) holds 2e-5 with ATen's CPU convolutions, the GPU runs MIOpen's.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import fill  # noqa: E402

DEV = "cuda:0"
TOL = 1e-4


@pytest.fixture(scope="module")
def gold():
    return torch.load(os.path.join(HERE, "golden", "reference_modules.pt"))


def _sub(t, s):
    return t[..., ::s, ::s]


def _close(a, b, tol=TOL):
    d = (a.detach().cpu() - b).abs().max().item()
    assert d <= tol, "max abs diff %.3e > %.1e" % (d, tol)
    return d


@pytest.fixture
def every_bn_pair_fused(monkeypatch):
    from ffwm_amd import norm
    monkeypatch.setattr(norm, "MIN_FUSED_NUMEL", 0)
    return norm


def test_flownet4_on_gpu_matches_reference_fixture(gold, every_bn_pair_fused):
    import copy
    from ffwm_amd import nets
    norm = every_bn_pair_fused
    plain = fill.fill_module(nets.FlowNet(4)).to(DEV)          # nn.BatchNorm2d + nn.LeakyReLU, the stock GPU path
    net = copy.deepcopy(plain)
    fused = norm.fuse_bn_lrelu(net)
    assert fused >= 30, fused                      # every conv block of FlowNet is conv + BN + LeakyReLU(0.2)
    x = fill.image(2, 3, 128, 128, "flownet_in").to(DEV)
    with torch.no_grad():
        net.eval()                                  # eval mode: running statistics, the fused kernel stands aside
        f128, f64, f32 = net(x)
        g = gold["flownet4_eval"]
        _close(_sub(f128, 2), g["flow128_s2"])
        _close(f64, g["flow64"])
        _close(f32, g["flow32"])
        assert abs(f128.double().sum().item() - g["sum128"].item()) < 5e-2
        # train mode: batch statistics through bn_lrelu_fwd_kernel.  Batch 2 at ngf = 4 normalises the 2 x 2 level over
        # EIGHT values per channel, which amplifies fp32 rounding whatever kernel produced it: the stock GPU path (MIOpen
        # convolutions + ATen batch norm) itself sits 2-4e-4 from the CPU-made fixture.  So the fused kernel is held to
        # the accuracy of the stock path, measured against a float64 evaluation of the same network on the GPU, and to
        # the fixture within what the stock path deviates from it.
        ref64 = copy.deepcopy(plain).double().train()
        net.train()
        plain.train()
        fused_out, plain_out, f64_out = net(x), plain(x), ref64(x.double())
        g = gold["flownet4_train"]
        for a, b, r64, ref in zip(fused_out, plain_out, f64_out, (None, g["flow64"], g["flow32"])):
            err_fused = (a.double() - r64).abs().max().item()
            err_stock = (b.double() - r64).abs().max().item()
            assert err_fused <= max(TOL, 2 * err_stock), (err_fused, err_stock)
            if ref is not None:
                stock = (b.cpu() - ref).abs().max().item()
                _close(a, ref, TOL + 2 * stock)
        stock = (_sub(plain_out[0], 2).cpu() - g["flow128_s2"]).abs().max().item()
        _close(_sub(fused_out[0], 2), g["flow128_s2"], TOL + 2 * stock)
        assert stock <= 2e-3, stock
        _close(net.conv0[1].running_mean, g["bn_mean_conv0"], 1e-5)


def test_ffwm_generator_on_gpu_with_hip_warp_matches_reference_fixture(gold):
    """netG with the product's own warp: WarpFlipCat (csrc/warp.hip) replaces grid_sample + flip + cat of
    base_networks.py:326-329, fuse_spectral_norm replaces the 52 per-layer hooks."""
    from ffwm_amd import nets
    from ffwm_amd.external_function import WarpFlipCat
    from ffwm_amd.spectral_norm import fuse_spectral_norm
    netG = fill.fill_module(nets.FFWM(sn=True)).to(DEV).eval()
    assert isinstance(netG._fused, WarpFlipCat)
    fuse_spectral_norm(netG)
    img = fill.image(1, 3, 128, 128, "netG_in").to(DEV)
    flows = [fill.flow_field(1, s, s, "netG_flow%d" % s).to(DEV) for s in (32, 64, 128)]
    with torch.no_grad():
        r32, r64, r128, att = netG(img, flow=flows, return_att=True)
    g = gold["ffwm_eval"]
    _close(r32, g["rec32"])
    _close(r64, g["rec64"])
    _close(_sub(r128, 2), g["rec128_s2"])
    _close(_sub(att, 8), g["att_s8"])
    assert abs(r128.double().sum().item() - g["sum128"].item()) < 0.2
    assert abs(att.double().sum().item() - g["att_sum"].item()) < 2.0


def test_flownet64_batch6_forward_full_size_properties(every_bn_pair_fused):
    """BASELINE configs[1]: FlowNetF forward-only, batch 6, 128 x 128 -- finite, in tanh's range, shaped as
    base_networks.py:157-165 returns them, and the fused train-mode path equals nn.BatchNorm2d + nn.LeakyReLU."""
    import copy
    from ffwm_amd import nets
    norm = every_bn_pair_fused
    torch.manual_seed(0)
    plain = nets.FlowNet(64).to(DEV)
    fused = copy.deepcopy(plain)
    assert norm.fuse_bn_lrelu(fused) >= 30
    x = torch.rand(6, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        for mode in ("eval", "train"):
            getattr(plain, mode)()
            getattr(fused, mode)()
            a, b = plain(x), fused(x)
            for fa, fb, s in zip(a, b, (128, 64, 32)):
                assert tuple(fa.shape) == (6, 2, s, s)
                assert torch.isfinite(fb).all() and fb.abs().max().item() <= 1.0
                assert (fa - fb).abs().max().item() <= TOL, mode
    # the launch-lean eval path (BatchNorm folded into the convolutions) against the module path
    from ffwm_amd import flownet_eval
    lean = flownet_eval.FoldedFlowNet(plain.eval())
    with torch.no_grad():
        for fa, fb in zip(plain(x), lean(x)):
            assert (fa - fb).abs().max().item() <= TOL


def test_ffwm_train_step_batch8_full_size_properties():
    """BASELINE configs[2] at its real batch: one full FFWM train step (netG + netD + flowNetF/B, all losses, three
    optimisers) at batch 8 -- finite losses, and the hand-written fast paths agree with the plain PyTorch paths."""
    from ffwm_amd import trainer
    batch = trainer.synthetic_batch(8, DEV, seed=11)
    plain = trainer.FFWMTrainer(DEV, seed=2, mfma_wgrad=False, mfma_fwd=False, flat_adam=False, fused_bn=False, fused_spectral_norm=False,
                                batched_losses=False, capturable=False)
    plain.red_G.set_gather(False)
    plain.red_D.set_gather(False)
    fast = trainer.FFWMTrainer(DEV, seed=2)
    lp, lf = plain.step(batch), fast.step(batch)
    torch.cuda.synchronize()
    for k in lp:
        a, b = float(lp[k].detach()), float(lf[k].detach())
        assert a == a and b == b and abs(b) < 1e6, (k, a, b)
        assert abs(a - b) <= 2e-3 * (1 + abs(a)), (k, a, b)
    for p in fast.netG.parameters():
        assert torch.isfinite(p).all()


def test_gradient_arena_steps_match_per_call_zero_fills(monkeypatch):
    """conv._GradArena (round 5): inside a trainer's step the weight-gradient buffers are slices of ONE tensor cleared by one launch.
    Three steps with the arena against three steps of an identical trainer whose calls clear their own buffers: the same losses (the
    gradients differ only by the order of float atomics), the arena sized by the first step, served from the second on, never active
    outside a step."""
    from ffwm_amd import conv, trainer
    batch = trainer.synthetic_batch(2, DEV, seed=5)
    conv.GRAD_ARENA.__init__()
    monkeypatch.setattr(trainer, "_GRAD_ARENA_ON", False)
    ta = trainer.FFWMTrainer(DEV, seed=4, ngf=16)
    la = [{k: float(v.detach()) for k, v in ta.step(batch).items()} for _ in range(3)]
    assert conv.GRAD_ARENA.buf is None and not conv.GRAD_ARENA.active
    monkeypatch.setattr(trainer, "_GRAD_ARENA_ON", True)
    tb = trainer.FFWMTrainer(DEV, seed=4, ngf=16)
    lb = []
    for i in range(3):
        lb.append({k: float(v.detach()) for k, v in tb.step(batch).items()})
        torch.cuda.synchronize()
        assert not conv.GRAD_ARENA.active
        if i == 0:
            assert conv.GRAD_ARENA.buf is not None and conv.GRAD_ARENA.buf.numel() >= conv.GRAD_ARENA.need > 0
        else:
            assert 0 < conv.GRAD_ARENA.off <= conv.GRAD_ARENA.buf.numel() and conv.GRAD_ARENA.need <= conv.GRAD_ARENA.buf.numel()
    for i, (sa, sb) in enumerate(zip(la, lb)):
        for k in sa:
            # the first step sees identical weights; behind it two trainers drift apart by the order of their float atomics (the DP tests'
            # bound for that: 30 %)
            tol = 2e-3 if i == 0 else 0.3
            assert sa[k] == sa[k] and sb[k] == sb[k] and abs(sa[k] - sb[k]) <= tol * (1 + abs(sa[k])), (i, k, sa[k], sb[k])
    # the choosers themselves, inside and outside an arena step: 3x3 (Winograd-domain / direct kernel), tiled stride 2, transposed 4x4
    g = torch.Generator().manual_seed(3)
    cases = [("c3", 8, 64, 64, 64, 3, 1), ("c3s2", 4, 32, 32, 64, 3, 2), ("t4", 4, 32, 16, 48, 4, 2)]
    for kind, B, C, H, K, k, st in cases:
        x = torch.randn(B, C, H, H, generator=g).to(DEV)
        if kind == "t4":
            w = torch.randn(C, K, 4, 4, generator=g).to(DEV)
            go = torch.randn(B, K, 2 * H, 2 * H, generator=g).to(DEV)
            fn = lambda: conv.conv_transpose_weight_grad(x, go, w, True)
        else:
            w = torch.randn(K, C, 3, 3, generator=g).to(DEV)
            Ho = (H + 2 - 3) // st + 1
            go = torch.randn(B, K, Ho, Ho, generator=g).to(DEV)
            fn = lambda: conv.conv_weight_grad(x, go, w, st, 1, True)
        ref = [t.clone() for t in fn()]
        conv.GRAD_ARENA.begin(torch.device(DEV))
        used = conv.GRAD_ARENA.off
        got = [t.clone() for t in fn()]
        got2 = [t.clone() for t in fn()]                 # a second call of the same layer in the same step gets its own slice
        assert conv.GRAD_ARENA.off > used, kind
        conv.GRAD_ARENA.end()
        for r, a, b in zip(ref, got, got2):
            scale = float(r.abs().max())
            assert float((r - a).abs().max()) <= 2e-5 * scale and float((r - b).abs().max()) <= 2e-5 * scale, kind
    assert conv.GRAD_ARENA.take(16, torch.empty(1, device=DEV)) is None          # outside a step: callers clear their own buffers
    # a slice handed out is zero, and the whole arena is zero again after the next step's begin()
    conv.GRAD_ARENA.begin(torch.device(DEV))
    s0 = conv.GRAD_ARENA.take(1000, torch.empty(1, device=DEV))
    assert s0 is not None and float(s0.abs().max()) == 0.0 and float(conv.GRAD_ARENA.buf.abs().max()) == 0.0
    conv.GRAD_ARENA.end()


# ------------------------------------------------------------------------------------------------ the product's OWN conv kernels
def _launch_counts(fn):
    """run fn with the library's launch profiler on; -> {scope name: launches}"""
    from ffwm_amd import _lib
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        _lib.prof_enable(False)
    return out, {k: v["launches"] for k, v in _lib.prof_collect().items()}


def _route_all(net, monkeypatch):
    """every route the trainer applies (trainer.py:111-130), with the size gates of the Winograd route lifted so that the
    small fixture batch takes the hand-written kernels wherever a production batch would"""
    from ffwm_amd import conv
    monkeypatch.setattr(conv, "WINOGRAD_MIN_PAIRS", 1)
    monkeypatch.setattr(conv, "WINOGRAD_MIN_TILES", 64)
    n_w = conv.route_conv_wgrad(net)
    n_wino = conv.route_conv_winograd(net)
    n_fwd = conv.route_conv_fwd(net)
    from ffwm_amd.residual import fuse_residual
    assert fuse_residual(net) >= 13 and net.fuse_gate        # add + activation of the residual blocks, the warp-attention gate
    return n_w, n_wino, n_fwd


def test_ffwm_generator_on_the_routed_conv_kernels_matches_reference_fixture(gold, monkeypatch):
    """Row a8 DIRECTLY: nets.FFWM(sn=True) with route_conv_wgrad + route_conv_winograd + route_conv_fwd applied -- the
    Winograd MFMA kernel (csrc/conv_winograd.hip), the direct MFMA kernel (csrc/conv_fwd.hip), the HIP warp and the batched
    spectral norm -- against the reference module's outputs (`ffwm_eval`, base_networks.py:274-347), <= 1e-4."""
    from ffwm_amd import nets
    from ffwm_amd.spectral_norm import fuse_spectral_norm
    netG = fill.fill_module(nets.FFWM(sn=True)).to(DEV).eval()
    n_w, n_wino, n_fwd = _route_all(netG, monkeypatch)
    assert n_wino >= 30 and n_fwd >= 3, (n_w, n_wino, n_fwd)
    fuse_spectral_norm(netG)
    img = fill.image(1, 3, 128, 128, "netG_in").to(DEV)
    flows = [fill.flow_field(1, s, s, "netG_flow%d" % s).to(DEV) for s in (32, 64, 128)]

    def run():
        with torch.no_grad():
            return netG(img, flow=flows, return_att=True)
    (r32, r64, r128, att), launches = _launch_counts(run)
    # the hand-written kernels really ran: Winograd forward for the 3x3 / stride-1 layers, conv_fwd for e1-e3
    assert sum(v for k, v in launches.items() if k.startswith("conv_winograd_fwd")) >= 25, launches      # (+ "_split": few-pair calls)
    assert sum(v for k, v in launches.items() if k.startswith("conv_fwd")) >= 3, launches
    g = gold["ffwm_eval"]
    _close(r32, g["rec32"])
    _close(r64, g["rec64"])
    _close(_sub(r128, 2), g["rec128_s2"])
    _close(_sub(att, 8), g["att_s8"])
    assert abs(r128.double().sum().item() - g["sum128"].item()) < 0.2


def test_ffwm_generator_routed_gradients_layer_by_layer(monkeypatch):
    """forward + backward of netG (base_networks.py:274-347, spectral norm, train mode) through EVERY route the trainer applies
    (Winograd forward / data gradient, conv_wgrad.hip / tiled conv_bwd.hip weight gradients, conv_fwd.hip): each convolution's
    output, data gradient, weight gradient and bias gradient -- as the routed kernels produced them inside the real pass -- against
    ATen's float64 convolution / convolution_backward on the SAME layer input, effective weight and grad_output.  (End to end the
    backward of this net at batch 2 amplifies fp32 rounding by orders of magnitude, whatever kernels run: layer-local is the
    comparison that isolates the kernels.)"""
    import torch.nn as nn
    import torch.nn.functional as F
    from ffwm_amd import conv as cv, nets
    net = fill.fill_module(nets.FFWM(sn=True)).to(DEV).train()
    _route_all(net, monkeypatch)
    assert cv.route_conv_bwd(net) >= 0
    img = fill.image(2, 3, 128, 128, "netG_in").to(DEV)
    flows = [fill.flow_field(2, s, s, "netG_flow%d" % s).to(DEV) for s in (32, 64, 128)]
    gos = [fill.image(2, 3, s, s, "go%d" % s).to(DEV) for s in (32, 64, 128)]
    rec = {}

    def fwd_hook(name):
        def hook(m, inp, out):
            w = m.weight                                   # the spectral-norm product of this forward (a non-leaf tensor)
            d = rec.setdefault(name, {"m": m})
            d.update(x=inp[0].detach(), w=w.detach(), y=out.detach(), b=None if m.bias is None else m.bias.detach())
            if w.requires_grad:
                w.register_hook(lambda g, d=d: d.__setitem__("gw", g.detach()))
            if m.bias is not None and m.bias.requires_grad:
                m.bias.register_hook(lambda g, d=d: d.__setitem__("gb", g.detach()))
            out.register_hook(lambda g, d=d: d.__setitem__("go", g.detach()))
        return hook
    convs = [(n, m) for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
    def bwd_hook(name):
        def hook(m, gin, gout):                            # this module's own d(input), whoever else consumes the tensor
            if gin[0] is not None:
                rec.setdefault(name, {"m": m})["gx"] = gin[0].detach()
        return hook
    for n, m in convs:
        m.register_forward_hook(fwd_hook(n))
        m.register_full_backward_hook(bwd_hook(n))

    def run():
        outs = net(img, flow=flows)
        torch.autograd.backward(list(outs), gos)
    _, launches = _launch_counts(run)
    assert sum(v for k, v in launches.items() if k.startswith("conv_winograd_dgrad")) >= 20 and launches.get("conv_wgrad_mfma_tiled", 0) >= 10, launches
    checked = checked_x = 0
    for n, d in rec.items():
        m = d["m"]
        if "go" not in d:
            continue
        x64, w64, go64 = d["x"].double(), d["w"].double(), d["go"].double()
        b64 = None if d["b"] is None else d["b"].double()
        y64 = F.conv2d(x64, w64, b64, m.stride, m.padding, m.dilation, m.groups)
        tol = lambda ref: 1e-4 * (1 + ref.abs().max().item())
        assert (d["y"].double() - y64).abs().max().item() <= tol(y64), ("forward", n)
        gx64, gw64, gb64 = torch.ops.aten.convolution_backward(go64, x64, w64, [w64.shape[0]], list(m.stride), list(m.padding), list(m.dilation),
                                                               False, [0, 0], m.groups, [True, True, True])
        if "gx" in d:
            assert (d["gx"].double() - gx64).abs().max().item() <= 1e-4 * (1 + gx64.abs().max().item()), ("data gradient", n)
            checked_x += 1
        if "gw" in d:
            scale = (x64.abs().max() * go64.abs().max()).item() * (go64[0, 0].numel() * go64.shape[0]) ** 0.5
            assert (d["gw"].double() - gw64).abs().max().item() <= 2e-6 * scale + tol(gw64) * 1e-1, ("weight gradient", n)
            checked += 1
        if "gb" in d:
            scale = go64.abs().max().item() * (go64[0, 0].numel() * go64.shape[0]) ** 0.5
            assert (d["gb"].double() - gb64).abs().max().item() <= 2e-6 * scale + 1e-5 * gb64.abs().max().item(), ("bias gradient", n)
    assert checked >= 45, checked
    assert checked_x >= 45, checked_x


def test_folded_flownet4_on_the_conv_fwd_kernel_matches_reference_fixture(gold):
    """Row a7 DIRECTLY: the launch-lean eval path (BatchNorm folded, bias + LeakyReLU epilogues, HIP flow heads / upsamplers) with
    EVERY stride-2 / transposed / small-plane convolution on csrc/conv_fwd.hip (channel gate lifted: FlowNet(4) is 4-64
    channels wide) against the reference module's outputs (`flownet4_eval`, base_networks.py:116-165)."""
    from ffwm_amd import flownet_eval, nets
    net = fill.fill_module(nets.FlowNet(4)).to(DEV).eval()
    lean = flownet_eval.FoldedFlowNet(net, mfma_min_channels=1)
    x = fill.image(2, 3, 128, 128, "flownet_in").to(DEV)
    (f128, f64, f32), launches = _launch_counts(lambda: lean(x))
    assert sum(v for k, v in launches.items() if k.startswith("conv_fwd")) >= 15, launches
    g = gold["flownet4_eval"]
    _close(_sub(f128, 2), g["flow128_s2"])
    _close(f64, g["flow64"])
    _close(f32, g["flow32"])
    assert abs(f128.double().sum().item() - g["sum128"].item()) < 5e-2


def test_flownet64_eval_forward_is_bit_reproducible_with_the_reduction_splits_on():
    """VERDICT r5 (missing 3), the part of it that is FlowNet's forward (BASELINE configs[1], batch 6): 18 of the lean path's layers are cut
    along their reduction (csrc/conv_fwd.hip).  Rounds 2-5 added the slices with float atomics -- the result changed run to run; since
    round 6 every slice stores into its own workspace slot and a fixed-order pass adds them, so two forwards of the same network on the
    same input are equal BIT FOR BIT, eagerly and replayed from the captured hipGraph, and the split launches really ran."""
    from ffwm_amd import _lib, flownet_eval, nets
    torch.manual_seed(0)
    net = nets.FlowNet(64).to(DEV).eval()
    x = torch.rand(6, 3, 128, 128, generator=torch.Generator().manual_seed(4)).to(DEV)
    lean = flownet_eval.FoldedFlowNet(net)
    _lib.prof_reset()
    _lib.prof_enable(True)
    a = [t.clone() for t in lean(x)]
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    scopes = _lib.prof_collect()
    assert scopes.get("conv_fwd_split_reduce", {}).get("launches", 0) >= 10, sorted(scopes)
    for _ in range(3):
        b = lean(x)
        assert all(torch.equal(p, q) for p, q in zip(a, b))
    graphed = flownet_eval.FoldedFlowNet(net, graph=True)
    for _ in range(3):
        c = graphed(x)
        assert all(torch.equal(p, q) for p, q in zip(a, c))


def test_spectral_norm_product_never_keeps_a_stale_winograd_transform(monkeypatch):
    """a spectral-normalised layer under no_grad hands the Winograd route a fresh plain tensor per forward (version 0,
    recycled address); its transformed weights must not be cached across an update of weight_orig."""
    import torch.nn as nn
    import torch.nn.functional as F
    from torch.nn.utils import spectral_norm
    from ffwm_amd import conv
    monkeypatch.setattr(conv, "WINOGRAD_MIN_PAIRS", 1)
    monkeypatch.setattr(conv, "WINOGRAD_MIN_TILES", 64)
    torch.manual_seed(0)
    layer = spectral_norm(nn.Conv2d(64, 64, 3, 1, 1)).to(DEV).eval()
    conv.route_conv_winograd(layer)
    assert type(layer) is conv.WinogradConv2d
    x = torch.randn(2, 64, 32, 32, device=DEV)
    with torch.no_grad():
        for it in range(3):
            y = layer(x)
            w = layer.weight.detach().clone()
            ref = F.conv2d(x.double(), w.double(), layer.bias.double(), 1, 1).float()
            assert (y - ref).abs().max().item() <= 1e-4 * (1 + ref.abs().max().item()), it
            layer.weight_orig.mul_(1.5).add_(0.01 * torch.randn_like(layer.weight_orig))      # a training update in between
    assert "_winograd_frozen" not in layer.__dict__ or not layer.__dict__["_winograd_frozen"]


def test_warp_attention_module_on_the_routed_kernels_matches_the_unrouted_module():
    """bench.py's warp + attention sub-path (base_networks.py:323-333 for all three levels, batch 8) runs its att convs on the
    trainer's kernels (conv.route_training_kernels): Winograd forward / data gradient, MFMA weight gradients, fused
    BatchNorm + LeakyReLU, the fused gate.  Same weights, same inputs, train mode.  The backward of conv -> BatchNorm(train) ->
    sigmoid gates amplifies fp32 rounding whatever kernels run, so the yardstick is the SAME module in float64: the routed module's
    distance to it may be at most 8 x the distance of the fp32 module on PyTorch-ROCm's convolutions (+ 1e-5 of the scale), for the
    outputs, the input / flow gradients and every parameter gradient -- and the hand-written kernels really ran.  (8 x: a Winograd
    F(2x2, 3x3) convolution rounds ~4 x coarser than a direct sum, measured 5 x on att0's first weight gradient, 1.5e-3 against
    3e-4 of its scale; each kernel by itself is held to 2e-5 against float64 in test_gpu_parity.py.)"""
    import copy
    from ffwm_amd import nets
    from ffwm_amd.conv import route_training_kernels
    from ffwm_amd.external_function import WarpFlipCat
    from ffwm_amd.spectral_norm import fuse_spectral_norm
    torch.manual_seed(3)
    ref = nets.WarpAttention(sn=True).to(DEV).train()
    own = copy.deepcopy(ref)
    r64 = nets.WarpAttention(sn=True, warp_flipcat=WarpFlipCat()).to(DEV).train()
    r64.load_state_dict(ref.state_dict())
    r64 = r64.double()
    fuse_spectral_norm(ref)
    counts = route_training_kernels(own)
    assert counts["winograd"] >= 9 and counts["residual"] == 3 and own.fuse_gate, counts
    g = torch.Generator().manual_seed(5)
    bs = 8
    feats = [torch.rand(bs, c, s, s, generator=g).to(DEV) for c, s in ref.LEVELS]
    flows = []
    for _, s in ref.LEVELS:
        lin = (torch.arange(s, dtype=torch.float32) + 0.5) / s * 2 - 1
        yy, xx = torch.meshgrid(lin, lin, indexing="ij")
        fl = torch.stack((xx + 0.05 * torch.sin(3 * yy), yy + 0.05 * torch.cos(2 * xx)), 0)
        flows.append(fl.unsqueeze(0).repeat(bs, 1, 1, 1).contiguous().to(DEV))
    gos = [torch.rand(bs, 2 * c, s, s, generator=g).to(DEV) for c, s in ref.LEVELS]

    def run(mod, dt=torch.float32):
        fs = [f.detach().to(dt).clone().requires_grad_(True) for f in feats]
        ws = [f.detach().to(dt).clone().requires_grad_(True) for f in flows]
        outs = mod(fs, ws)
        torch.autograd.backward(outs, [go.to(dt) for go in gos])
        res = {"out%d" % i: o.detach() for i, o in enumerate(outs)}
        res.update({"dfeat%d" % i: f.grad for i, f in enumerate(fs)})
        res.update({"dflow%d" % i: f.grad for i, f in enumerate(ws)})
        res.update({n: p.grad for n, p in mod.named_parameters() if p.grad is not None})
        return res
    r_own, launches = _launch_counts(lambda: run(own))
    assert sum(v for k, v in launches.items() if k.startswith("conv_winograd_fwd")) >= 6, launches
    assert sum(v for k, v in launches.items() if k.startswith("conv_winograd_dgrad")) >= 6, launches
    assert sum(v for k, v in launches.items() if k.startswith("conv3x3_wgrad")) >= 3, launches
    r_ref = run(ref)
    r_64 = run(r64, torch.float64)
    dist = lambda a, b: (a.double() - b).abs().max().item() / (1e-30 + b.abs().max().item())
    n = 0
    for name, want in r_64.items():
        if name not in r_own or name not in r_ref:
            continue
        e_own, e_ref = dist(r_own[name], want), dist(r_ref[name], want)
        assert e_own <= 8 * e_ref + 1e-5, (name, e_own, e_ref)
        n += 1
    assert n >= 9 + 18, n


def test_winograd_transforms_prepared_with_the_spectral_norm_equal_the_per_call_ones(monkeypatch):
    """From its second pass on a spectrally normalised network prepares the Winograd weight transforms of all its 3x3 layers in one
    launch per 24 (spectral_norm.SpectralNormGroup._prepare_winograd -> ffwm_conv3x3_winograd_weights_multi) instead of one launch in
    front of every convolution: the transforms are BIT-identical to the per-call ones, the passes agree, and the per-call transform
    launches are gone."""
    import copy
    from ffwm_amd import conv, nets
    from ffwm_amd.spectral_norm import fuse_spectral_norm
    a = fill.fill_module(nets.FFWM(sn=True)).to(DEV).train()
    b = copy.deepcopy(a)
    for net in (a, b):
        _route_all(net, monkeypatch)
        fuse_spectral_norm(net)
    img = fill.image(2, 3, 128, 128, "netG_in").to(DEV)
    flows = [fill.flow_field(2, s, s, "netG_flow%d" % s).to(DEV) for s in (32, 64, 128)]
    gos = [fill.image(2, 3, s, s, "go%d" % s).to(DEV) for s in (32, 64, 128)]

    def run(net):
        net.zero_grad(set_to_none=True)
        outs = net(img, flow=flows)
        torch.autograd.backward(list(outs), gos)
        return [o.detach() for o in outs], {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    monkeypatch.setattr(conv, "_WINO_BATCH", True)
    run(a)                                             # first pass: notes which transforms every layer uses
    (oa, ga), la = _launch_counts(lambda: run(a))      # second pass: prepared with the spectral norm
    monkeypatch.setattr(conv, "_WINO_BATCH", False)
    run(b)
    (ob, gb), lb = _launch_counts(lambda: run(b))
    assert la.get("conv_winograd_weights_multi", 0) >= 2 and la.get("conv_winograd_weights", 0) == 0, la
    assert lb.get("conv_winograd_weights_multi", 0) == 0 and lb.get("conv_winograd_weights", 0) >= 40, lb
    # two passes of a batch-2 GAN generator differ by atomics noise (split-K / split-reduction launches) amplified through ~30
    # BatchNorm layers whatever prepares the transforms: the passes agree to that noise, the TRANSFORMS are compared bit for bit below
    # (measured with the thresholds lifted as here: two identically built nets differ by up to 2e-2 in their outputs with OR without the
    #  prepared transforms -- tools: the same comparison with the switch off on both sides)
    for x, y in zip(oa, ob):
        assert (x - y).abs().max().item() <= 8e-2
    assert set(ga) == set(gb)
    from ffwm_amd import ops
    g = torch.Generator().manual_seed(4)
    for K, C, W in ((195, 195, 64), (64, 128, 32), (256, 70, 30), (3, 195, 128)):
        w = torch.randn(K, C, 3, 3, generator=g).to(DEV)
        for dg in (0, 1):
            (ws,) = ops.conv3x3_winograd_weights_multi([(w, dg, W % 4 == 0)])
            kept = {}
            Kc, Cc = (C, K) if dg else (K, C)
            ops.conv3x3_winograd(torch.randn(1, Cc, 8, W, generator=g).to(DEV), w, None, data_gradient=bool(dg), frozen=kept)
            (percall,) = kept.values()
            # the part of the workspace a call writes: whole 64-channel tiles of transformed weights, then the thin tail's plain weights
            tail = Kc % 64
            thin = (Kc > 64 or Kc <= 4) and 1 <= tail <= 4 and W % 4 == 0
            n = -(-(Kc - tail if thin else Kc) // 64) * -(-Cc // 8) * 8192 + (Cc * 36 if thin else 0)
            assert n <= ws.numel() and torch.equal(ws[:n], percall[:n]), (K, C, W, dg)


def test_reference_written_checkpoint_and_composed_test_forward_on_the_gpu(tmp_path):
    """SURVEY 8(f) rank 4 / VERDICT r4 next 8, on the product path: FFWMTrainer on the GPU (HIP warp kernels, routed convolutions,
    batched spectral norm, guided-filter kernel) loads a checkpoint directory in the reference's layout -- flowNetF from the file the
    REFERENCE's FlowNet(4) wrote (tests/golden/ckpt), netG / netD re-derived under the reference's key names and held to the
    reference's per-tensor checksums -- and its test_forward (models/ffwm_model.py:183-189) reproduces the reference's composed
    flowNetF -> WarpNet -> netG -> GuidedFilter(32) outputs (tests/golden/reference_eval.pt) to 1e-4 of their scale."""
    import test_trainer_cpu as tc
    from ffwm_amd import trainer
    gold_eval, ckpt_dir = tc._eval_golden()
    ep = tc._prepare_reference_checkpoints(tmp_path, gold_eval, ckpt_dir)
    t = trainer.FFWMTrainer(torch.device(DEV), seed=5, ngf=4)
    t.load_networks(str(tmp_path), ep, names=("flowNetF", "netG", "netD"))
    ref_sd = torch.load(os.path.join(ckpt_dir, "%s_net_flowNetF.pth" % ep))
    for k, v in t.flowNetF.state_dict().items():
        assert torch.equal(v.cpu(), ref_sd[k]), k
    for n in t.MODEL_NAMES:
        getattr(t, n).eval()
    b = {"img_S": fill.image(2, 3, 128, 128, "eval_img_S").to(DEV), "img_F": fill.image(2, 3, 128, 128, "eval_img_F").to(DEV)}
    fake, gf, warped, att = t.test_forward(b)
    tf = gold_eval["test_forward"]
    with torch.no_grad():
        flows = t.flowNetF(b["img_S"])
    worst = {}
    for got, key in zip(flows, ("flow_F128", "flow_F64", "flow_F32")):
        worst[key] = tc._packed_close(got, tf[key], 1e-4)
    worst["img_S_warp"] = tc._packed_close(warped, tf["img_S_warp"], 1e-4)
    worst["fake_F128"] = tc._packed_close(fake, tf["fake_F128"], 1e-4)
    worst["att"] = tc._packed_close(att, tf["att"], 1e-4)
    # the guided filter: on the input pair, and as the end of the composed forward (round 6: a generated image of std 0.14, the composed
    # img_GF128 held to 2e-4 and to the reference's own distance from float64 -- round 5's near-constant image only allowed a bound of 0.13)
    with torch.no_grad():
        worst["gf128_on_images"] = tc._packed_close(t.gf[128](b["img_S"], b["img_F"]), gold_eval["gf128_on_images"], 1e-4)
    worst["img_GF128_vs_fp64"] = tc._gf_close(gf, gold_eval)
    with torch.no_grad():
        score = t.netD(fake)
    ref_score = gold_eval["netD_score_of_fake"]
    assert float((score.cpu() - ref_score).abs().max()) <= 1e-4 * (1 + float(ref_score.abs().max()))
    # the identity feature of the evaluation path (ffwm_model.py:191-202) runs on the loaded generator's output
    assert t.identity_feature(fake).shape[0] == 2


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 6])
@pytest.mark.parametrize("B,C,H,W,K", [(2, 6, 40, 72, 64), (3, 34, 64, 64, 32), (1, 18, 33, 130, 16), (2, 5, 7, 9, 8), (1, 7, 16, 64, 24)])
def test_thin_channel_direct_convolution_matches_aten(B, C, H, W, K, variant):
    """Round 6: FlowNet's thin full-resolution 3 x 3 layers on the direct kernel (a pixel and 8 / 16 output channels per lane, scalar weights,
    three or one input channels per step): Conv2d(C, K, 3, 1, 1) + bias + LeakyReLU against ATen in float64, ragged widths, no bias."""
    import torch.nn.functional as F
    from ffwm_amd import flownet_eval as fe, _lib
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.2
    b = torch.randn(K, generator=g)
    ref0 = F.conv2d(x.double(), w.double(), None, 1, 1)
    ref = F.leaky_relu(ref0 + b.double().view(1, -1, 1, 1), 0.2)
    wt = fe.thin_weights(w.cuda())
    tol = 2e-6 * max(1.0, (C * 9) ** 0.5 / 4)
    _lib.set_option("conv_thin_variant", variant)
    try:
        out = fe.conv_thin(x.cuda(), wt, b.cuda(), fe.LRELU, 0.2)
        plain = fe.conv_thin(x.cuda(), wt, None, fe.NONE)
    finally:
        _lib.set_option("conv_thin_variant", 0)
    assert float((out.cpu().double() - ref).abs().max()) <= tol * float(ref.abs().max())
    assert float((plain.cpu().double() - ref0).abs().max()) <= tol * float(ref0.abs().max())
