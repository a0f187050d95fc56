"""Host-side logic of the in-place module rewrites (no GPU): conv weight-gradient routing, BatchNorm + LeakyReLU
fusion, the reducer's flat gradient layout.  On the CPU every rewritten module must behave exactly like the module it
replaced (the kernels are GPU-only; the fallbacks are the original PyTorch ops, never the oracle)."""
import copy

import torch
import torch.nn as nn

from ffwm_amd import nets
from ffwm_amd.conv import MfmaWgradConv2d, route_conv_wgrad, wgrad_route_ok
from ffwm_amd.dp import BucketedGradReducer
from ffwm_amd.norm import BatchNormLeakyReLU2d, fuse_bn_lrelu


def test_route_conv_wgrad_picks_the_3x3_layers_and_keeps_the_state_dict():
    torch.manual_seed(0)
    net = nets.FFWM(sn=True)
    keys = list(net.state_dict().keys())
    n = route_conv_wgrad(net)
    assert n == sum(isinstance(m, MfmaWgradConv2d) for m in net.modules()) and n >= 32
    assert list(net.state_dict().keys()) == keys
    for m in net.modules():
        if isinstance(m, MfmaWgradConv2d):
            assert m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1)
            assert min(m.in_channels, m.out_channels) >= 64 or (min(m.in_channels, m.out_channels) <= 3 <= 64 <= max(m.in_channels, m.out_channels))
    # CPU tensors never take the kernel
    assert not wgrad_route_ok(torch.zeros(1, 64, 8, 64), torch.zeros(64, 64, 3, 3))


def test_routed_conv_is_a_plain_conv_on_the_cpu():
    torch.manual_seed(1)
    conv = nn.Conv2d(64, 70, 3, 1, 1)
    ref = copy.deepcopy(conv)
    holder = nn.Sequential(conv)
    assert route_conv_wgrad(holder) == 1
    x = torch.randn(2, 64, 6, 64)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    holder(xa).square().sum().backward()
    ref(xb).square().sum().backward()
    assert torch.equal(xa.grad, xb.grad) and torch.equal(conv.weight.grad, ref.weight.grad) and torch.equal(conv.bias.grad, ref.bias.grad)


def test_fuse_bn_lrelu_structure_and_cpu_equivalence():
    torch.manual_seed(2)
    net = nets.FlowNet(8)
    ref = copy.deepcopy(net)
    keys = list(net.state_dict().keys())
    n = fuse_bn_lrelu(net)
    assert n > 10 and n == sum(isinstance(m, BatchNormLeakyReLU2d) for m in net.modules())
    assert list(net.state_dict().keys()) == keys
    assert not any(isinstance(m, nn.LeakyReLU) for s in net.modules() if isinstance(s, nn.Sequential)
                   for a, m in zip(list(s.children()), list(s.children())[1:]) if isinstance(a, BatchNormLeakyReLU2d))
    x = torch.rand(2, 3, 128, 128)
    for mode in (True, False):
        net.train(mode), ref.train(mode)
        for a, b in zip(net(x), ref(x)):
            assert torch.equal(a, b)
    sa, sb = net.state_dict(), ref.state_dict()
    for k in keys:
        assert torch.equal(sa[k], sb[k]), k          # running statistics and batch counters advanced identically


def test_reducer_lays_every_parameter_on_a_16_byte_boundary_of_one_flat_array():
    net = nn.Sequential(nn.Linear(5, 3), nn.Linear(3, 7), nn.Linear(7, 2))
    red = BucketedGradReducer(net.parameters(), bucket_bytes=64)
    assert red.flat is not None and len(red.buckets) > 1
    spans = sorted(red.offset[p] for p in net.parameters())
    assert all(a % 4 == 0 for a, _ in spans) and all(b0 <= a1 for (_, b0), (a1, _) in zip(spans, spans[1:]))
    for p in net.parameters():
        a, b = red.offset[p]
        assert p.grad.data_ptr() == red.flat.data_ptr() + 4 * a and b - a == p.numel()
    red.zero_grad()
    net(torch.ones(4, 5)).sum().backward()
    red.finish()
    assert all(float(red.flat[a:b].abs().sum()) > 0 for a, b in spans[:1])
    assert torch.equal(net[0].weight.grad, red.flat[red.offset[net[0].weight][0]:red.offset[net[0].weight][1]].view(3, 5))
