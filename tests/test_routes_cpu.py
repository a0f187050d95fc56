"""conv.route_training_kernels on the CPU: the re-classed module keeps its parameters, buffers and state-dict keys and -- where no
kernel applies (CPU tensors) -- computes exactly what the untouched module computes, forward and backward.  (The GPU side of the same
routes: tests/test_gpu_nets_golden.py::test_warp_attention_module_on_the_routed_kernels_matches_the_unrouted_module.)"""
import copy

import torch

import torch_refs
from ffwm_amd import nets
from ffwm_amd.conv import route_training_kernels


def test_routes_keep_the_module_and_fall_back_to_the_composition_on_cpu():
    torch.manual_seed(0)
    ref = nets.WarpAttention(sn=True, warp_flipcat=torch_refs.warp_flipcat).train()      # (per-layer spectral-norm hooks: the batched group is GPU-only)
    own = copy.deepcopy(ref)
    # route_training_kernels minus its last step (fuse_spectral_norm: GPU-only)
    assert callable(route_training_kernels)
    from ffwm_amd import conv
    from ffwm_amd.norm import fuse_bn_lrelu
    from ffwm_amd.residual import fuse_residual
    n = {"mfma_wgrad": conv.route_conv_wgrad(own), "winograd": conv.route_conv_winograd(own), "mfma_fwd": conv.route_conv_fwd(own),
         "own_bwd": conv.route_conv_bwd(own), "bn_lrelu": fuse_bn_lrelu(own), "residual": fuse_residual(own)}
    assert n["winograd"] >= 9 and n["residual"] == 3 and n["own_bwd"] >= 3 and own.fuse_gate, n     # own_bwd: the 1x1 shortcuts
    assert list(own.state_dict().keys()) == list(ref.state_dict().keys())
    g = torch.Generator().manual_seed(1)
    bs = 2
    feats = [torch.rand(bs, c, s // 4, s // 4, generator=g) for c, s in ref.LEVELS]
    flows = []
    for _, s in ref.LEVELS:
        lin = (torch.arange(s // 4, dtype=torch.float32) + 0.5) / (s // 4) * 2 - 1
        yy, xx = torch.meshgrid(lin, lin, indexing="ij")
        flows.append(torch.stack((xx, yy), 0).unsqueeze(0).repeat(bs, 1, 1, 1))
    gos = [torch.rand(bs, 2 * c, s // 4, s // 4, generator=g) for c, s in ref.LEVELS]

    def run(mod):
        fs = [f.clone().requires_grad_(True) for f in feats]
        outs = mod(fs, flows)
        torch.autograd.backward(outs, gos)
        return outs, fs
    (oa, fa), (ob, fb) = run(ref), run(own)
    for a, b in zip(oa, ob):
        assert torch.allclose(a, b, rtol=0, atol=1e-6)
    for a, b in zip(fa, fb):
        assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-6)
    for (k, p), (_, q) in zip(ref.named_parameters(), own.named_parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6), k
    for (k, p), (_, q) in zip(ref.named_buffers(), own.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-6, atol=1e-7), k
