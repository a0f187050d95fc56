"""Out-of-bounds probe of the hot-path ops through the caller-allocates C ABI: every input sits between two blocks of NaN, every
output / gradient buffer between two blocks of a sentinel value.  A kernel that READS outside an input poisons its result (compared with
the same call on plain tensors), one that WRITES outside an output disturbs a sentinel.  Ragged sizes, flows that leave the image,
planes on both sides of the plane-kernel / tile-kernel switch."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = 4096                     # guard elements on each side (16 KiB: keeps the payload 16-byte aligned)
SENTINEL = 1234.5


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def guarded_input(t):
    flat = torch.full((t.numel() + 2 * G,), float("nan"), device=DEV, dtype=t.dtype)
    flat[G:G + t.numel()] = t.to(DEV).flatten()
    return flat[G:G + t.numel()].view(t.shape)


class Out:
    def __init__(self, shape, dtype=torch.float32, fill=0.0):
        n = 1
        for s in shape:
            n *= s
        self.flat = torch.full((n + 2 * G,), SENTINEL, device=DEV, dtype=dtype)
        self.flat[G:G + n] = fill
        self.t = self.flat[G:G + n].view(shape)
        self.n = n

    def intact(self):
        return bool((self.flat[:G] == SENTINEL).all()) and bool((self.flat[G + self.n:] == SENTINEL).all())


def same(a, b, tol=0.0):
    assert torch.isfinite(a).all()
    assert (a - b).abs().max().item() <= tol * (1 + b.abs().max().item())


@pytest.mark.parametrize("case", [(2, 5, 13, 17, 3, 2.0), (1, 3, 37, 21, 3, 40.0), (1, 2, 160, 200, 3, 2.0), (2, 4, 9, 9, 2, 1.5), (1, 2, 140, 136, 3, 70.0)])
def test_block_extractor_stays_inside_its_tensors(case):
    from ffwm_amd import ops
    B, C, H, W, k, amp = case
    g = _gen(sum(case[:5]))
    src, flow = torch.rand(B, C, H, W, generator=g), (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * amp
    go = torch.rand(B, C, k * H, k * W, generator=g)
    ref = ops.block_extractor_forward(src.to(DEV), flow.to(DEV), k)
    gs_ref, gf_ref = torch.zeros(B, C, H, W, device=DEV), torch.zeros(B, 2, H, W, device=DEV)
    ops.block_extractor_backward(src.to(DEV), flow.to(DEV), go.to(DEV), k, gs_ref, gf_ref)
    s, f, gd = guarded_input(src), guarded_input(flow), guarded_input(go)
    out, gs, gf = Out((B, C, k * H, k * W)), Out((B, C, H, W)), Out((B, 2, H, W))
    ops.block_extractor_forward(s, f, k, out=out.t)
    ops.block_extractor_backward(s, f, gd, k, gs.t, gf.t)
    assert out.intact() and gs.intact() and gf.intact()
    same(out.t, ref)
    same(gs.t, gs_ref, 1e-5)
    same(gf.t, gf_ref, 1e-4)


@pytest.mark.parametrize("case", [(2, 3, 7, 11), (1, 3, 30, 30), (3, 5, 122, 61)])
def test_local_attn_reshape_stays_inside_its_tensors(case):
    from ffwm_amd import ops
    B, k, H, W = case
    x = torch.rand(B, k * k, H, W, generator=_gen(sum(case)))
    ref = ops.local_attn_reshape_forward(x.to(DEV), k)
    out, gi = Out((B, 1, k * H, k * W)), Out((B, k * k, H, W))
    ops.local_attn_reshape_forward(guarded_input(x), k, out=out.t)
    ops.local_attn_reshape_backward(guarded_input(ref.cpu()), k, grad_inputs=gi.t)
    assert out.intact() and gi.intact()
    assert torch.equal(out.t, ref) and torch.equal(gi.t, x.to(DEV))


@pytest.mark.parametrize("case", [(1, 5, 19, 23, 4, 3.0), (2, 3, 33, 17, 2, 6.0), (1, 4, 150, 140, 4, 3.0), (1, 64, 128, 128, 4, 3.0), (1, 2, 31, 29, 6, 40.0)])
def test_resample2d_stays_inside_its_tensors(case):
    from ffwm_amd import ops
    B, C, H, W, ks, amp = case
    g = _gen(sum(case[:5]))
    in1 = torch.rand(B, C, H, W, generator=g)
    in2 = torch.cat(((torch.rand(B, 2, H, W, generator=g) * 2 - 1) * amp, torch.full((B, 1, H, W), 2.0)), 1)
    go = torch.rand(B, C, H, W, generator=g)
    ref = ops.resample2d_forward(in1.to(DEV), in2.to(DEV), ks, 1)
    g1_ref, g2_ref = torch.zeros(B, C, H, W, device=DEV), torch.zeros(B, 3, H, W, device=DEV)
    ops.resample2d_backward(in1.to(DEV), in2.to(DEV), go.to(DEV), ks, 1, g1_ref, g2_ref)
    a, b, gd = guarded_input(in1), guarded_input(in2), guarded_input(go)
    out, g1, g2 = Out((B, C, H, W)), Out((B, C, H, W)), Out((B, 3, H, W))
    ops.resample2d_forward(a, b, ks, 1, out=out.t)
    ops.resample2d_backward(a, b, gd, ks, 1, g1.t, g2.t)
    assert out.intact() and g1.intact() and g2.intact()
    same(out.t, ref)
    same(g1.t, g1_ref, 1e-5)
    same(g2.t, g2_ref, 1e-4)


@pytest.mark.parametrize("flip", [True, False])
@pytest.mark.parametrize("case", [(2, 5, 13, 17, 1.4), (1, 64, 64, 64, 1.1), (1, 3, 150, 130, 1.3), (2, 8, 32, 32, 3.0), (1, 4, 260, 264, 1.05)])
def test_warp_stays_inside_its_tensors(case, flip):
    from ffwm_amd import ops
    B, C, H, W, amp = case
    g = _gen(int(sum(case[:4])))
    feat = torch.rand(B, C, H, W, generator=g)
    flow = (torch.rand(B, 2, H, W, generator=g) * 2 - 1) * amp            # normalised coordinates: |.| > 1 leaves the image
    Co = 2 * C if flip else C
    go = torch.rand(B, Co, H, W, generator=g)
    ref = ops.warp_forward(feat.to(DEV), flow.to(DEV), flip)
    gf_ref, gl_ref = torch.zeros(B, C, H, W, device=DEV), torch.zeros(B, 2, H, W, device=DEV)
    ops.warp_backward(feat.to(DEV), flow.to(DEV), go.to(DEV), flip, gf_ref, gl_ref)
    a, b, gd = guarded_input(feat), guarded_input(flow), guarded_input(go)
    out, gf, gl = Out((B, Co, H, W)), Out((B, C, H, W)), Out((B, 2, H, W))
    ops.warp_forward(a, b, flip, out=out.t)
    ops.warp_backward(a, b, gd, flip, gf.t, gl.t)
    assert out.intact() and gf.intact() and gl.intact()
    same(out.t, ref)
    same(gf.t, gf_ref, 1e-5)
    same(gl.t, gl_ref, 1e-4)


@pytest.mark.parametrize("case", [(2, 19, 7, 9, 70), (2, 195, 64, 64, 195), (8, 256, 32, 32, 256), (1, 66, 8, 12, 65), (2, 70, 32, 32, 3), (3, 33, 17, 30, 130)])
def test_winograd_convolution_stays_inside_its_tensors(case):
    from ffwm_amd import ops
    B, C, H, W, K = case
    g = _gen(sum(case))
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(K, generator=g)
    go = torch.randn(B, K, H, W, generator=g)
    ref = ops.conv3x3_winograd(x.to(DEV), w.to(DEV), b.to(DEV))
    dref = ops.conv3x3_winograd(go.to(DEV), w.to(DEV), None, data_gradient=True)
    out, dx = Out((B, K, H, W), fill=7.0), Out((B, C, H, W), fill=7.0)
    ops.conv3x3_winograd(guarded_input(x), guarded_input(w), guarded_input(b), out=out.t)
    ops.conv3x3_winograd(guarded_input(go), guarded_input(w), None, data_gradient=True, out=dx.t)
    assert out.intact() and dx.intact()
    same(out.t, ref, 1e-5)
    same(dx.t, dref, 1e-5)


@pytest.mark.parametrize("case", [(2, 70, 10, 12, 130, 3, 1, 1), (2, 64, 32, 32, 128, 4, 2, 1), (8, 195, 16, 16, 195, 1, 1, 0), (3, 18, 20, 24, 16, 3, 1, 1),
                                  (2, 130, 8, 8, 66, 3, 2, 1)])
def test_tiled_weight_gradient_reads_nothing_outside_its_operands(case):
    from ffwm_amd import ops
    B, C, H, W, K, k, stride, pad = case
    g = _gen(sum(case))
    x = torch.randn(B, C, H, W, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    go = torch.randn(B, K, Ho, Wo, generator=g)
    dw_ref, db_ref = ops.conv2d_wgrad_tiled(go.to(DEV), x.to(DEV), k, stride, pad, want_bias=True)
    dw, db = ops.conv2d_wgrad_tiled(guarded_input(go), guarded_input(x), k, stride, pad, want_bias=True)
    same(dw, dw_ref, 1e-5)
    same(db, db_ref, 1e-5)


@pytest.mark.parametrize("case", [(2, 67, 64, 128, 131), (1, 64, 64, 64, 64), (2, 195, 64, 64, 195), (2, 3, 128, 128, 64)])
def test_3x3_weight_gradients_stay_inside_their_tensors(case):
    from ffwm_amd import _lib, ops
    B, C, H, W, K = case
    g = _gen(sum(case))
    x, go = torch.randn(B, C, H, W, generator=g), torch.randn(B, K, H, W, generator=g)
    if not ops.conv3x3_wgrad_supported(x.to(DEV), go.to(DEV)):
        pytest.skip("shape not served by conv_wgrad.hip")
    for wino in (0, 2):                          # auto (Winograd-domain kernel where it applies) / direct kernel
        _lib.set_option("conv_wgrad_wino", wino)
        try:
            dw_ref, db_ref = torch.zeros(K, C, 3, 3, device=DEV), torch.zeros(K, device=DEV)
            ops.conv3x3_wgrad(x.to(DEV), go.to(DEV), dw_ref, db_ref)
            dw, db = Out((K, C, 3, 3)), Out((K,))
            ops.conv3x3_wgrad(guarded_input(x), guarded_input(go), dw.t, db.t)
            assert dw.intact() and db.intact()
            same(dw.t, dw_ref, 1e-4)
            same(db.t, db_ref, 1e-4)
        finally:
            _lib.set_option("conv_wgrad_wino", 0)


@pytest.mark.parametrize("case", [(8, 195, 64, 64, "lrelu"), (3, 70, 9, 11, "sigmoid"), (4, 600, 4, 4, "lrelu"), (2, 64, 128, 128, "lrelu")])
def test_residual_tail_kernel_reads_nothing_outside_its_operands(case):
    import torch.nn as nn
    from ffwm_amd import norm
    B, C, H, W, act = case
    g = _gen(sum(case[:4]))
    x, res, go = (torch.randn(B, C, H, W, generator=g) for _ in range(3))
    rb = torch.randn(C, generator=g)
    outs = []
    for guard in (False, True):
        put = guarded_input if guard else (lambda t: t.to(DEV))
        bn = nn.BatchNorm2d(C).to(DEV).train()
        xs, rs = put(x).requires_grad_(True), put(res).requires_grad_(True)
        y = norm.bn_res_act(xs, bn, rs, put(rb), norm.RES_ACTS[act], 0.2)
        y.backward(put(go))
        outs.append((y.detach(), xs.grad, rs.grad, bn.weight.grad, bn.bias.grad, bn.running_var.clone()))
    for a, b in zip(outs[1], outs[0]):
        same(a, b, 1e-5)
