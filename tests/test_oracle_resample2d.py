"""resample2d has NO test in the reference and its call path is dead code
(SURVEY D4): parity is UNPINNED by the reference.  What pins the oracle's restatement of
/root/reference/cuda/resample2d_package/resample2d_kernel.cu:21-330 here is
 (1) hand-derived known answers, and
 (2) a second, independent, vectorised torch restatement of Appendix A.1 whose autograd
     must reproduce the oracle's analytic backward (A.2) -- d_input1 only without the
     reference's int() truncation quirk or for non-negative sample coordinates.
"""
import pytest
import torch


def resample2d_torch(in1, in2, ks=2, dil=1):
    """Vectorised restatement of resample2d_kernel.cu:47-93 (floor treated as constant)."""
    B, _, H, W = in2.shape
    _, C, Hi, Wi = in1.shape
    dt = in1.dtype
    dx, dy, sg = in2[:, 0], in2[:, 1], in2[:, 2]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    xf, yf = xs + dx, ys + dy
    flx, fly = torch.floor(xf).detach(), torch.floor(yf).detach()
    al, be = xf - flx, yf - fly

    def P(v):
        den = 2 * sg * sg
        return torch.exp(-(v * v) / den)

    bidx = torch.arange(B).view(B, 1, 1, 1)
    cidx = torch.arange(C).view(1, C, 1, 1)
    val = torch.zeros(B, C, H, W, dtype=dt)
    tot = torch.zeros(B, H, W, dtype=dt)
    for fy in range(ks // 2):
        yT = (fly - fy * dil).clamp(0, Hi - 1).long()
        yB = (fly + (fy + 1) * dil).clamp(0, Hi - 1).long()
        for fx in range(ks // 2):
            xL = (flx - fx * dil).clamp(0, Wi - 1).long()
            xR = (flx + (fx + 1) * dil).clamp(0, Wi - 1).long()
            xLP, xRP = P(fx * dil + al), P((1 + fx) * dil - al)
            yTP, yBP = P(fy * dil + be), P((1 + fy) * dil - be)
            for yy, yp in ((yT, yTP), (yB, yBP)):
                for xx, xp in ((xL, xLP), (xR, xRP)):
                    w = (yp * xp).unsqueeze(1)
                    val = val + w * in1[bidx, cidx, yy.unsqueeze(1), xx.unsqueeze(1)]
                    tot = tot + yp * xp
    return val / tot.unsqueeze(1)


def test_tiny_sigma_zero_flow_is_identity(oracle):
    g = torch.Generator().manual_seed(0)
    in1 = torch.rand(1, 3, 6, 7, generator=g)
    in2 = torch.zeros(1, 3, 6, 7)
    in2[:, 2] = 0.05      # weights: exp(0)=1 on the top-left tap, exp(-200)~0 elsewhere
    out = oracle.resample2d_forward(in1, in2, 2, 1)
    assert torch.allclose(out, in1, atol=1e-7)


def test_huge_sigma_is_four_neighbour_mean(oracle):
    in1 = torch.arange(30.).view(1, 1, 5, 6)
    in2 = torch.zeros(1, 3, 5, 6)
    in2[:, 0] = 0.5
    in2[:, 1] = 0.5
    in2[:, 2] = 1e4
    out = oracle.resample2d_forward(in1, in2, 2, 1)
    # interior: mean of the 2x2 block starting at (y, x)
    want = (in1[0, 0, :-1, :-1] + in1[0, 0, :-1, 1:] + in1[0, 0, 1:, :-1] + in1[0, 0, 1:, 1:]) / 4
    assert torch.allclose(out[0, 0, :-1, :-1], want, atol=1e-4)


def test_sigma_zero_is_finite(oracle):
    # SAFE_DIV's EPS branch (resample2d_kernel.cu:15): sigma == 0 must not produce NaN/inf
    g = torch.Generator().manual_seed(1)
    in1 = torch.rand(1, 2, 5, 5, generator=g)
    in2 = torch.rand(1, 3, 5, 5, generator=g)
    in2[:, 2] = 0
    in2[0, 0, 0, 0] = 0.0
    in2[0, 1, 0, 0] = 0.0      # exact-zero distance -> weight 1 on that tap
    out = oracle.resample2d_forward(in1, in2, 4, 1)
    assert torch.isfinite(out).all()
    assert out[0, 0, 0, 0].item() == pytest.approx(in1[0, 0, 0, 0].item(), abs=1e-7)
    assert out[0, 0, 2, 2].item() == 0.0      # sum == 0 -> val / EPS with val == 0


@pytest.mark.parametrize("ks,dil", [(2, 1), (4, 1), (4, 2), (6, 1)])
@pytest.mark.parametrize("sigma", [0.3, 2.0, 5.0])
def test_forward_matches_torch_restatement_fp64(oracle, ks, dil, sigma):
    g = torch.Generator().manual_seed(ks * 10 + dil)
    in1 = torch.rand(2, 5, 9, 8, generator=g, dtype=torch.float64)
    flow = torch.rand(2, 2, 9, 8, generator=g, dtype=torch.float64) * 6 - 3
    in2 = torch.cat((flow, torch.full((2, 1, 9, 8), sigma, dtype=torch.float64)), 1)
    out = oracle.resample2d_forward(in1, in2, ks, dil)
    ref = resample2d_torch(in1, in2, ks, dil)
    assert (out - ref).abs().max().item() < 1e-13


def test_forward_output_shape_follows_input2(oracle):
    # out = [B_in2, C_in1, H_in2, W_in2] (models/external_function.py:122-124)
    g = torch.Generator().manual_seed(2)
    in1 = torch.rand(2, 4, 12, 10, generator=g, dtype=torch.float64)
    in2 = torch.rand(2, 3, 5, 7, generator=g, dtype=torch.float64) + 0.5
    out = oracle.resample2d_forward(in1, in2, 4, 1)
    assert out.shape == (2, 4, 5, 7)
    assert (out - resample2d_torch(in1, in2, 4, 1)).abs().max().item() < 1e-13


@pytest.mark.parametrize("ks,dil", [(2, 1), (4, 1), (4, 2)])
def test_backward_matches_autograd_of_forward(oracle, ks, dil):
    g = torch.Generator().manual_seed(ks + dil)
    in1 = torch.rand(2, 4, 8, 9, generator=g, dtype=torch.float64, requires_grad=True)
    flow = torch.rand(2, 2, 8, 9, generator=g, dtype=torch.float64) * 6 - 3
    sig = torch.rand(2, 1, 8, 9, generator=g, dtype=torch.float64) * 2 + 0.5
    in2 = torch.cat((flow, sig), 1).requires_grad_(True)
    go = torch.rand(2, 4, 8, 9, generator=g, dtype=torch.float64)
    ref = resample2d_torch(in1, in2, ks, dil)
    g1_ref, g2_ref = torch.autograd.grad(ref, (in1, in2), go)
    g1, g2 = oracle.resample2d_backward(in1.detach(), in2.detach(), go, ks, dil,
                                        reference_quirk=False)
    assert (g1 - g1_ref).abs().max().item() < 1e-12
    assert (g2 - g2_ref).abs().max().item() < 1e-11


def test_backward_input1_quirk_only_bites_negative_coordinates(oracle):
    """alpha = xf - int(xf) (resample2d_kernel.cu:137-138) equals the floor form for xf,yf >= 0
    and differs for negative sample coordinates when ks >= 4."""
    g = torch.Generator().manual_seed(4)
    in1 = torch.rand(1, 2, 8, 8, generator=g, dtype=torch.float64)
    go = torch.rand(1, 2, 8, 8, generator=g, dtype=torch.float64)
    pos = torch.cat((torch.rand(1, 2, 8, 8, generator=g, dtype=torch.float64) * 3,
                     torch.full((1, 1, 8, 8), 2.0, dtype=torch.float64)), 1)
    a, _ = oracle.resample2d_backward(in1, pos, go, 4, 1, reference_quirk=True)
    b, _ = oracle.resample2d_backward(in1, pos, go, 4, 1, reference_quirk=False)
    assert torch.equal(a, b)
    neg = pos.clone()
    neg[:, :2] -= 3.0
    a, _ = oracle.resample2d_backward(in1, neg, go, 4, 1, reference_quirk=True)
    b, _ = oracle.resample2d_backward(in1, neg, go, 4, 1, reference_quirk=False)
    assert (a - b).abs().max().item() > 1e-3
    # ... and is invisible for ks == 2 because the clamped taps collapse onto one pixel
    a, _ = oracle.resample2d_backward(in1, neg, go, 2, 1, reference_quirk=True)
    b, _ = oracle.resample2d_backward(in1, neg, go, 2, 1, reference_quirk=False)
    far = neg.clone()
    far[:, :2] = -20.0
    a, _ = oracle.resample2d_backward(in1, far, go, 2, 1, reference_quirk=True)
    b, _ = oracle.resample2d_backward(in1, far, go, 2, 1, reference_quirk=False)
    assert (a - b).abs().max().item() < 1e-12


def test_fp32_close_to_fp64(oracle):
    g = torch.Generator().manual_seed(6)
    in1 = torch.rand(1, 8, 16, 16, generator=g)
    in2 = torch.cat((torch.rand(1, 2, 16, 16, generator=g) * 6 - 3, torch.full((1, 1, 16, 16), 2.0)), 1)
    go = torch.rand(1, 8, 16, 16, generator=g)
    o32 = oracle.resample2d_forward(in1, in2, 4, 1)
    o64 = oracle.resample2d_forward(in1.double(), in2.double(), 4, 1)
    assert (o32.double() - o64).abs().max().item() < 1e-5
    g1, g2 = oracle.resample2d_backward(in1, in2, go, 4, 1)
    h1, h2 = oracle.resample2d_backward(in1.double(), in2.double(), go.double(), 4, 1)
    assert (g1.double() - h1).abs().max().item() < 1e-5
    assert (g2.double() - h2).abs().max().item() < 1e-4
