"""Pins the CPU oracle (oracle/ffwm_oracle.c) before anything is compared against it.

Known answers come from the reference's own manual test scripts
(/root/reference/cuda/block_extractor/test_block_extractor.py:77-81,
 /root/reference/cuda/local_attn_reshape/test_local_attn_reshape.py:29-43,66-70) and from
the exact stock-PyTorch identities established in SURVEY.md D6 / section 4(3):
  local_attn_reshape == pixel_shuffle,
  block_extractor    == k*k shifted grid_sample(border, align_corners=True),
  block_extractor with constant flow k//2 == unfold,
  warp               == F.grid_sample(bilinear, zeros, align_corners=False).
"""
import pytest
import torch
import torch.nn.functional as F


def _extract_via_grid_sample(source, flow, k):
    """k*k shifted grid_sample(border, align_corners=True) -- SURVEY D6."""
    B, C, Hs, Ws = source.shape
    _, _, Hf, Wf = flow.shape
    out = source.new_zeros(B, C, k * Hf, k * Wf)
    ys, xs = torch.meshgrid(torch.arange(Hf, dtype=source.dtype),
                            torch.arange(Wf, dtype=source.dtype), indexing="ij")
    for i in range(k):
        for j in range(k):
            px = xs + flow[:, 0] + (j - k // 2)
            py = ys + flow[:, 1] + (i - k // 2)
            gx = 2 * px / (Ws - 1) - 1
            gy = 2 * py / (Hs - 1) - 1
            grid = torch.stack((gx, gy), -1)
            out[:, :, i::k, j::k] = F.grid_sample(source, grid, mode="bilinear",
                                                  padding_mode="border", align_corners=True)
    return out


def test_local_attn_reshape_range9_known_answer(oracle):
    # reference test_local_attn_reshape.py:29-43 expects [[0,1,2],[3,4,5],[6,7,8]] tiles
    x = torch.arange(9.).view(1, -1, 1, 1).repeat(2, 1, 10, 10).float()
    out = oracle.local_attn_reshape_forward(x.contiguous(), 3)
    assert out.shape == (2, 1, 30, 30)
    want = torch.tensor([[0., 1, 2], [3, 4, 5], [6, 7, 8]])
    assert torch.equal(out[0, 0, :3, :3], want)
    assert torch.equal(out[1, 0, 27:, 27:], want)


@pytest.mark.parametrize("k", [2, 3, 5, 7])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_local_attn_reshape_is_pixel_shuffle(oracle, k, dtype):
    g = torch.Generator().manual_seed(k)
    x = torch.rand(3, k * k, 6, 5, generator=g, dtype=dtype)
    out = oracle.local_attn_reshape_forward(x, k)
    assert torch.equal(out, F.pixel_shuffle(x, k))
    go = torch.rand(out.shape, generator=g, dtype=dtype)
    assert torch.equal(oracle.local_attn_reshape_backward(go, k), F.pixel_unshuffle(go, k))


def test_local_attn_reshape_gradcheck_reference_recipe(oracle):
    # reference test_local_attn_reshape.py:66-70: rand(4,9,14,10).double(), k=3
    x = torch.rand(4, 9, 14, 10, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda t: oracle.LocalAttnReshapeOracleFn.apply(t, 3), (x,))


@pytest.mark.parametrize("k", [2, 3, 4, 5])
def test_block_extractor_matches_shifted_grid_sample_fp64(oracle, k):
    g = torch.Generator().manual_seed(10 + k)
    src = torch.rand(2, 3, 9, 11, generator=g, dtype=torch.float64)
    flow = (torch.rand(2, 2, 9, 11, generator=g, dtype=torch.float64) - 0.5) * 8   # leaves the image
    out = oracle.block_extractor_forward(src, flow, k)
    ref = _extract_via_grid_sample(src, flow, k)
    assert (out - ref).abs().max().item() < 1e-12


def test_block_extractor_flow_smaller_than_source(oracle):
    # flow H x W may differ from the source's (block_extractor_kernel.cu:29 assert is commented out)
    g = torch.Generator().manual_seed(3)
    src = torch.rand(1, 2, 12, 12, generator=g, dtype=torch.float64)
    flow = torch.rand(1, 2, 5, 7, generator=g, dtype=torch.float64) * 3
    out = oracle.block_extractor_forward(src, flow, 3)
    assert out.shape == (1, 2, 15, 21)
    assert (out - _extract_via_grid_sample(src, flow, 3)).abs().max().item() < 1e-12


@pytest.mark.parametrize("kz,hw", [(3, 32), (5, 16), (7, 20)])
def test_block_extractor_constant_flow_is_unfold(oracle, kz, hw):
    # the reference's only real use: flow == kz//2 on an (h-kz+1)^2 grid (models/losses.py:214-216)
    g = torch.Generator().manual_seed(kz)
    grid = torch.rand(2, 1, hw, hw, generator=g) * 128
    h = hw - kz + 1
    f = torch.zeros(2, 2, h, h) + float(kz // 2)
    out = oracle.block_extractor_forward(grid, f, kz)
    unf = F.unfold(grid, kz).view(2, kz, kz, h, h).permute(0, 3, 1, 4, 2).reshape(2, 1, h * kz, h * kz)
    assert torch.equal(out, unf)


def test_block_extractor_gradcheck_reference_recipe(oracle):
    # reference test_block_extractor.py:77-81: rand(4,6,14,10).double(), flow rand*1.8, k=3
    g = torch.Generator().manual_seed(0)
    src = torch.rand(4, 6, 14, 10, generator=g, dtype=torch.float64, requires_grad=True)
    flow = (torch.rand(4, 2, 14, 10, generator=g, dtype=torch.float64) * 1.8).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda s, f: oracle.BlockExtractorOracleFn.apply(s, f, 3),
                                    (src, flow))


def test_block_extractor_backward_matches_autograd_of_identity(oracle):
    g = torch.Generator().manual_seed(5)
    src = torch.rand(2, 3, 8, 9, generator=g, dtype=torch.float64, requires_grad=True)
    flow = ((torch.rand(2, 2, 8, 9, generator=g, dtype=torch.float64) - 0.5) * 5).requires_grad_(True)
    go = torch.rand(2, 3, 24, 27, generator=g, dtype=torch.float64)
    ref = _extract_via_grid_sample(src, flow, 3)
    gs_ref, gf_ref = torch.autograd.grad(ref, (src, flow), go)
    gs, gf = oracle.block_extractor_backward(src.detach(), flow.detach(), go, 3)
    assert (gs - gs_ref).abs().max().item() < 1e-12
    assert (gf - gf_ref).abs().max().item() < 1e-11


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float64, 1e-13)])
@pytest.mark.parametrize("flipcat", [False, True])
def test_warp_matches_torch_grid_sample(oracle, dtype, tol, flipcat):
    g = torch.Generator().manual_seed(7)
    feat = torch.rand(2, 5, 12, 10, generator=g, dtype=dtype)
    flow = (torch.rand(2, 2, 9, 13, generator=g, dtype=dtype) * 2.4 - 1.2)   # partly outside [-1,1]
    out = oracle.warp_forward(feat, flow, flipcat)
    ref = F.grid_sample(feat, flow.permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros",
                        align_corners=False)
    if flipcat:
        ref = torch.cat((ref, torch.flip(ref, (3,))), 1)
    assert (out - ref).abs().max().item() < tol


@pytest.mark.parametrize("flipcat", [False, True])
def test_warp_backward_matches_torch_autograd(oracle, flipcat):
    g = torch.Generator().manual_seed(8)
    feat = torch.rand(2, 4, 8, 8, generator=g, dtype=torch.float64, requires_grad=True)
    flow = (torch.rand(2, 2, 8, 8, generator=g, dtype=torch.float64) * 2.4 - 1.2).requires_grad_(True)
    ref = F.grid_sample(feat, flow.permute(0, 2, 3, 1), mode="bilinear", padding_mode="zeros",
                        align_corners=False)
    if flipcat:
        ref = torch.cat((ref, torch.flip(ref, (3,))), 1)
    go = torch.rand(ref.shape, generator=g, dtype=torch.float64)
    gfe_ref, gfl_ref = torch.autograd.grad(ref, (feat, flow), go)
    gfe, gfl = oracle.warp_backward(feat.detach(), flow.detach(), go, flipcat)
    assert (gfe - gfe_ref).abs().max().item() < 1e-12
    assert (gfl - gfl_ref).abs().max().item() < 1e-11
