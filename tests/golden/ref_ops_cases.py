"""Seeded inputs of the reference-op golden vectors (tests/golden/reference_ops_gfx950.pt).

Shared by the generator (tests/golden/make_ref_ops_golden.py, runs the REFERENCE's own kernels from oracle/_ref on
the MI355X) and by the tests that compare the CPU oracle and the HIP path with those vectors.  CPU torch.Generator
streams are reproducible across machines; the golden file stores an input checksum per case to prove it.
"""
import torch

SAMPLE_STRIDE = 101         # big tensors are stored as flat[::101] plus float64 sum / abs-sum
CHUNK = 512                 # ... and the float64 sum of every run of 512 consecutive elements: every element is covered


def _gen(seed):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- resample2d
# (B, C, Hi, Wi, H, W, ks, dil, sigma, seed)      sigma < 0: per-pixel sigma ~ U[0.5, 2.5)
RS_CASES = {
    "ks2_module_default": (1, 8, 20, 24, 20, 24, 2, 1, 5.0, 0),
    "ks4_ffwm": (2, 5, 17, 70, 17, 70, 4, 1, 2.0, 1),            # Resample2d(4,1,sigma=2), models/losses.py:329
    "ks4_dil2_sharp": (1, 3, 16, 16, 16, 16, 4, 2, 0.3, 2),
    "ks4_grid_mismatch": (1, 4, 12, 12, 7, 9, 4, 1, 2.0, 3),
    "ks6": (1, 3, 14, 14, 14, 14, 6, 1, 2.0, 4),
    "ks8": (1, 2, 14, 14, 14, 14, 8, 1, 2.0, 5),
    "ks4_sigma0": (1, 2, 10, 10, 10, 10, 4, 1, 0.0, 6),          # SAFE_DIV's EPS arm
    "ks5_odd": (1, 9, 11, 13, 11, 13, 5, 1, 1.0, 7),
    "ks4_varsigma": (2, 6, 24, 40, 24, 40, 4, 1, -1.0, 8),
    "ks2_varsigma": (1, 7, 33, 31, 33, 31, 2, 1, -1.0, 9),
    "ks4_far_flow": (1, 3, 18, 22, 18, 22, 4, 1, 2.0, 10),       # flow x 12: most taps clamp to the border
    # BASELINE configs[0]: 1x64x128x128, flow ~ U[-3,3) px
    "cfg1_ks4_sigma2": (1, 64, 128, 128, 128, 128, 4, 1, 2.0, 100),
    "cfg1_ks2_sigma5": (1, 64, 128, 128, 128, 128, 2, 1, 5.0, 101),
    "cfg1_ks4_sigma03": (1, 64, 128, 128, 128, 128, 4, 1, 0.3, 102),
}


def rs_inputs(name, dtype):
    B, C, Hi, Wi, H, W, ks, dil, sigma, seed = RS_CASES[name]
    g = _gen(seed)
    in1 = torch.rand(B, C, Hi, Wi, generator=g, dtype=dtype)
    flow = torch.rand(B, 2, H, W, generator=g, dtype=dtype) * 6 - 3
    if name == "ks4_far_flow":
        flow = flow * 12
    if sigma < 0:
        sg = torch.rand(B, 1, H, W, generator=g, dtype=dtype) * 2 + 0.5
    else:
        sg = torch.full((B, 1, H, W), sigma, dtype=dtype)
    in2 = torch.cat((flow, sg), 1).contiguous()
    go = torch.rand(B, C, H, W, generator=g, dtype=dtype)
    return in1, in2, go, ks, dil


# ---------------------------------------------------------------- block_extractor
# (B, C, Hs, Ws, Hf, Wf, k, flow_scale, seed)       flow_scale 0: the constant flow k//2 the reference really uses
BE_CASES = {
    "gradcheck_recipe": (4, 6, 14, 10, 14, 10, 3, -1.8, 0),      # test_block_extractor.py:77-81: rand * 1.8 (>= 0)
    "ragged": (1, 5, 37, 70, 37, 70, 3, 4.0, 1),
    "leaves_image": (2, 4, 20, 33, 20, 33, 3, 64.0, 2),
    "grid_mismatch": (1, 2, 16, 16, 9, 21, 3, 3.0, 3),
    "k1": (1, 3, 12, 12, 12, 12, 1, 2.0, 4),
    "k2": (1, 3, 12, 13, 12, 13, 2, 2.0, 5),
    "k4": (1, 2, 12, 13, 12, 13, 4, 2.0, 6),
    "k5_const": (2, 1, 64, 64, 60, 60, 5, 0.0, 7),               # models/losses.py:214-216 usage
    "k7_const": (2, 1, 128, 128, 122, 122, 7, 0.0, 8),
    "k3_tile": (1, 3, 70, 150, 70, 150, 3, 2.0, 9),              # larger than one 64 x 32 tile
    "k3_big_plane": (1, 2, 160, 200, 160, 200, 3, 2.5, 10),      # plane > 128 x 128: the shared-cell tile kernels
}


def be_inputs(name, dtype):
    B, C, Hs, Ws, Hf, Wf, k, scale, seed = BE_CASES[name]
    g = _gen(seed)
    src = torch.rand(B, C, Hs, Ws, generator=g, dtype=dtype)
    if scale == 0.0:
        flow = torch.zeros(B, 2, Hf, Wf, dtype=dtype) + float(k // 2)
    elif scale < 0:
        flow = torch.rand(B, 2, Hf, Wf, generator=g, dtype=dtype) * (-scale)
    else:
        flow = (torch.rand(B, 2, Hf, Wf, generator=g, dtype=dtype) * 2 - 1) * scale
    go = torch.rand(B, C, k * Hf, k * Wf, generator=g, dtype=dtype)
    return src, flow, go, k


# ---------------------------------------------------------------- local_attn_reshape
# (B, H, W, k, seed)
LAR_CASES = {
    "recipe": (4, 14, 10, 3, 0),                                  # test_local_attn_reshape.py:66-70
    "k5": (2, 60, 60, 5, 1),
    "k7": (1, 122, 122, 7, 2),
    "k2": (1, 7, 9, 2, 3),
}


def lar_inputs(name, dtype):
    B, H, W, k, seed = LAR_CASES[name]
    g = _gen(seed)
    x = torch.rand(B, k * k, H, W, generator=g, dtype=dtype)
    go = torch.rand(B, 1, k * H, k * W, generator=g, dtype=dtype)
    return x, go, k


DTYPES = {"f32": torch.float32, "f64": torch.float64}


def pack(t):
    """What the golden file keeps of a tensor: everything when small, a strided sample + two float64 moments
    when big."""
    t = t.detach().cpu().contiguous()
    d = {"shape": tuple(t.shape), "sum": float(t.double().sum()), "abs_sum": float(t.double().abs().sum())}
    if t.numel() <= 4096:
        d["full"] = t.clone()
    else:
        d["sample"] = t.flatten()[::SAMPLE_STRIDE].clone()
        d["chunk_sums"] = chunk_sums(t)
    return d


def chunk_sums(t):
    """float64 sums of consecutive runs of CHUNK elements of the flattened tensor (the last run zero-padded)"""
    f = t.detach().cpu().flatten().double()
    pad = (-f.numel()) % CHUNK
    if pad:
        f = torch.cat((f, torch.zeros(pad, dtype=torch.float64)))
    return f.view(-1, CHUNK).sum(1)


def compare(got, packed, tol, relative=True):
    """max abs diff of `got` against a packed golden tensor (on the stored elements) and of the moments.
    Returns the max abs diff; raises AssertionError beyond tol * (1 + max|ref|)."""
    got = got.detach().cpu().contiguous()
    assert tuple(got.shape) == packed["shape"], (tuple(got.shape), packed["shape"])
    ref = packed["full"] if "full" in packed else packed["sample"]
    mine = got if "full" in packed else got.flatten()[::SAMPLE_STRIDE]
    ref = ref.to(mine.dtype)
    scale = (1 + ref.abs().max().item()) if relative else 1.0
    diff = (mine - ref).abs().max().item() if ref.numel() else 0.0
    assert diff <= tol * scale, "max abs diff %.3e > %.3e" % (diff, tol * scale)
    n = max(1, got.numel())
    dsum = abs(float(got.double().sum()) - packed["sum"]) / n
    # 1e-12: the two float64 sums were taken on different machines (GPU reduction order vs CPU)
    assert dsum <= (tol + 1e-12) * scale, "mean drift %.3e > %.3e" % (dsum, tol * scale)
    if "chunk_sums" in packed:
        # every element takes part: a run of CHUNK elements may drift by CHUNK * tol at most; independent rounding adds up to
        # ~sqrt(CHUNK) * tol, so 4 sqrt(CHUNK) tol separates a wrong region from noise
        cd = (chunk_sums(got) - packed["chunk_sums"]).abs().max().item()
        bound = 4.0 * (CHUNK ** 0.5) * (tol + 1e-12) * scale
        assert cd <= bound, "a run of %d elements differs by %.3e in sum > %.3e" % (CHUNK, cd, bound)
    return diff
