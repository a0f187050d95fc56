"""block attention backward at cfg-5 per GPU ([4,128,256,256], k 3): the backward by linearity (ba_bwd_fused 1-3: source kernel tile rows /
threads 32/256, 16/256, 32/512; key=value arguments set further options, e.g. ba_bwd_pix=1) (rounds 4-5's tile2 + weights launches, removed:
622 / 540 us random / smooth on the same box).  HIP-event time per scope, us,
cold caches; random U[-2,2) and smooth flow; the largest difference between the modes' results."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
B, C, H, W = 4, 128, 256, 256
src = torch.rand(B, C, H, W, generator=g).to(dev)
rnd = (torch.rand(B, 2, H, W, generator=g) * 4 - 2).to(dev)
lin = torch.linspace(-1, 1, H)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((2 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 2 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)), 0)[None].repeat(B, 1, 1, 1).to(dev)
wts = torch.rand(B, 9, H, W, generator=g).to(dev)
go = torch.rand(B, C, H, W, generator=g).to(dev)
flush = torch.empty(128 << 20, device=dev)
modes = [int(a) for a in sys.argv[1:] if "=" not in a] or [1, 2, 3]
for a in sys.argv[1:]:
    if "=" in a:
        _lib.set_option(a.split("=")[0], int(a.split("=")[1]))
for name, fl in (("random", rnd), ("smooth", sm)):
    ref = None
    for m in modes:
        _lib.set_option("ba_bwd_fused", m)
        gs, gf, gw = torch.zeros_like(src), torch.zeros_like(fl), torch.zeros_like(wts)
        ops.block_attention_backward(src, fl, wts, go, 3, gs, gf, gw)
        res = [t.clone() for t in (gs, gf, gw)]
        if ref is None:
            ref = res
        err = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(res, ref)]
        for _ in range(2):
            ops.block_attention_backward(src, fl, wts, go, 3, gs, gf, gw)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(6):
            flush.sum()
            ops.block_attention_backward(src, fl, wts, go, 3, gs, gf, gw)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        rows = {k.replace("block_attention_bwd_", ""): round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}
        print("%-6s mode %d %s total %.1f  rel diff to the first mode (gs, gf, gw) %s" % (name, m, rows, sum(rows.values()), ["%.1e" % e for e in err]))
_lib.set_option("ba_bwd_fused", 3)
# the forward: ba_fwd_pix_kernel (1) against rounds 3-5's be_fwd_lds_kernel<.., MODE 1> (0)
bo = torch.empty(B, C, H, W, device=dev)
for name, fl in (("random", rnd), ("smooth", sm)):
    ref = None
    for m in (0, 1, 2, 3, 4):
        _lib.set_option("ba_fwd_pix", m)
        for _ in range(2):
            ops.block_attention_forward(src, fl, wts, 3, out=bo)
        res = bo.clone()
        ref = res if ref is None else ref
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(6):
            flush.sum()
            ops.block_attention_forward(src, fl, wts, 3, out=bo)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        rows = {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}
        print("%-6s forward ba_fwd_pix %d %s  rel diff to 0: %.1e" % (name, m, rows, float((res - ref).abs().max() / ref.abs().max())))
_lib.set_option("ba_fwd_pix", 1)
