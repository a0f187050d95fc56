"""A/B of the block_extractor / block attention backward at cfg-5 per GPU ([4,128,256,256], k = 3): the shared-cell tile kernel with
double accumulator cells (be_bwd_fixed=2, rounds 2-4) against 32-bit fixed-point cells (be_bwd_fixed=0, round 5), random U[-2,2) and
smooth flow, HIP-event time of the launches; and the two results against each other."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
src = torch.rand(4, 128, 256, 256, generator=g).to(dev)
rnd = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(dev)
yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
sm = torch.stack([2 * torch.sin(xx / 41.0 + yy / 67.0), 2 * torch.cos(xx / 53.0 - yy / 37.0)]).unsqueeze(0).repeat(4, 1, 1, 1).to(dev)
go = torch.rand(4, 128, 768, 768, generator=g).to(dev)
wts = torch.rand(4, 9, 256, 256, generator=g).to(dev)
bo = torch.rand(4, 128, 256, 256, generator=g).to(dev)
res = {}
for variant in (2, 0):
    _lib.set_option("be_bwd_fixed", variant)
    for name, fl in (("random", rnd), ("smooth", sm)):
        gs, gf = torch.zeros_like(src), torch.zeros_like(fl)
        for _ in range(2):
            ops.block_extractor_backward(src, fl, go, 3, gs, gf)
        gs.zero_(); gf.zero_()
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            ops.block_extractor_backward(src, fl, go, 3, gs, gf)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("variant", variant, "extractor", name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
        res[("be", name, variant)] = (gs / 5, gf / 5)
        gs, gf, gw = torch.zeros_like(src), torch.zeros_like(fl), torch.zeros_like(wts)
        for _ in range(2):
            ops.block_attention_backward(src, fl, wts, bo, 3, gs, gf, gw)
        gs.zero_(); gf.zero_(); gw.zero_()
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            ops.block_attention_backward(src, fl, wts, bo, 3, gs, gf, gw)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("variant", variant, "attention", name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
        res[("ba", name, variant)] = (gs / 5, gf / 5)
for kind in ("be", "ba"):
    for name in ("random", "smooth"):
        a, b = res[(kind, name, 2)], res[(kind, name, 0)]
        print(kind, name, "fixed-point vs double cells: d(source) %.3g  d(flow) %.3g  (max abs diff / (1 + max|ref|))" % (
            float((a[0] - b[0]).abs().max() / (1 + a[0].abs().max())), float((a[1] - b[1]).abs().max() / (1 + a[1].abs().max()))), flush=True)
