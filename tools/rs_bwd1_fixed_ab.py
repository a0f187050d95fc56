"""resample2d d_input1 at [8,64,512,512], ks = 4: the tile kernel with 32-bit fixed-point box cells (rs_bwd1_fixed=0, round 5) against
double cells (=2), alone for every call (rs_bwd1_variant=6) and inside the flow-adaptive pair of rounds 3-4 (variant 0: tap-lane kernel
for an irregular flow, tile kernel for a smooth one); random U[-3,3) and smooth flow; HIP-event time per launch, and the results of
the variants against each other."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
B, C, S = 8, 64, 512
in1 = torch.rand(B, C, S, S, generator=g).to(dev)
rnd = torch.cat((torch.rand(B, 2, S, S, generator=g) * 6 - 3, torch.full((B, 1, S, S), 2.0)), 1).to(dev)
lin = torch.linspace(-1, 1, S)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((3 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 3 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy),
                  torch.full((S, S), 2.0)), 0).unsqueeze(0).repeat(B, 1, 1, 1).contiguous().to(dev)
go = torch.rand(B, C, S, S, generator=g).to(dev)
res = {}
for variant, fixed, rpt in ((0, 2, 0), (6, 2, 0), (6, 0, 0), (6, 0, 2), (6, 2, 2)):
    _lib.set_option("rs_bwd1_variant", variant)
    _lib.set_option("rs_bwd1_fixed", fixed)
    _lib.set_option("rs_bwd1_rpt", rpt)
    for name, fl in (("random", rnd), ("smooth", sm)):
        g1 = torch.zeros_like(in1)
        for _ in range(2):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        g1.zero_()
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(3):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("variant %d cells %s rows/thread %d %s" % (variant, "fixed " if fixed == 0 else "double", rpt or 4, name), {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
        res[(variant, fixed, name)] = g1 / 3 if rpt == 0 else res.get((variant, fixed, name))
_lib.set_option("rs_bwd1_variant", 0)
_lib.set_option("rs_bwd1_fixed", 0)
_lib.set_option("rs_bwd1_rpt", 0)
for name in ("random", "smooth"):
    a, b = res[(6, 2, name)], res[(6, 0, name)]
    print(name, "fixed-point vs double cells (tile kernel): max abs diff / (1 + max|ref|) = %.3g" % float((a - b).abs().max() / (1 + a.abs().max())), flush=True)
