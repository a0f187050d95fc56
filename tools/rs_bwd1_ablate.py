"""resample2d d_input1 tile kernel at [8,64,512,512], ks = 4, with parts switched off (ablate bits of rs_bwd1_tile_kernel: 1 = no LDS
atomics, 2 = no fold (global) atomics, 4 = no box clearing, 8 = no Gaussian weights) -- timing only, the results are wrong by
construction.  RS_BWD1_FIXED=0 fixed-point cells (round 5), 2 double cells."""
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
g1 = torch.zeros_like(in1)
fixed = int(os.environ.get("RS_BWD1_FIXED", "0"))
_lib.set_option("rs_bwd1_variant", 6)
_lib.set_option("rs_bwd1_fixed", fixed)
print("box cells:", "fixed-point" if fixed != 2 else "double")
for ab in (0, 1, 2, 3, 8, 11, 15):
    _lib.set_option("ablate", ab)
    for name, fl in (("random", rnd), ("smooth", sm)):
        for _ in range(2):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(3):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("ablate %2d %s" % (ab, name), {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
_lib.set_option("ablate", 0)
_lib.set_option("rs_bwd1_variant", 0)
_lib.set_option("rs_bwd1_fixed", 0)
