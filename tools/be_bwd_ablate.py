"""block_extractor backward at cfg-5 per GPU: time of be_bwd_tile2 with parts switched off (ablate bits: 1 = no LDS atomics,
2 = no global flush atomics, 4 = no d(flow) arithmetic) -- timing only, the results are wrong by construction."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
src = torch.rand(4, 128, 256, 256, generator=g).to(dev)
rnd = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(dev)
lin = torch.linspace(-1, 1, 256)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((2 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 2 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)), 0).unsqueeze(0).repeat(4, 1, 1, 1).contiguous().to(dev)
go = torch.rand(4, 128, 768, 768, generator=g).to(dev)
gs, gf = torch.zeros_like(src), torch.zeros_like(rnd)
lib = _lib.load()
fixed = int(os.environ.get("BE_BWD_FIXED", "0"))          # 0 = 32-bit fixed-point cells (round 5), 2 = double cells
lib.ffwm_set_option(b"be_bwd_fixed", fixed)
print("accumulator cells:", "fixed-point" if fixed != 2 else "double", "(ablate 8 = the ablation instantiation with nothing switched off)")
for ab in [int(a) for a in (sys.argv[1:] or ["0", "8", "1", "2", "4", "3", "7"])]:
    lib.ffwm_set_option(b"ablate", ab)
    for name, fl in (("random", rnd), ("smooth", sm)):
        for _ in range(2):
            ops.block_extractor_backward(src, fl, go, 3, gs, gf)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            ops.block_extractor_backward(src, fl, go, 3, gs, gf)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("ablate", ab, name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
lib.ffwm_set_option(b"ablate", 0)
