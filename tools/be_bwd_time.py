"""block_extractor backward at cfg-5 per GPU ([4,128,256,256], k = 3), random and smooth flow: HIP-event time of the launches."""
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
out = ops.block_extractor_forward(src, rnd, 3)
go = torch.rand_like(out)
gs, gf = torch.zeros_like(src), torch.zeros_like(rnd)
for name, fl in (("random", rnd), ("smooth", sm)):
    for _ in range(2):
        ops.block_extractor_backward(src, fl, go, 3, gs, gf)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(5):
        ops.block_extractor_backward(src, fl, go, 3, gs, gf)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    print(name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
