"""block_extractor / block attention backward at cfg-5 per GPU ([4,128,256,256], k = 3), random and smooth flow, cold caches (a 512 MiB
read between the launches): HIP-event time per launch scope, us.  For same-box A/Bs of library builds (tools/ab/ab.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
src = torch.rand(4, 128, 256, 256, generator=g).to(dev)
rnd = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(dev)
lin = torch.linspace(-1, 1, 256)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((2 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 2 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)), 0).unsqueeze(0).repeat(4, 1, 1, 1).contiguous().to(dev)
go = torch.rand(4, 128, 768, 768, generator=g).to(dev)
gatt = torch.rand(4, 128, 256, 256, generator=g).to(dev)
wts = torch.rand(4, 9, 256, 256, generator=g).to(dev)
flush = torch.empty(128 << 20, device=dev)
gs, gf, gw = torch.zeros_like(src), torch.zeros_like(rnd), torch.zeros_like(wts)


def run(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(n):
        flush.sum()
        fn()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    return {k.replace("block_extractor_", "").replace("block_attention_", ""): round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}


for name, fl in (("random", rnd), ("smooth", sm)):
    print("%-6s extractor" % name, run(lambda: ops.block_extractor_backward(src, fl, go, 3, gs, gf)))
    print("%-6s attention" % name, run(lambda: ops.block_attention_backward(src, fl, wts, gatt, 3, gs, gf, gw)))
