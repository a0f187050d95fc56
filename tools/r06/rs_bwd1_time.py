"""resample2d d_input1 at [8,64,512,512], ks 4, sigma 2: random U[-3,3) and smooth flow, cold caches, owned tiles (rs_bwd1_owned = 0) against
rounds 3-5's shared-cell tile kernel (2); `+=` mode and overwrite mode.  HIP-event time per launch scope, us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 64, 512, 512
in1 = torch.rand(B, C, H, W, generator=g).to(dev)
rnd = torch.rand(B, 2, H, W, generator=g) * 6 - 3
lin = torch.linspace(-1, 1, H)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((3 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 3 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)), 0)[None].repeat(B, 1, 1, 1)
sg = torch.full((B, 1, H, W), 2.0)
go = torch.rand(B, C, H, W, generator=g).to(dev)
flush = torch.empty(128 << 20, device=dev)
g1 = torch.zeros_like(in1)


def run(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(n):
        flush.sum()
        fn()
    torch.cuda.synchronize(); _lib.prof_enable(False)
    return {k.replace("resample2d_bwd_input1_", ""): round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}


modes = [int(a) for a in sys.argv[1:] if "=" not in a] or [0, 2]
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        _lib.set_option(k, int(v))
for name, fl in (("random", rnd), ("smooth", sm)):
    in2 = torch.cat((fl, sg), 1).contiguous().to(dev)
    for owned in modes:
        _lib.set_option("rs_bwd1_owned", owned)
        for ov in (False, True):
            print("%-6s owned=%d overwrite=%d" % (name, owned, ov), run(lambda: ops.resample2d_backward(in1, in2, go, 4, 1, g1, None, overwrite_input1=ov)))
_lib.set_option("rs_bwd1_owned", 0)
