import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 64, 512, 512
in1 = torch.rand(B, C, H, W, generator=g).to(dev)
go = torch.rand(B, C, H, W, generator=g).to(dev)
lin = torch.linspace(-1, 1, 512)
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
sm = torch.stack((3 * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), 3 * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy), torch.full((512, 512), 2.0)), 0).unsqueeze(0).repeat(8, 1, 1, 1).contiguous().to(dev)
rnd = torch.cat((torch.rand(B, 2, H, W, generator=g) * 6 - 3, torch.full((B, 1, H, W), 2.0)), 1).to(dev)
g1 = torch.zeros_like(in1)
for name, fl in (("smooth", sm), ("random", rnd)):
    for variant in (0, 2):
        lib.ffwm_set_option(b"rs_bwd1_variant", variant)
        for _ in range(2):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(4):
            ops.resample2d_backward(in1, fl, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print(name, "variant", variant, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
lib.ffwm_set_option(b"rs_bwd1_variant", 0)
