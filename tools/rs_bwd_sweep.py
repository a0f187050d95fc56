import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
g = torch.Generator().manual_seed(0)
for shape in ((1, 64, 128, 128), (8, 64, 128, 128), (8, 64, 512, 512)):
    B, C, H, W = shape
    in1 = torch.rand(B, C, H, W, generator=g).cuda()
    in2 = torch.cat((torch.rand(B, 2, H, W, generator=g) * 6 - 3, torch.full((B, 1, H, W), 2.0)), 1).cuda()
    go = torch.rand(B, C, H, W, generator=g).cuda()
    g1, g2 = torch.zeros_like(in1), torch.zeros_like(in2)
    for v in (0, 2, 1):
        if v == 2 and H > 128: continue
        _lib.set_option("scatter_variant", v)
        for _ in range(2): ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5): ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        rows = _lib.prof_collect()
        print(shape, "scatter_variant", v, {k: (round(r["avg_ms"] * 1e3, 1), round(r["bytes_per_launch"] / r["avg_ms"] / 1e9, 2)) for k, r in rows.items()})
    _lib.set_option("scatter_variant", 0)
