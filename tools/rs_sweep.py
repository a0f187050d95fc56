"""resample2d forward variants at [8,64,512,512] and cfg-1 (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
for shape in ((8, 64, 512, 512), (1, 64, 128, 128), (32, 64, 256, 256)):
    B, C, H, W = shape
    in1 = torch.rand(B, C, H, W, generator=g).cuda()
    for amp, name in ((3.0, "U[-3,3)"), (0.0, "smooth")):
        if amp:
            fl = torch.rand(B, 2, H, W, generator=g) * 2 * amp - amp
        else:
            yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
            fl = torch.stack((3 * torch.sin(xx / 40 + yy / 55), 3 * torch.cos(yy / 35 - xx / 60)), 0).unsqueeze(0).repeat(B, 1, 1, 1)
        in2 = torch.cat((fl, torch.full((B, 1, H, W), 2.0)), 1).cuda()
        o = torch.empty_like(in1)
        nbytes = 4.0 * B * H * W * (2 * C + 3)
        for v in (1, 2, 3, 4, 5, 6, 7, 0):
            _lib.set_option("rs_fwd_variant", v)
            us = t(lambda: ops.resample2d_forward(in1, in2, 4, 1, out=o))
            print("%-18s %-8s variant %d: %8.1f us  %6.2f TB/s" % (shape, name, v, us, nbytes / us / 1e6))
    del in1, in2, o
