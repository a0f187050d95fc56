"""warp + flip + cat forward at the HBM-resident shape [32,64,256,256] (1.6 GB moved), option sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
B, C, S = 32, 64, 256
feat = torch.rand(B, C, S, S, generator=g).cuda()
lin = (torch.arange(S, dtype=torch.float32) + 0.5) / S * 2 - 1
yy, xx = torch.meshgrid(lin, lin, indexing="ij")
amp = 6.0 / S
fl = torch.stack((xx + amp * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx), yy + amp * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)), 0).unsqueeze(0).repeat(B, 1, 1, 1).contiguous().cuda()
o = torch.empty(B, 2 * C, S, S, device="cuda")
nbytes = 4.0 * B * S * S * (3 * C + 2)
for opts in ({}, {"channel_slab": 8}, {"channel_slab": 32}, {"channel_slab": 64}, {"xcd_remap": 0}, {"channel_slab": 32, "xcd_remap": 0}):
    for k, v in opts.items(): _lib.set_option(k, v)
    us = t(lambda: ops.warp_forward(feat, fl, True, out=o))
    print("%-40s %8.1f us  %5.2f TB/s  %.3f" % (opts, us, nbytes / us / 1e6, nbytes / us / 1e6 / 8))
    _lib.set_option("channel_slab", 0); _lib.set_option("xcd_remap", 1)
go = torch.rand(B, 2 * C, S, S, device="cuda")
gfe, gfl = torch.zeros_like(feat), torch.zeros_like(fl)
us = t(lambda: ops.warp_backward(feat, fl, go, True, gfe, gfl), 5)
print("backward (feat + flow) %8.1f us" % us)
