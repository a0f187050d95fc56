"""resample2d and warp forward / backward with a random per-pixel flow and with a smooth one of the same amplitude (extrema next
to integers): looks for per-pixel slow paths that a random field never exercises."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
dev = "cuda"
g = torch.Generator().manual_seed(0)
def fields(B, H, W, amp):
    yy, xx = torch.meshgrid(torch.arange(float(H)), torch.arange(float(W)), indexing="ij")
    smooth = torch.stack([amp * torch.sin(xx / 41.0 + yy / 67.0), amp * torch.cos(xx / 53.0 - yy / 37.0)]).unsqueeze(0).repeat(B, 1, 1, 1)
    rnd = torch.rand(B, 2, H, W, generator=g) * 2 * amp - amp
    return (("random", rnd), ("smooth", smooth))
def report(tag):
    for k, v in sorted(_lib.prof_collect().items()):
        print("%-34s %-34s %8.1f us  %5.2f TB/s" % (tag, k, v["avg_ms"] * 1e3, v["bytes_per_launch"] / v["avg_ms"] / 1e9))
def timed(fn, tag, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(n): fn()
    torch.cuda.synchronize(); _lib.prof_enable(False); report(tag)
for (B, C, H) in ((8, 64, 512), (1, 64, 128)):
    in1 = torch.rand(B, C, H, H, generator=g).to(dev)
    go = torch.rand(B, C, H, H, generator=g).to(dev)
    for name, f in fields(B, H, H, 3.0):
        in2 = torch.cat((f, torch.full((B, 1, H, H), 2.0)), 1).to(dev)
        o = torch.empty_like(in1); g1, g2 = torch.zeros_like(in1), torch.empty_like(in2)
        timed(lambda: (ops.resample2d_forward(in1, in2, 4, 1, out=o), ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2)), "resample2d [%d,%d,%d,%d] %s" % (B, C, H, H, name))
for (B, C, H) in ((8, 64, 128), (8, 64, 64)):
    feat = torch.rand(B, C, H, H, generator=g).to(dev)
    go = torch.rand(B, 2 * C, H, H, generator=g).to(dev)
    for name, f in fields(B, H, H, 3.0):
        # WarpNet takes the ABSOLUTE sampling grid in [-1, 1] (base_networks.py:168-173): identity grid (align_corners=False) + offset
        ys, xs = torch.meshgrid((torch.arange(float(H)) + 0.5) / H * 2 - 1, (torch.arange(float(H)) + 0.5) / H * 2 - 1, indexing="ij")
        fl = (torch.stack([xs, ys]).unsqueeze(0) + f / (H / 2)).contiguous().to(dev)
        out = torch.empty(B, 2 * C, H, H, device=dev); gfe, gfl = torch.zeros_like(feat), torch.zeros_like(fl)
        timed(lambda: (ops.warp_forward(feat, fl, True, out=out), ops.warp_backward(feat, fl, go, True, gfe, gfl)), "warp+flip+cat [%d,%d,%d,%d] %s" % (B, C, H, H, name))
