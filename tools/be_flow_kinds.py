"""block_extractor forward / backward at the cfg-5 per-GPU shape with a random flow (U[-2,2) per pixel: the bench's) and a smooth one
(low-frequency, amplitude 2 px: what a trained flow net produces): the LDS-atomic scatter of the backward is conflict-bound
only in the random case."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
dev = "cuda"
g = torch.Generator().manual_seed(0)
src = torch.rand(4, 128, 256, 256, generator=g).to(dev)
out = torch.empty(4, 128, 768, 768, device=dev)
yy, xx = torch.meshgrid(torch.arange(256.0), torch.arange(256.0), indexing="ij")
smooth = torch.stack([2 * torch.sin(xx / 41.0 + yy / 67.0), 2 * torch.cos(xx / 53.0 - yy / 37.0)]).unsqueeze(0).repeat(4, 1, 1, 1).to(dev)
rnd = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(dev)
for name, flow in (("random U[-2,2)", rnd), ("smooth, amplitude 2", smooth)):
    gs, gf = torch.zeros_like(src), torch.zeros_like(flow)
    for _ in range(2):
        ops.block_extractor_forward(src, flow, 3, out=out); ops.block_extractor_backward(src, flow, out, 3, gs, gf)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10):
        ops.block_extractor_forward(src, flow, 3, out=out); ops.block_extractor_backward(src, flow, out, 3, gs, gf)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    for k, v in sorted(_lib.prof_collect().items()):
        print("%-22s %-28s %8.1f us  %6.2f TB/s (%4.1f %% of 8 TB/s)" % (name, k, v["avg_ms"] * 1e3, v["bytes_per_launch"] / v["avg_ms"] / 1e9, v["bytes_per_launch"] / v["avg_ms"] / 1e9 / 8 * 100))
