"""warp + flip + cat d(feat) at the HBM-resident shape [32,64,256,256] (owned-tile kernel, warp.hip): channel groups per block.
The auto choice (split until >= 768 blocks) leaves 800 blocks of 32 groups each for 512 resident slots (119 VGPRs: two 8-wave blocks per
CU): two rounds, the second 44 % empty."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = "cuda"
g = torch.Generator().manual_seed(0)
feat = torch.rand(32, 64, 256, 256, generator=g).to(dev)
nflow = bench.smooth_flow(32, 256).to(dev)
wo = torch.rand(32, 128, 256, 256, generator=g).to(dev)
gfeat, gflow = torch.zeros_like(feat), torch.zeros_like(nflow)
ref = None
for gps in (0, 32, 16, 8, 4, 2, 1):
    _lib.set_option("warp_feat_gps", gps)
    for _ in range(2):
        ops.warp_backward(feat, nflow, wo, True, gfeat, None)
    gfeat.zero_()
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(3):
        ops.warp_backward(feat, nflow, wo, True, gfeat, None)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    out = gfeat / 3
    if ref is None:
        ref = out.clone()
    print("groups per block %2d" % gps, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()},
          "max diff vs auto %.2g" % float((out - ref).abs().max()), flush=True)
_lib.set_option("warp_feat_gps", 0)
# overwrite mode (flipcat bit 1): grad_feat uninitialised, the tiles store instead of read-modify-write
gfeat.fill_(float("nan"))
for _ in range(2):
    ops.warp_backward(feat, nflow, wo, True, gfeat, None, overwrite_feat=True)
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
for _ in range(3):
    ops.warp_backward(feat, nflow, wo, True, gfeat, None, overwrite_feat=True)
torch.cuda.synchronize(); _lib.prof_enable(False)
print("overwrite mode      ", {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()},
      "max diff vs accumulate mode %.2g" % float((gfeat - ref).abs().max()), flush=True)
