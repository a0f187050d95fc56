"""netG's warp + flip + cat backward at the HBM-resident shape [32,64,256,256], smooth flow, overwrite mode, cold caches: d(feat) owned tiles with
32-bit fixed-point cells (warp_feat_fixed = 1) against the double cells of rounds 3-5 (0, the default); HIP-event time per launch scope, us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
feat = torch.rand(32, 64, 256, 256, generator=g).to(dev)
nflow = bench.smooth_flow(32, 256).to(dev)
wo = torch.rand(32, 128, 256, 256, generator=g).to(dev)
gfeat, gflow = torch.empty_like(feat), torch.zeros_like(nflow)
flush = torch.empty(128 << 20, device=dev)
for rep in range(2):
    for mode in ([int(a) for a in sys.argv[1:]] or [0, 3]):
        _lib.set_option("warp_feat_fixed", mode)
        for _ in range(2):
            ops.warp_backward(feat, nflow, wo, True, gfeat, gflow, overwrite_feat=True)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            flush.sum()
            ops.warp_backward(feat, nflow, wo, True, gfeat, gflow, overwrite_feat=True)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("warp_feat_fixed=%d" % mode, {k.replace("warp_flipcat_", ""): round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
_lib.set_option("warp_feat_fixed", 0)
