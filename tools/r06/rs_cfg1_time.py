"""resample2d d_input1 at cfg-1 ([1,64,128,128], ks 4, warm): the tap-lane kernel (default below 2^18 pixels) against the owned tiles forced
onto the small call (rs_bwd1_owned_min_pixels = 1), several slab sizes.  HIP-event time per scope, us."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
for shape in ((1, 64, 128, 128), (8, 64, 128, 128), (2, 64, 256, 256)):
    B, C, H, W = shape
    in1 = torch.rand(B, C, H, W, generator=g).to(dev)
    in2 = torch.cat((torch.rand(B, 2, H, W, generator=g) * 6 - 3, torch.full((B, 1, H, W), 2.0)), 1).to(dev)
    go = torch.rand(B, C, H, W, generator=g).to(dev)
    g1 = torch.zeros_like(in1)
    for minpix, blocks in ((0, 0), (1, 0), (1, 256), (1, 512), (1, 2048)):
        _lib.set_option("rs_bwd1_owned_min_pixels", minpix)
        _lib.set_option("rs_bwd1_owned_blocks", blocks)
        for _ in range(3):
            ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(20):
            ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print(shape, "min_pixels", minpix, "blocks", blocks, {k.replace("resample2d_bwd_input1_", ""): round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
_lib.set_option("rs_bwd1_owned_min_pixels", 0); _lib.set_option("rs_bwd1_owned_blocks", 0)
