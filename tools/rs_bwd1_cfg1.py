"""resample2d d_input1 at BASELINE configs[0] ([1,64,128,128], ks 4, flow ~ U[-3,3)) and at [8,64,512,512]: the tap-lane kernel (option
rs_bwd1_variant 0 = default, 5 = 16-row tiles) against round 2's plane / tile kernels (1 = plane where it fits, 2 = tile), channel
slabs, and the ablations of the tile kernel; HIP-event time of the backward launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import _lib, ops
lib = _lib.load()
g = torch.Generator().manual_seed(0)
dev = "cuda"
for shape in ((1, 64, 128, 128), (8, 64, 512, 512)):
    B, C, H, W = shape
    in1 = torch.rand(*shape, generator=g).to(dev)
    in2 = torch.cat((torch.rand(B, 2, H, W, generator=g) * 6 - 3, torch.full((B, 1, H, W), 2.0)), 1).to(dev)
    go = torch.rand(*shape, generator=g).to(dev)
    g1 = torch.zeros_like(in1)
    for variant in (0, 5, 1, 2):
        for slab in ((0, 16) if variant in (0, 5) else (0,)):
            lib.ffwm_set_option(b"rs_bwd1_variant", variant)
            lib.ffwm_set_option(b"channel_slab", slab)
            for _ in range(2):
                ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
            torch.cuda.synchronize()
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(5):
                ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
            torch.cuda.synchronize(); _lib.prof_enable(False)
            rows = _lib.prof_collect()
            print(shape, "variant", variant, "slab", slab, {k: round(v["avg_ms"] * 1e3, 1) for k, v in rows.items()})
    lib.ffwm_set_option(b"rs_bwd1_variant", 0)
    lib.ffwm_set_option(b"channel_slab", 0)

# ablations (bench-only option `ablate`: 1 = no LDS atomics, 2 = no global atomics in the fold, 16 = gradients not used (g = 1))
for shape in ((1, 64, 128, 128), (8, 64, 512, 512)):
    B, C, H, W = shape
    in1 = torch.rand(*shape, generator=g).to(dev)
    in2 = torch.cat((torch.rand(B, 2, H, W, generator=g) * 6 - 3, torch.full((B, 1, H, W), 2.0)), 1).to(dev)
    go = torch.rand(*shape, generator=g).to(dev)
    g1 = torch.zeros_like(in1)
    for variant in (5, 2):
        lib.ffwm_set_option(b"rs_bwd1_variant", variant)
        for ab in (0, 1, 2, 3, 16):
            lib.ffwm_set_option(b"ablate", ab)
            for _ in range(2):
                ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
            torch.cuda.synchronize()
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(5):
                ops.resample2d_backward(in1, in2, go, 4, 1, g1, None)
            torch.cuda.synchronize(); _lib.prof_enable(False)
            print(shape, "variant", variant, "ablate", ab, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
    lib.ffwm_set_option(b"ablate", 0)
    lib.ffwm_set_option(b"rs_bwd1_variant", 0)
