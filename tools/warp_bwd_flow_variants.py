import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
from bench import smooth_flow
lib = _lib.load(); dev = "cuda"
g = torch.Generator().manual_seed(0)
for (B, C, S) in ((8, 64, 128), (8, 64, 64), (32, 64, 256)):
    feat = torch.rand(B, C, S, S, generator=g).to(dev); flow = smooth_flow(B, S).to(dev)
    go = torch.rand(B, 2 * C, S, S, generator=g).to(dev); gfl = torch.zeros_like(flow)
    for name, opts in (("direct", {"warp_multi_lds": 1}), ("lds cs auto", {"warp_fwd_variant": 2}), ("lds cs8", {"warp_fwd_variant": 2, "channel_slab": 8}),
                       ("lds cs16", {"warp_fwd_variant": 2, "channel_slab": 16}), ("lds cs32", {"warp_fwd_variant": 2, "channel_slab": 32}), ("lds cs64", {"warp_fwd_variant": 2, "channel_slab": 64})):
        for k, v in opts.items(): lib.ffwm_set_option(k.encode(), v)
        for _ in range(3): ops.warp_backward(feat, flow, go, True, None, gfl)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(10): ops.warp_backward(feat, flow, go, True, None, gfl)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print((B, C, S), name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, flush=True)
        for k in opts: lib.ffwm_set_option(k.encode(), 0)
