"""netG's three warp + flip + cat levels at batch 8 as the train step issues them: the multi-problem forward / d(flow) launches on
direct gathers (warp_multi_lds = 1) vs LDS-staged tiles for the large levels (0 = auto, 2 = all levels), per channel slab.
HIP-event time per launch (us), warm (tensors in L2 / MALL) and cold (1 GiB fill between the launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import _lib, ops
from bench import smooth_flow

lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
bs = 8
feats = [torch.rand(bs, c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
flows = [smooth_flow(bs, s).to(dev) for s in (32, 64, 128)]
gos = [torch.rand(bs, 2 * c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
big = torch.empty(256 << 20, device=dev)


def run(tag, cold):
    gfl = [torch.zeros_like(f) for f in flows]
    for _ in range(2):
        ops.warp_multi_forward(feats, flows, True)
        ops.warp_multi_backward(feats, flows, gos, True, [None] * 3, gfl)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(8):
        if cold:
            _lib.prof_enable(False); big.fill_(1.0); _lib.prof_enable(True)
        ops.warp_multi_forward(feats, flows, True)
        if cold:
            _lib.prof_enable(False); big.fill_(1.0); _lib.prof_enable(True)
        ops.warp_multi_backward(feats, flows, gos, True, [None] * 3, gfl)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    rows = _lib.prof_collect()
    print("%-34s %s" % (tag + (" cold" if cold else " warm"), {k: round(v["avg_ms"] * 1e3, 1) for k, v in rows.items()}), flush=True)


for mode, name in ((1, "direct"), (0, "auto"), (2, "lds all")):
    lib.ffwm_set_option(b"warp_multi_lds", mode)
    for slab in ((0,) if mode == 1 else (0, 8, 32, 64)):
        lib.ffwm_set_option(b"channel_slab", slab)
        for cold in (False, True):
            run("%s slab %d" % (name, slab), cold)
lib.ffwm_set_option(b"channel_slab", 0)
lib.ffwm_set_option(b"warp_multi_lds", 0)
