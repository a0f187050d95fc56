"""netG's three warp + flip + cat levels at batch 8 as the train step issues them (multi-problem launches): forward, d(flow), d(feat)
under the library's tuning options.  HIP-event time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import smooth_flow

lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
bs = 8
feats = [torch.rand(bs, c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
flows = [smooth_flow(bs, s).to(dev) for s in (32, 64, 128)]
gos = [torch.rand(bs, 2 * c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
big = torch.empty(256 << 20, device=dev)            # 1 GiB: flushes L2 / MALL between repetitions (the step's launches are HBM-cold)


def run(tag, cold):
    gfe = [torch.zeros_like(f) for f in feats]
    gfl = [torch.zeros_like(f) for f in flows]
    for _ in range(2):
        ops.warp_multi_forward(feats, flows, True)
        ops.warp_multi_backward(feats, flows, gos, True, gfe, gfl)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(8):
        if cold:
            _lib.prof_enable(False); big.fill_(1.0); _lib.prof_enable(True)
        ops.warp_multi_forward(feats, flows, True)
        if cold:
            _lib.prof_enable(False); big.fill_(1.0); _lib.prof_enable(True)
        ops.warp_multi_backward(feats, flows, gos, True, gfe, gfl)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    rows = _lib.prof_collect()
    print("%-28s %s" % (tag + (" cold" if cold else " warm"), {k: round(v["avg_ms"] * 1e3, 1) for k, v in rows.items()}))


for cold in (False, True):
    for slab in (0, 8, 32, 64):
        for remap in (1, 0):
            lib.ffwm_set_option(b"channel_slab", slab)
            lib.ffwm_set_option(b"xcd_remap", remap)
            run("slab %d remap %d" % (slab, remap), cold)
lib.ffwm_set_option(b"channel_slab", 0); lib.ffwm_set_option(b"xcd_remap", 1)
lib.ffwm_set_option(b"warp_fwd_variant", 2)
run("fwd LDS variant", False); run("fwd LDS variant", True)
lib.ffwm_set_option(b"warp_fwd_variant", 0)
for nt in (0, 1):
    lib.ffwm_set_option(b"warp_nt", nt)
    run("nt stores %d" % nt, False); run("nt stores %d" % nt, True)
lib.ffwm_set_option(b"warp_nt", 0)
for order in (1, 0, 1, 0):
    lib.ffwm_set_option(b"warp_multi_order", order)
    run("order %s" % ("caller's" if order else "largest first"), False); run("order %s" % ("caller's" if order else "largest first"), True)
lib.ffwm_set_option(b"warp_multi_order", 0)
