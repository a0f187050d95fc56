"""warp d(flow), multi-problem launch of netG's three levels (batch 8): one 8-byte load per corner row (warp_pair_loads=1) against two
dword gathers (=0); HIP-event us per launch, warm caches and flushed caches; parity of the two variants (bit-identical expected)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import _lib, ops
from bench import smooth_flow

dev = "cuda"
g = torch.Generator().manual_seed(0)
bs = 8
feats = [torch.rand(bs, c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
flows = [smooth_flow(bs, s).to(dev) for s in (32, 64, 128)]
gos = [torch.rand(bs, 2 * c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
big = torch.empty(256 << 20, device=dev)


def run(cold):
    gfe = [torch.zeros_like(f) for f in feats]
    gfl = [torch.zeros_like(f) for f in flows]
    for _ in range(2):
        ops.warp_multi_backward(feats, flows, gos, True, gfe, gfl)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10):
        if cold:
            _lib.prof_enable(False); big.fill_(1.0); _lib.prof_enable(True)
        ops.warp_multi_backward(feats, flows, gos, True, gfe, gfl)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    rows = _lib.prof_collect()
    return round(rows["warp_flipcat_bwd_flow_multi"]["avg_ms"] * 1e3, 1)


res = {}
for pair in (1, 0, 1, 0):
    _lib.set_option("warp_pair_loads", pair)
    print("pair_loads", pair, "warm", run(False), "cold", run(True), flush=True)
    gfl = [torch.zeros_like(f) for f in flows]
    ops.warp_multi_backward(feats, flows, gos, True, [torch.zeros_like(f) for f in feats], gfl)
    res[pair] = [t.clone() for t in gfl]
_lib.set_option("warp_pair_loads", 1)
print("max |difference| of d(flow), pair vs dword loads:", [float((a - b).abs().max()) for a, b in zip(res[1], res[0])],
      "scale", [float(b.abs().max()) for b in res[0]])
# a flow with samples on and beyond the borders (half rows: the dword path inside the pair kernel)
wide = [(f * 1.3).contiguous() for f in flows]
out = {}
for pair in (1, 0):
    _lib.set_option("warp_pair_loads", pair)
    gfl = [torch.zeros_like(f) for f in wide]
    ops.warp_multi_backward(feats, wide, gos, True, [torch.zeros_like(f) for f in feats], gfl)
    out[pair] = gfl
_lib.set_option("warp_pair_loads", 1)
print("border flow: max |difference|", [float((a - b).abs().max()) for a, b in zip(out[1], out[0])])
