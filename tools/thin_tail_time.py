"""Winograd forward / data gradient of netG's 195 -> 195 layers at 128^2 and 64^2 (batch 8) and the 195 -> 3 image head: HIP-event time of
the launches (conv_winograd_fwd / _dgrad, conv3x3_thin_tail)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import _lib, ops
dev = "cuda"
g = torch.Generator().manual_seed(0)
for (C, K, S) in ((195, 195, 128), (195, 195, 64), (195, 3, 128), (384, 3, 32)):
    x = torch.randn(8, C, S, S, generator=g).to(dev)
    w = (torch.randn(K, C, 3, 3, generator=g) * 0.02).to(dev)
    b = torch.randn(K, generator=g).to(dev)
    o = torch.empty(8, K, S, S, device=dev)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1)
    for _ in range(3):
        ops.conv3x3_winograd(x, w, b, out=o)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10):
        ops.conv3x3_winograd(x, w, b, out=o)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    err = float((o.double() - ref).abs().max() / (1 + ref.abs().max()))
    print("%d -> %d @%d" % (C, K, S), {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}, "error %.2g" % err, flush=True)
