"""Times of the Winograd kernel (+ its thin tail) on the 3x3 layers of the train step, forward with bias + LeakyReLU and data gradient;
prints a step-weighted total.  Used for same-box A/B of library variants (tools/ab/ab.sh).   python tools/wino_ab_layers.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
# (B, C, H, K, forward calls, data-gradient calls) per step (profiles/r05_winograd_layers_of_the_step.txt)
LAYERS = [(8, 195, 128, 195, 4, 4), (8, 128, 128, 128, 3, 3), (8, 195, 64, 195, 4, 4), (8, 384, 32, 384, 4, 4), (8, 128, 64, 128, 5, 4),
          (8, 195, 64, 256, 1, 1), (8, 256, 32, 256, 3, 3), (8, 64, 128, 64, 4, 3), (8, 384, 32, 256, 1, 1), (8, 64, 64, 128, 4, 0), (40, 64, 32, 64, 2, 1)]
if os.environ.get("FFWM_WINO_WS"):
    _lib.set_option("conv_wino_ws", int(os.environ["FFWM_WINO_WS"]))
# the first measurements of a process run on cold clocks (195 -> 195 @128 read 460 us first and 376 us a second later): warm up
_x = torch.randn(8, 256, 128, 128, device="cuda"); _w = torch.randn(256, 256, 3, 3, device="cuda") * 0.05
for _ in range(300): ops.conv3x3_winograd(_x, _w, None)
torch.cuda.synchronize(); del _x, _w
total = 0.0
for B, C, H, K, nf, nd in LAYERS:
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
    gy = torch.randn(B, K, H, H, device="cuda")
    res = {}
    for name, fn in (("fwd", lambda: ops.conv3x3_winograd(x, w, b, act=1, slope=0.2)), ("dgrad", lambda: ops.conv3x3_winograd(gy, w, None, data_gradient=True))):
        for _ in range(3): fn()
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(10): fn()
        torch.cuda.synchronize(); _lib.prof_enable(False)
        pr = _lib.prof_collect()
        res[name] = sum(v["avg_ms"] * 1e3 for k, v in pr.items() if k.startswith("conv_winograd_" + name) or k == "conv3x3_thin_tail")
    total += nf * res["fwd"] + nd * res["dgrad"]
    print("%3d -> %3d @%3d B %2d: fwd %7.1f  dgrad %7.1f us" % (C, K, H, B, res["fwd"], res["dgrad"]), flush=True)
print("step-weighted total %.2f ms" % (total / 1e3))
