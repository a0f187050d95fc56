"""Tiled MFMA weight gradient (csrc/conv_bwd.hip, conv_wgrad_tile_kernel) on shapes of the train step that it serves (stride-2 3x3, small planes,
4x4): kernel time by scope and the error against ATen float64.  For same-box A/B of library variants.   python tools/wgrad_tile_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
SHAPES = [(8, 256, 32, 256, 3, 1), (8, 384, 32, 384, 3, 1), (8, 128, 32, 128, 3, 1), (8, 512, 16, 512, 3, 1), (8, 64, 128, 128, 3, 2), (8, 128, 64, 256, 3, 2),
          (8, 256, 32, 512, 3, 2), (6, 512, 8, 512, 3, 1), (6, 1024, 2, 1024, 3, 1), (8, 64, 64, 64, 3, 1), (8, 3, 128, 64, 3, 1), (8, 195, 64, 195, 4, 2)]
if os.environ.get("FFWM_SLICE_TARGET"):
    _lib.set_option("conv_wgrad_slice_target", int(os.environ["FFWM_SLICE_TARGET"]))
g = torch.Generator().manual_seed(0)
tot = 0.0
for B, C, H, K, k, s in SHAPES:
    x = torch.randn(B, C, H, H, generator=g).cuda()
    Ho = (H + 2 - k) // s + 1
    go = (torch.randn(B, K, Ho, Ho, generator=g) * 0.1).cuda()
    if not ops.conv2d_wgrad_tiled_ok(go):
        print("skip", (B, C, H, K, k, s)); continue
    for _ in range(3): gw, gb = ops.conv2d_wgrad_tiled(go, x, k, s, 1, want_bias=True)
    torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): gw, gb = ops.conv2d_wgrad_tiled(go, x, k, s, 1, want_bias=True)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    pr = {n: round(v["avg_ms"] * 1e3, 1) for n, v in _lib.prof_collect().items()}
    ref = torch.ops.aten.convolution_backward(go.double(), x.double(), torch.zeros(K, C, k, k, device="cuda", dtype=torch.float64), [K], [s, s], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
    err = (gw.double().reshape(ref[1].shape) - ref[1]).abs().max().item() / ref[1].abs().max().item()
    eb = (gb.double() - ref[2]).abs().max().item() / ref[2].abs().max().item()
    t = sum(pr.values()); tot += t
    print("B %d C %4d H %3d K %4d k%d s%d: %s  err %.1e bias %.1e" % (B, C, H, K, k, s, pr, err, eb), flush=True)
print("sum %.1f us" % tot)
