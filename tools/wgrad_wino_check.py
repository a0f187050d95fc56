"""Winograd-domain weight gradient (csrc/conv_wgrad_wino.hip, option conv_wgrad_wino = 1) against the direct MFMA kernel and ATen fp64:
max error relative to the result's scale, HIP-event time per launch, on the train step's layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import _lib, ops
lib = _lib.load()
dev = "cuda"
shapes = [(2, 64, 64, 64, 64), (2, 67, 131, 64, 128), (8, 195, 195, 128, 128), (8, 128, 128, 128, 128), (8, 64, 64, 128, 128), (8, 195, 195, 64, 64), (8, 128, 128, 64, 64), (8, 195, 256, 64, 64)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
g = torch.Generator().manual_seed(0)
for (B, C, K, H, W) in shapes:
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, K, H, W, generator=g).to(dev) * 0.1
    ref = None
    if B * C * K * H * W <= 8 * 195 * 195 * 64 * 64:
        ref = torch.ops.aten.convolution_backward(go.double(), x.double(), torch.zeros(K, C, 3, 3, device=dev, dtype=torch.float64), [K], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
    out = {}
    for mode in (0, 1, 2):
        lib.ffwm_set_option(b"conv_wgrad_wino", 2 if mode == 0 else 1)
        lib.ffwm_set_option(b"xcd_remap", 0 if mode == 2 else 1)
        for _ in range(2):
            gw = torch.zeros(K, C, 3, 3, device=dev); gb = torch.zeros(K, device=dev)
            ops.conv3x3_wgrad(x, go, gw, gb)
        torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(5):
            gw = torch.zeros(K, C, 3, 3, device=dev); gb = torch.zeros(K, device=dev)
            ops.conv3x3_wgrad(x, go, gw, gb)
        torch.cuda.synchronize(); _lib.prof_enable(False)
        out[mode] = (gw, gb, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
    lib.ffwm_set_option(b"conv_wgrad_wino", 0)
    scale = out[0][0].abs().max().item()
    lib.ffwm_set_option(b"xcd_remap", 1)
    msg = "(%d,%d->%d,%d,%d) direct %s | wino %s | wino, no xcd remap %s | wino-vs-direct %.2e" % (B, C, K, H, W, out[0][2], out[1][2], out[2][2].get("conv3x3_wgrad_winograd"), (out[1][0] - out[0][0]).abs().max().item() / scale)
    if ref is not None:
        msg += " | vs fp64: direct %.2e wino %.2e bias %.2e" % ((out[0][0].double() - ref[1]).abs().max().item() / scale, (out[1][0].double() - ref[1]).abs().max().item() / scale,
                                                                  (out[1][1].double() - ref[2]).abs().max().item() / ref[2].abs().max().item())
    print(msg, flush=True)
