"""The wave-specialised Winograd kernel (option conv_wino_ws = 1) against the standard persistent kernel: same bits expected (the consumer
waves run the same MFMA order and the same output transform), and ATen float64.   python tools/wino_ws_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
g = torch.Generator().manual_seed(0)
bad = 0
for B, C, H, K, act in [(2, 64, 32, 64, 0), (2, 195, 64, 195, 1), (8, 128, 128, 128, 1), (3, 72, 16, 130, 0), (8, 195, 128, 195, 1), (2, 200, 64, 64, 0), (16, 64, 32, 192, 1)]:
    x = torch.randn(B, C, H, H, generator=g).cuda(); w = (torch.randn(K, C, 3, 3, generator=g) * 0.05).cuda(); b = torch.randn(K, generator=g).cuda()
    outs = []
    for ws in (0, 1):
        _lib.set_option("conv_wino_ws", ws)
        outs.append(ops.conv3x3_winograd(x, w, b, act=act, slope=0.2).clone())
        outs.append(ops.conv3x3_winograd(x, w.transpose(0, 1).contiguous() if False else w, None, data_gradient=False).clone())
    _lib.set_option("conv_wino_ws", 0)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if act: ref = torch.nn.functional.leaky_relu(ref, 0.2)
    e_ws = (outs[2].double() - ref).abs().max().item() / ref.abs().max().item()
    same = bool((outs[0] == outs[2]).all()) and bool((outs[1] == outs[3]).all())
    d = (outs[0] - outs[2]).abs().max().item()
    print("B %d C %d H %d K %d act %d: ws vs fp64 %.2e, bit-identical to the standard kernel: %s (max diff %.2e)" % (B, C, H, K, act, e_ws, same, d), flush=True)
    bad += (e_ws > 2e-5)
print("FAILURES" if bad else "OK")
