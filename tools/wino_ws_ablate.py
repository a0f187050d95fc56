import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
B, C, H, K = 16, 256, 128, 256
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
for _ in range(100): ops.conv3x3_winograd(x, w, b)
for ws, ab, name in ((0, 0, "standard kernel"), (1, 0, "wave-specialised"), (1, 1, "ws: producers alone"), (1, 2, "ws: consumers alone"), (1, 4, "ws: no U commits"), (1, 8, "ws: no transform (patch reads, V writes)"), (1, 16, "ws: no raw writes"), (1, 32, "ws: no global loads"), (1, 28, "ws: global loads only"), (0, 0, "standard again")):
    _lib.set_option("conv_wino_ws", ws); _lib.set_option("ablate", ab)
    for _ in range(3): ops.conv3x3_winograd(x, w, b)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): ops.conv3x3_winograd(x, w, b)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    t = _lib.prof_collect()["conv_winograd_fwd"]["avg_ms"] * 1e3
    print("%-44s %8.1f us = %.3f us per chunk (16 pairs x 32 chunks per CU)" % (name, t, t / 512), flush=True)
_lib.set_option("conv_wino_ws", 0); _lib.set_option("ablate", 0)
