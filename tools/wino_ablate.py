import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
B, C, H, K = 8, 256, 128, 256
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
for ab in [0, 0, 1, 2, 3, 7]:
    _lib.set_option("ablate", ab)
    for _ in range(3): ops.conv3x3_winograd(x, w, b)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(5): ops.conv3x3_winograd(x, w, b)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    print("ablate %d: %.1f us" % (ab, _lib.prof_collect()["conv_winograd_fwd"]["avg_ms"] * 1e3))
