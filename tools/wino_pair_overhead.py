"""Where the per-pair cost of the persistent Winograd kernel goes: the batch sets the pairs per CU (slope = cost of a pair, intercept =
cost of a launch), with the epilogue's global stores (ablate 16) or the whole epilogue (ablate 32) compiled out.
    python tools/wino_pair_overhead.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
for C in (64, 128, 256):
    K, H = C, 128
    for B in (8 // (C // 64), 16 // (C // 64), 32 // (C // 64), 64 // (C // 64)):
        x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
        row = []
        for ab in [0, 16, 32]:
            _lib.set_option("ablate", ab)
            for _ in range(3): ops.conv3x3_winograd(x, w, b)
            _lib.prof_reset(); _lib.prof_enable(True)
            for _ in range(10): ops.conv3x3_winograd(x, w, b)
            torch.cuda.synchronize(); _lib.prof_enable(False)
            row.append(_lib.prof_collect()["conv_winograd_fwd"]["avg_ms"] * 1e3)
        _lib.set_option("ablate", 0)
        pairs = B * (H // 2) * (H // 2) // 64 * (K // 64)
        print("%d -> %d @%d B %d: %.2f pairs per CU of %d chunks: full %.1f | no stores %.1f | no epilogue %.1f us" % (
            C, K, H, B, pairs / 256, (C + 7) // 8, row[0], row[1], row[2]), flush=True)
