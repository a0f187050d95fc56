"""Steady-state cost of a chunk of the persistent Winograd kernel with parts compiled out (no epilogue in every row; 256 -> 256 @128^2,
batch 16 = 16 pairs of 32 chunks per CU).   python tools/wino_chunk_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
B, C, H, K = 16, 256, 128, 256
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
NAMES = {0: "full kernel", 32: "no epilogue", 35: "no epilogue, no global loads", 36: "no epilogue, no LDS commits (no patch reads)", 39: "no epilogue, no loads, no commits",
         40: "no epilogue, no operand reads", 47: "no epilogue: MFMAs + barrier only",
         99: "no epi, no loads, no U writes", 163: "no epi, no loads, no V writes (no transform)", 291: "no epi, no loads, no raw writes",
         547: "no epi, no loads, no patch reads", 227: "no epi, no loads, no U / V writes"}
for ab in (0, 32, 35, 99, 163, 291, 547, 227, 39, 47):
    _lib.set_option("ablate", ab)
    for _ in range(3): ops.conv3x3_winograd(x, w, b)
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): ops.conv3x3_winograd(x, w, b)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    t = _lib.prof_collect()["conv_winograd_fwd"]["avg_ms"] * 1e3
    print("%-50s %8.1f us = %.3f us per chunk (MFMA-bound: 1.707)" % (NAMES[ab], t, t / (16 * 32)), flush=True)
_lib.set_option("ablate", 0)
