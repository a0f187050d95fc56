"""Why is the forward of a layer slower than its data gradient?  Kernel times by scope with / without bias and activation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
for B, C, H, K in ((8, 195, 128, 195), (8, 128, 128, 128)):
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
    wt = w.transpose(0, 1).contiguous()
    cases = {"fwd bias+act": lambda: ops.conv3x3_winograd(x, w, b, act=1, slope=0.2), "fwd bias": lambda: ops.conv3x3_winograd(x, w, b),
             "fwd plain": lambda: ops.conv3x3_winograd(x, w, None), "fwd act": lambda: ops.conv3x3_winograd(x, w, None, act=1, slope=0.2),
             "dgrad": lambda: ops.conv3x3_winograd(x, w, None, data_gradient=True), "fwd plain, transposed weight": lambda: ops.conv3x3_winograd(x, wt, None)}
    for name, fn in cases.items():
        for _ in range(3): fn()
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(10): fn()
        torch.cuda.synchronize(); _lib.prof_enable(False)
        print("%d -> %d @%d  %-30s %s" % (C, K, H, name, {k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()}), flush=True)
