"""Split-reduction variant of the Winograd kernel (conv_winograd.hip, WinoGeo::CS) on the calls with few (strip, k tile) pairs: own
kernel with the split / without it / the vendor's convolution, forward and data gradient, torch-event us per call (warm)."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import ops, _lib

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

torch.backends.cudnn.benchmark = False
for (B, C, H, K) in ((8, 256, 32, 256), (16, 256, 32, 256), (8, 512, 16, 512), (16, 512, 16, 512), (8, 128, 32, 128), (8, 384, 32, 384), (8, 256, 16, 256), (8, 64, 64, 64)):
    x = torch.randn(B, C, H, H, device="cuda")
    w = torch.randn(K, C, 3, 3, device="cuda") * 0.02
    go = torch.randn(B, K, H, H, device="cuda")
    cs = ops.conv3x3_winograd_splits(B, C, H, H, K, 0)
    row = {}
    for split in (1, 0):
        _lib.set_option("conv_wino_split", split)
        row["fwd split=%d" % split] = timeit(lambda: ops.conv3x3_winograd(x, w))
        row["dgrad split=%d" % split] = timeit(lambda: ops.conv3x3_winograd(go, w, None, data_gradient=True))
    _lib.set_option("conv_wino_split", 1)
    row["vendor fwd"] = timeit(lambda: F.conv2d(x, w, None, 1, 1))
    row["vendor dgrad"] = timeit(lambda: torch.ops.aten.convolution_backward(go, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    print("(%d,%d,%d,%d)->%d splits %d | " % (B, C, H, H, K, cs) + " | ".join("%s %.1f" % kv for kv in row.items()), flush=True)
