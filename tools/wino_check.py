"""conv3x3_winograd vs fp64 ATen + timing next to MIOpen (HIP events, warm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ffwm_amd import ops, _lib
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (B, C, H, W, K) in [(1, 8, 4, 4, 64), (2, 5, 6, 10, 3), (2, 19, 7, 9, 70), (1, 64, 32, 32, 64)]:
    x = torch.randn(B, C, H, W, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.2; b = torch.randn(K, device="cuda")
    y = ops.conv3x3_winograd(x, w, b)
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    e = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    gy = torch.randn(B, K, H, W, device="cuda")
    dx = ops.conv3x3_winograd(gy, w, None, data_gradient=True)
    dref = torch.nn.grad.conv2d_input((B, C, H, W), w.double(), gy.double(), 1, 1)
    e2 = (dx.double() - dref).abs().max().item() / dref.abs().max().item()
    ya = ops.conv3x3_winograd(x, w, b, act=1, slope=0.2)
    e3 = (ya.double() - F.leaky_relu(ref, 0.2)).abs().max().item() / ref.abs().max().item()
    print("B%d C%d %dx%d K%d  fwd rel err %.2e  dgrad %.2e  lrelu %.2e" % (B, C, H, W, K, e, e2, e3))
if len(sys.argv) > 1 and sys.argv[1] == "check": sys.exit(0)
for (B, C, H, K) in [(8, 195, 128, 195), (8, 195, 64, 195), (8, 128, 128, 128), (8, 384, 32, 384), (8, 256, 32, 256), (8, 64, 128, 64), (8, 128, 64, 128), (8, 256, 16, 256), (8, 512, 8, 512), (8, 192, 128, 192), (8, 256, 128, 256)]:
    x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
    mine = t(lambda: ops.conv3x3_winograd(x, w, b))
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(5): ops.conv3x3_winograd(x, w, b)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    kr = _lib.prof_collect()
    # (a few-pair call runs the split instantiation under its own scope; a frozen layer's transform may come from the cache)
    kus = sum(v["avg_ms"] for k, v in kr.items() if k.startswith("conv_winograd_fwd")) * 1e3
    wus = kr.get("conv_winograd_weights", {"avg_ms": 0.0})["avg_ms"] * 1e3
    vend = t(lambda: F.conv2d(x, w, b, 1, 1))
    gf = 2.0 * B * H * H * K * C * 9 / 1e9
    print("C=%3d H=%3d K=%3d  kernel %7.1f us (%6.1f TF direct-equiv, %5.1f TF mfma) weights %5.1f us  call %7.1f us   MIOpen %7.1f us (%5.1f TF)" % (
        C, H, K, kus, gf / kus * 1e3, gf / kus * 1e3 * 16 / 36, wus, mine, vend, gf / vend * 1e3))
