"""Every distinct Winograd call of one train step (shape, direction, count) timed against the vendor library's kernel for
the same convolution: which shapes the route should keep."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ffwm_amd import ops, trainer, _lib

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

dev = torch.device("cuda", 0)
tr = trainer.FFWMTrainer(dev, world_size=1, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
tr.pretrain_flow_identity(batch, steps=5)
for _ in range(2): tr.step(batch, batch_increment=0)
calls = collections.Counter()
orig = ops.conv3x3_winograd
def spy(x, weight, bias=None, data_gradient=False, **kw):
    B, C, H, W = x.shape
    K = weight.shape[1] if data_gradient else weight.shape[0]
    calls[(B, C, H, W, K, bool(data_gradient), kw.get("act", 0))] += 1
    return orig(x, weight, bias, data_gradient, **kw)
ops.conv3x3_winograd = spy
tr.step(batch, batch_increment=0)
torch.cuda.synchronize()
ops.conv3x3_winograd = orig
tot_m = tot_v = 0.0
print("%-34s %5s %9s %9s %8s" % ("B,C,H,W -> K", "calls", "mine us", "vendor us", "gain ms"))
for (B, C, H, W, K, dg, act), n in sorted(calls.items(), key=lambda kv: -kv[1] * kv[0][1] * kv[0][4] * kv[0][2] * kv[0][3]):
    x = torch.randn(B, C, H, W, device=dev)
    if dg:
        w = torch.randn(C, K, 3, 3, device=dev) * 0.05
        mine = t(lambda: ops.conv3x3_winograd(x, w, None, data_gradient=True))
        vend = t(lambda: torch.ops.aten.convolution_backward(x, torch.empty(B, K, H, W, device=dev), w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    else:
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        b = torch.randn(K, device=dev)
        mine = t(lambda: ops.conv3x3_winograd(x, w, b, act=act))
        vend = t(lambda: F.conv2d(x, w, b, 1, 1))
    gain = (vend - mine) * n / 1e3
    tot_m += mine * n / 1e3; tot_v += vend * n / 1e3
    print("%-34s %5d %9.1f %9.1f %8.2f  %s" % ("%d,%d,%d,%d -> %d %s" % (B, C, H, W, K, "dgrad" if dg else "fwd"), n, mine, vend, gain, "<-- slower" if gain < 0 else ""))
print("total per step: mine %.2f ms, vendor %.2f ms" % (tot_m, tot_v))
