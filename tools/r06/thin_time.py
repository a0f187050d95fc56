"""FlowNet(64)'s thin full-resolution layers at batch 6: the direct kernel (ffwm_conv_thin_forward, variants: 1 / 2 = 8 / 16
output channels per lane, +4 = one input channel per step) against the Winograd kernel the lean path ran before.  HIP events, warm, us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ffwm_amd import _lib, ops, flownet_eval as fe
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B = 6
for name, C, H, K, tr in (("conv0", 6, 128, 64, False), ("inter1", 34, 64, 32, False), ("inter0", 18, 128, 16, False)):
    x = torch.randn(B, C, H, H, device="cuda")
    w = torch.randn(*((C, K, 4, 4) if tr else (K, C, 3, 3)), device="cuda") * 0.05
    b = torch.randn(K, device="cuda")
    wt = fe.thin_weights(w)
    row = {}
    if tr:
        pass
    else:
        cache = {}
        row["winograd"] = t(lambda: ops.conv3x3_winograd(x, w, b, act=1, slope=0.2, frozen=cache))
        for v in (1, 2, 5, 6):
            _lib.set_option("conv_thin_variant", v)
            row["direct v%d" % v] = t(lambda: fe.conv_thin(x, wt, b, fe.LRELU, 0.2))
        _lib.set_option("conv_thin_variant", 0)
    print(name, {k: round(v, 1) for k, v in row.items()})
