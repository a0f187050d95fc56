import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops, _lib
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
g = torch.Generator().manual_seed(0)
B, C, H, W = 8, 64, 512, 512
in1 = torch.rand(B, C, H, W, generator=g).cuda()
fl = torch.rand(B, 2, H, W, generator=g) * 6 - 3
in2 = torch.cat((fl, torch.full((B, 1, H, W), 2.0)), 1).cuda()
o = torch.empty_like(in1)
for v in (4, 5, 2):
    for ab in (0, 1, 2, 3):
        _lib.set_option("rs_fwd_variant", v); _lib.set_option("ablate", ab)
        print("variant %d ablate %d: %.1f us" % (v, ab, t(lambda: ops.resample2d_forward(in1, in2, 4, 1, out=o))))
_lib.set_option("ablate", 0)
print("copy_ %.1f us" % t(lambda: o.copy_(in1)))
print("fill %.1f us" % t(lambda: o.fill_(1.0)))
print("read (sum) %.1f us" % t(lambda: in1.sum()))
