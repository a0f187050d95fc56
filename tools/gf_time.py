"""guided filter fwd + bwd of the train step's call ([8,3,128,128], r = 32): HIP-event time per direction (GF_NOPROF=1: no
events, for tools/gf_trace.sh)."""
import os, torch, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import ops, _lib
x = torch.rand(8, 3, 128, 128, device="cuda"); y = torch.rand_like(x); go = torch.rand_like(x)
for _ in range(3):
    out, saved = ops.guided_filter_forward(x, y, 32); gx = ops.guided_filter_backward(x, y, saved, go, 32)
torch.cuda.synchronize(); _lib.prof_reset(); _lib.prof_enable(os.environ.get("GF_NOPROF") != "1")
for _ in range(20):
    out, saved = ops.guided_filter_forward(x, y, 32); gx = ops.guided_filter_backward(x, y, saved, go, 32)
torch.cuda.synchronize(); _lib.prof_enable(False)
print({k: round(v["avg_ms"] * 1e3, 1) for k, v in _lib.prof_collect().items()})
