import sys; sys.path.insert(0,'/root/repo')
import torch
from ffwm_amd import _lib
from ffwm_amd.losses import MultiAffineRegularizationLoss
dev="cuda:0"
g=torch.Generator().manual_seed(0)
flows=[(torch.rand(6,2,s,s,generator=g)*2-1).to(dev).requires_grad_(True) for s in (32,64,128)]
m=MultiAffineRegularizationLoss({1:7,2:5,3:3})
def step():
    for f in flows: f.grad=None
    m(flows).backward()
for _ in range(3): step()
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(5): step()
torch.cuda.synchronize()
_lib.prof_enable(False)
for k,r in sorted(_lib.prof_collect().items(), key=lambda kv:-kv[1]["total_ms"]): print(k, r["launches"], round(r["avg_ms"]*1e3,1),"us")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=8, max_name_column_width=60))
