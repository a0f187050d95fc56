"""Which ATen ops does one steady-state train step issue, and how much device time do they take?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import _lib, trainer

dev = torch.device("cuda", 0)
_lib.load()
t = trainer.FFWMTrainer(dev, world_size=1, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch, steps=10)
for _ in range(3):
    t.step(batch, batch_increment=0)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:110]))
tot = {}
for dt, n, k, s in rows:
    a = tot.setdefault(k, [0, 0.0]); a[0] += n; a[1] += dt
print("---- by op (self device time)")
for k, (n, dt) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%9.1f us x%-5d %s" % (dt, n, k))
print("---- add / add_ by shape")
for dt, n, k, s in sorted(rows, reverse=True):
    if k in ("aten::add", "aten::add_") and dt > 40:
        print("%9.1f us x%-4d %-12s %s" % (dt, n, k, s))
