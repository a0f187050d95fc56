"""Which Python lines launch the small elementwise kernels of the train step?  One eager step under torch.profiler with stacks: fills,
adds, copies and reductions grouped by the innermost frame inside ffwm_amd/ (or by the aten op when none).   python tools/step_small_ops.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ffwm_amd import trainer
dev = torch.device("cuda", 0)
t = trainer.FFWMTrainer(dev, world_size=1, seed=0, titers=1)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch)
for _ in range(2): t.step(batch, batch_increment=0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
ev = prof.events()
# map: cpu op -> its kernels' time; walk up to find the ffwm frame
acc = collections.defaultdict(lambda: [0, 0.0])
WATCH = ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::copy_", "aten::mul", "aten::sum", "aten::clone", "aten::contiguous",
         "aten::ones_like", "aten::sub", "aten::div", "aten::neg", "aten::mean", "aten::cat")
for e in ev:
    if e.name not in WATCH or not e.kernels:
        continue
    # only leaf-most watched ops: skip when a child is also watched and has kernels
    if any(c.name in WATCH and c.kernels for c in e.cpu_children):
        continue
    frame = None
    for s in (e.stack or []):
        if "ffwm_amd/" in s or "bench.py" in s:
            frame = s.split("ffwm_amd/")[-1] if "ffwm_amd/" in s else s
            break
    if frame is None:
        frame = "(autograd engine / torch internals)"
    k = (e.name, frame[:110])
    acc[k][0] += len(e.kernels)
    acc[k][1] += sum(kk.duration for kk in e.kernels)
rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
tot_n = sum(v[0] for v in acc.values()); tot_t = sum(v[1] for v in acc.values())
print("watched small ops of one eager step: %d kernels, %.2f ms" % (tot_n, tot_t / 1e3))
for (name, frame), (n, tt) in rows[:45]:
    print("%4d x %8.1f us  %-18s %s" % (n, tt, name, frame))
