"""Find host-side synchronisation inside one eager train step: HIP runtime calls seen by the profiler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from ffwm_amd import trainer, miopen_tuning
miopen_tuning.install()
t = trainer.FFWMTrainer("cuda", seed=0)
b = trainer.synthetic_batch(8, "cuda", seed=1)
t.pretrain_flow_identity(b)
for _ in range(3): t.step(b, batch_increment=0)
torch.cuda.synchronize()
# phase timing on the host: forward+D issue, D step + G forward/backward issue, G step
import types
marks = []
def timed_phase(name, fn):
    t0 = time.perf_counter(); fn(); marks.append((name, (time.perf_counter() - t0) * 1e3))
for _ in range(3):
    marks.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timed_phase("seg_forward_and_D", lambda: t._seg_forward_and_D(b))
    timed_phase("red_D.finish", t.red_D.finish)
    timed_phase("seg_stepD_and_G", lambda: t._seg_stepD_and_G(b))
    timed_phase("red_G.finish", t.red_G.finish)
    timed_phase("seg_stepG", t._seg_stepG)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host %.1f ms, drain %.1f ms:" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), ["%s %.1f" % m for m in marks])
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    t.step(b, batch_increment=0)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("hip") or "ync" in e.key or "emcpy" in e.key]
for e in sorted(rows, key=lambda e: -e.self_cpu_time_total)[:15]:
    print("%-40s calls %5d  self cpu %.2f ms" % (e.key[:40], e.count, e.self_cpu_time_total / 1e3))
