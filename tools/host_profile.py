"""Where the HOST time of one eager train step goes (torch.profiler, CPU side): the step is launch-bound (GPU busy ~78 %)."""
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
t0 = time.perf_counter()
for _ in range(5): t.step(b, batch_increment=0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host-issue time per step %.1f ms, + drain %.1f ms" % ((t1 - t0) / 5 * 1e3, (t2 - t1) * 1e3))
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(2): t.step(b, batch_increment=0)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=28, max_name_column_width=60))
