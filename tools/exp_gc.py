"""Experiment: does Python's cyclic GC cost step time? (gc.freeze + gc.disable around the timed steps)"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import _lib, trainer
dev = torch.device("cuda", 0)
_lib.load()
t = trainer.FFWMTrainer(dev, world_size=1, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch, steps=20)
def run(n=15):
    for _ in range(3):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("gc on : %.2f ms" % run())
gc.collect(); gc.freeze(); gc.disable()
print("gc off: %.2f ms" % run())
gc.enable()
print("gc on : %.2f ms" % run())
