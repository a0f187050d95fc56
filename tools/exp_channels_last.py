"""Experiment: channels_last for the conv nets that have no custom op inside (VGG19, LightCNN, netD, FlowNets)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import trainer

which = sys.argv[1].split(",") if len(sys.argv) > 1 else []
torch.backends.cudnn.benchmark = False
dev = torch.device("cuda", 0)
t = trainer.FFWMTrainer(dev, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch)
for name in which:
    getattr(t, name).to(memory_format=torch.channels_last)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


run(3)
print(which, "ms/step", run(10))
