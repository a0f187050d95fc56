"""Experiment: does running the train step on a side stream change MIOpen's solver choice / speed?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import trainer

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
torch.backends.cudnn.benchmark = False
dev = torch.device("cuda", 0)
t = trainer.FFWMTrainer(dev, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if mode == "default":
    run(3)
    print(mode, "ms/step", run(5))
elif mode == "side":
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(3)
        print(mode, "ms/step", run(5))
elif mode == "default_then_side":
    run(3)
    print("default", "ms/step", run(5))
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run(2)
        print("side after default", "ms/step", run(5))
elif mode == "graph_after_default":
    run(3)
    print("default", "ms/step", run(5))
    t.capture(batch, warmup=2)
    run(2)
    print("graph after default warmup", "ms/step", run(5))
