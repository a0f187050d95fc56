"""Experiment: FlowNetF forward (cfg-2, bs=6) eager vs hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import nets

torch.backends.cudnn.benchmark = False
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = nets.FlowNet(64).to(dev).eval()
x = torch.rand(6, 3, 128, 128, device=dev)


def run(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager():
    with torch.no_grad():
        return net(x)


print("eager ms", run(eager))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        eager()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = eager()
print("graph ms", run(g.replay))
ref = eager()
g.replay()
torch.cuda.synchronize()
print("max diff", max(float((a - b).abs().max()) for a, b in zip(ref, out)))
