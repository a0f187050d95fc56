"""Experiment: (1) hipGraph replay overhead for a long chain of tiny kernels; (2) which MIOpen kernels a captured
conv forward+backward replays."""
import sys, time
import torch
import torch.nn as nn
import torch.nn.functional as F

dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = False
mode = sys.argv[1]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


if mode == "chain":
    x = torch.zeros(1024, device=dev)
    N = 3000

    def eager():
        for _ in range(N):
            x.add_(1.0)
    print("eager %d tiny kernels: %.2f ms" % (N, timeit(eager)))
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eager()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eager()
    print("graph replay %d nodes: %.2f ms" % (N, timeit(g.replay)))
else:
    torch.manual_seed(0)
    net = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1), nn.LeakyReLU(0.2), nn.Conv2d(64, 128, 4, 2, 1), nn.LeakyReLU(0.2),
                        nn.Conv2d(128, 195, 3, 1, 1)).to(dev)
    x = torch.rand(8, 64, 128, 128, device=dev, requires_grad=True)

    def step():
        for p in net.parameters():
            p.grad = None
        x.grad = None
        net(x).square().mean().backward()
    print("eager conv fwd+bwd: %.3f ms" % timeit(step))
    if mode == "conv_graph":
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        print("graph conv fwd+bwd: %.3f ms" % timeit(g.replay))
