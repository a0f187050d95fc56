"""Experiment: how much of the train step is the bias add (+ bias gradient) of convs that feed a BatchNorm?
In training mode BN removes a per-channel constant exactly, so those adds are dead work."""
import os, sys, time
import torch
import torch.nn as nn
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import _lib, trainer


def strip(net):
    n = 0
    for m in net.modules():
        if isinstance(m, nn.Sequential):
            mods = list(m.children())
            for a, b in zip(mods, mods[1:]):
                if isinstance(a, (nn.Conv2d, nn.ConvTranspose2d)) and isinstance(b, nn.BatchNorm2d) and a.bias is not None:
                    a.bias = None
                    n += 1
    return n


def run(do_strip):
    dev = torch.device("cuda", 0)
    t = trainer.FFWMTrainer(dev, world_size=1, seed=0)
    n = 0
    if do_strip:
        for net in (t.netG, t.flowNetF, t.flowNetB, t.netD):
            n += strip(net)
    batch = trainer.synthetic_batch(8, dev, seed=1)
    for _ in range(3):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    print("strip=%s convs=%d  %.2f ms/step" % (do_strip, n, (time.perf_counter() - t0) / 8 * 1e3))


_lib.load()
run(False)
run(True)
