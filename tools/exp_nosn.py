"""Experiment: train-step time with spectral norm removed (upper bound of what a fused SN kernel can give)."""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import trainer
from ffwm_amd.dp import BucketedGradReducer

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


run(3)
print("with SN ms/step", run(8))
n = 0
for net in (t.netG, t.netD):
    for m in net.modules():
        try:
            torch.nn.utils.remove_spectral_norm(m)
            n += 1
        except ValueError:
            pass
print("removed SN from", n, "layers")
flow_params = [p for net in (t.flowNetF, t.flowNetB) for nm, p in net.named_parameters() if not nm.startswith("inter_conv_occ")]
t.opt_G = torch.optim.Adam(t.netG.parameters(), lr=0.0004, betas=(0.5, 0.999))
t.opt_D = torch.optim.Adam(t.netD.parameters(), lr=0.0004, betas=(0.5, 0.999))
t.red_G = BucketedGradReducer(itertools.chain(flow_params, t.netG.parameters()))
t.red_D = BucketedGradReducer(t.netD.parameters())
run(3)
print("without SN ms/step", run(8))
