"""Margins of tests/test_gpu_parity.py::test_train_step_fast_paths_match_the_plain_pytorch_paths over repeated trials."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import trainer
DEV = "cuda:0"
for trial in range(3):
    batch = trainer.synthetic_batch(2, DEV, seed=5)
    plain = trainer.FFWMTrainer(DEV, seed=1, mfma_wgrad=False, flat_adam=False, fused_bn=False, fused_spectral_norm=False,
                                batched_losses=False, capturable=False)
    plain.red_G.set_gather(False); plain.red_D.set_gather(False)
    fast = trainer.FFWMTrainer(DEV, seed=1)
    lp, lf = plain.step(batch), fast.step(batch)
    torch.cuda.synchronize()
    worst = max(abs(float(lp[k]) - float(lf[k])) / (1 + abs(float(lp[k]))) for k in lp)
    out = []
    for net in ("netG", "netD"):
        pp = torch.cat([p.detach().flatten() for p in getattr(plain, net).parameters()])
        pf = torch.cat([p.detach().flatten() for p in getattr(fast, net).parameters()])
        out.append(((pp - pf).abs() <= 1e-4).float().mean().item())
    print("trial", trial, "worst loss rel diff %.2e" % worst, "agree", out)
    del plain, fast
