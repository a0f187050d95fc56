"""One-GPU diagnostics of the data-parallel capture modes (FFWMTrainer.capture): a ONE-rank process group with force_collectives=True
issues every collective for real.   python tools/dp_capture_probe.py <backend: gloo|nccl> <mode: segments|serial|ingraph|eager> [steps]"""
import faulthandler
import os
import sys

faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

backend, mode = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
RANK, WORLD = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))      # (torch.distributed.run: several gloo ranks share the GPU)
if backend == "nccl":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("gloo", rank=RANK, world_size=WORLD)
from ffwm_amd import trainer  # noqa: E402

if mode == "probe":
    print("probe_collective_capture:", trainer.probe_collective_capture(dev), flush=True)
    sys.exit(0)
t = trainer.FFWMTrainer(dev, world_size=WORLD, seed=40 + RANK, ngf=16, bucket_bytes=8 << 20, capturable=not mode.startswith("eager"), force_collectives=True,
                        segmented_backward=mode == "eager-segmented")
if mode == "eager-segmented":
    t.red_G.set_overlap(False)
batch = trainer.synthetic_batch(2, dev, seed=800 + RANK)
if not mode.startswith("eager"):
    print("capturing", mode, flush=True)
    t.capture(batch, warmup=2, mode=mode)
    print("captured", len(t._graphs), "graphs", flush=True)
for i in range(steps):
    bi = batch if i % 2 == 0 or i < 3 else trainer.synthetic_batch(2, dev, seed=900 + 10 * i)
    t.step(bi)
    torch.cuda.synchronize()
    v = t.loss_values()
    w = torch.cat([p.detach().flatten() for m in (t.flowNetF, t.flowNetB, t.netG, t.netD) for p in m.parameters()])
    fin = {g: all(bool(torch.isfinite(b["flat"]).all()) for b in t.red_G.buckets if b["group"] == g) for g in (0, 1, 2)}
    print("rank", RANK, i, {k: round(x, 4) for k, x in v.items()}, "weights finite:", bool(torch.isfinite(w).all()), "grads finite F/B/netG:", fin,
          "D:", bool(torch.isfinite(t.red_D.flat).all()), flush=True)
    for n, pp in t.netG.named_parameters():
        if n.endswith("blocks.3.bias") and pp.grad is not None and not bool(torch.isfinite(pp.grad).all()):
            bad = (~torch.isfinite(pp.grad)).nonzero().flatten()
            print("   ", n, "non-finite idx range", int(bad.min()), int(bad.max()), "count", bad.numel(), flush=True)
    if os.environ.get("FFWM_PROBE_LIST") == "1":
        print("   netG per-param non-finite:", [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in t.netG.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())], flush=True)
dist.destroy_process_group()
