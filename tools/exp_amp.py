"""Experiment: the train step under bf16 autocast (conv stacks on bf16 MFMA; HIP ops stay fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import trainer, external_function as ef

torch.backends.cudnn.benchmark = False
dev = torch.device("cuda", 0)
t = trainer.FFWMTrainer(dev, seed=0)
batch = trainer.synthetic_batch(8, dev, seed=1)
t.pretrain_flow_identity(batch)


def run(n, amp):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                t._seg_forward_and_D(batch)
            t.red_D.finish()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                t._seg_stepD_and_G(batch)
            t.red_G.finish()
            t._seg_stepG()
        else:
            t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# the HIP ops are float32: cast at their boundary like torch.amp.custom_fwd(cast_inputs=float32) does
for name in ("WarpFunction", "GuidedFilterFunction"):
    fn = getattr(ef, name)
    orig = fn.apply

    def make(orig):
        def apply(*a):
            a = [x.float() if torch.is_tensor(x) and x.is_floating_point() else x for x in a]
            with torch.autocast("cuda", enabled=False):
                return orig(*a)
        return apply
    fn.apply = staticmethod(make(orig))

run(3, False)
print("fp32 ms/step", run(10, False))
try:
    run(3, True)
    print("bf16 autocast ms/step", run(10, True), {k: round(v, 4) for k, v in t.loss_values().items()})
except Exception as e:
    import traceback; traceback.print_exc()
