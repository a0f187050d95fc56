"""Per-layer convolution cost of the train step (torch profiler, grouped by input shape).

    python tools/exp_conv_shapes.py > gpurun_out/conv_shapes.txt

Prints, for every distinct (op, input shapes) of aten::convolution / convolution_backward /
miopen_* in ONE steady-state train step, the call count, the device time and the TFLOP/s of the
dense contraction, so that the layers MIOpen serves badly can be named.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from ffwm_amd import _lib, trainer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _lib.load()
    t = trainer.FFWMTrainer(dev, world_size=1, seed=0)
    batch = trainer.synthetic_batch(8, dev, seed=1)
    t.pretrain_flow_identity(batch, steps=10)
    for _ in range(3):
        t.step(batch, batch_increment=0)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        t.step(batch, batch_increment=0)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        if "conv" not in e.key.lower():
            continue
        dt = getattr(e, "device_time_total", None)
        if dt is None:
            dt = e.cuda_time_total
        rows.append((dt, e.key, e.count, str(e.input_shapes)))
    rows.sort(reverse=True)
    for dt, key, n, shapes in rows[:90]:
        print("%10.1f us  x%-3d %-40s %s" % (dt, n, key, shapes))
    # kernel-level attribution: which kernels serve which conv op
    agg = {}
    for ev in prof.events():
        if ev.name not in ("aten::convolution_backward", "aten::miopen_convolution", "aten::miopen_convolution_transpose",
                           "aten::cudnn_convolution"):
            continue
        key = (ev.name, str(ev.input_shapes)[:150])
        d = agg.setdefault(key, {})
        for k in ev.kernels:
            c = d.setdefault(k.name[:70], [0, 0.0])
            c[0] += 1
            c[1] += k.duration
    out = []
    for key, d in agg.items():
        out.append((sum(v[1] for v in d.values()), key, d))
    out.sort(key=lambda r: -r[0])
    print("---- kernels per conv op")
    for tot, key, d in out[:80]:
        print("%9.1f us %s %s" % (tot, key[0], key[1]))
        for name, (n, dur) in sorted(d.items(), key=lambda kv: -kv[1][1]):
            print("        %9.1f us x%-3d %s" % (dur, n, name))


if __name__ == "__main__":
    main()
