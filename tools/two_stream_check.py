"""FFWM_STREAMS: flowNetB and the loss networks' side passes on their own streams must not change a single number.  The optimizers' learning rates are set to ZERO,
so every step (eager or replayed from the captured graph) repeats the same computation from the same weights: the losses of all
steps of all runs must agree to the noise of the float atomics (~1e-6 relative), with one stream and with two.  A race shows up as
a run or a step that does not."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import trainer

dev = torch.device("cuda", 0)
batch = trainer.synthetic_batch(8, dev, seed=1)


def run(two, graph, steps=4):
    os.environ["FFWM_STREAMS"] = "1" if two else "0"
    torch.manual_seed(0)
    t = trainer.FFWMTrainer(dev, seed=0, capturable=graph)
    assert (t.flow_stream is not None) == two
    t.pretrain_flow_identity(batch, steps=20)            # realistic (smooth) flows; its own optimizer -- before the comparison starts
    w = torch.cat([p.detach().flatten() for p in list(t.flowNetB.parameters()) + list(t.flowNetF.parameters())]).clone()
    for o in (t.opt_F, t.opt_G, t.opt_D):
        o.lr = 0.0
    if graph:
        t.capture(batch, warmup=2)
    out = []
    for _ in range(steps):
        t.step(batch, batch_increment=0)
        torch.cuda.synchronize()
        out.append({k: float(v.detach()) for k, v in t.losses.items()})
    return w, out


ok = True
for graph in (False, True):
    w0, ref = run(False, graph)
    base = ref[0]
    worst1 = max(abs(s[k] - base[k]) / (1e-6 + abs(base[k])) for s in ref for k in base)
    print("%s, one stream : %s   step-to-step spread %.2e" % ("graph" if graph else "eager", {k: round(v, 5) for k, v in base.items()}, worst1))
    for i in range(4):
        w1, got = run(True, graph)
        # the pre-fit itself is chaotic across runs (its own Adam steps): compare only runs that start from the same flow-net weights
        same = (w1 - w0).abs().max().item()
        spread = max(abs(s[k] - got[0][k]) / (1e-6 + abs(got[0][k])) for s in got for k in got[0])
        vs1 = max(abs(got[0][k] - base[k]) / (1e-6 + abs(base[k])) for k in base)
        print("   two streams #%d: step-to-step spread %.2e; vs one stream %.2e (start weights differ by %.1e)" % (i, spread, vs1, same))
        ok = ok and spread <= max(5e-5, 5 * worst1)
print("RESULT", "ok" if ok else "RACE?")
