"""Every conv / linear call of one FFWM train step (CPU run of the trainer): network, layer, shapes, which gradients autograd
will ask for, GFLOP per direction.  Used to decide the routing of the hand-written kernels (no GPU needed)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from ffwm_amd import trainer
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch_refs

def main(bs=8):
    torch.set_num_threads(8)
    t = trainer.FFWMTrainer("cpu", seed=0, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat)
    b = trainer.synthetic_batch(bs, "cpu", seed=1)
    calls = []
    names = {}
    for nn_name in ("flowNetF", "flowNetB", "netG", "netD", "lightCNN", "vgg"):
        for n, m in getattr(t, nn_name).named_modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                names[id(m)] = nn_name + "." + n
                def hook(m, inp, out, key=nn_name + "." + n):
                    x = inp[0]
                    w = getattr(m, "weight_orig", m.weight)
                    g = torch.is_grad_enabled()
                    calls.append((key, type(m).__name__, tuple(x.shape), tuple(out.shape),
                                  getattr(m, "kernel_size", None), getattr(m, "stride", None), getattr(m, "padding", None), getattr(m, "dilation", None),
                                  bool(g and x.requires_grad), bool(g and w.requires_grad)))
                m.register_forward_hook(hook)
    t.step(b)
    agg = collections.OrderedDict()
    for c in calls:
        k = c[1:]  # by shape signature
        a = agg.setdefault(k, [0, []])
        a[0] += 1
        a[1].append(c[0])
    print("%d calls, %d distinct signatures" % (len(calls), len(agg)))
    for k, (n, who) in agg.items():
        typ, xs, os_, ks, st, pd, dl, dg, wg = k
        if typ == "Linear":
            macs = os_[0] * os_[1] * xs[1]
        elif typ == "ConvTranspose2d":
            macs = xs[0] * xs[1] * xs[2] * xs[3] * os_[1] * ks[0] * ks[1]
        else:
            macs = os_[0] * os_[1] * os_[2] * os_[3] * xs[1] * ks[0] * ks[1]
        print("%3dx %-15s x%-20s -> %-20s k%s s%s p%s d%s dgrad=%d wgrad=%d %7.3f GFLOP/dir  %s" % (
            n, typ, xs, os_, ks and ks[0], st and st[0], pd and pd[0], dl and dl[0], dg, wg, 2 * macs / 1e9, ",".join(sorted(set(who)))[:110]))

if __name__ == "__main__":
    main()
