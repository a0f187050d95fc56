"""Count the convolution / linear FLOPs a step REALLY executes (forward + the backward passes autograd will run).

`bench.py` quotes fp32_flop_frac from THIS count, not from the reference's 444 GFLOP per image: this
trainer skips work the reference does (no weight gradients for the frozen LightCNN / VGG19, target-side extractors
under no_grad, the two LightCNN passes of a step deduplicated), so pricing its img/s with the reference's FLOPs would
overstate the achieved TFLOP/s.

Forward hooks on every nn.Conv2d / nn.ConvTranspose2d / nn.Linear: 2 * MACs per call; when grad mode is on, another
2 * MACs if the weight requires grad (weight gradient) and another 2 * MACs if the input requires grad (data gradient).
Element-wise work, normalisations, the warp kernels and the optimizers are not counted (they are priced in bytes).
"""
import torch
import torch.nn as nn


class FlopCounter(object):
    def __init__(self, modules):
        self.fwd = 0.0
        self.bwd = 0.0
        self.calls = 0
        self._hooks = []
        seen = set()
        for root in modules:
            for m in root.modules():
                if id(m) in seen:
                    continue
                seen.add(id(m))
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                    self._hooks.append(m.register_forward_hook(self._hook))

    def _hook(self, m, inputs, out):
        x = inputs[0]
        if isinstance(m, nn.Linear):
            macs = out.numel() * m.in_features
        elif isinstance(m, nn.ConvTranspose2d):
            kh, kw = m.kernel_size
            macs = x.numel() * (m.out_channels // m.groups) * kh * kw          # every input element meets C_out * k * k weights
        else:
            kh, kw = m.kernel_size
            macs = out.numel() * (m.in_channels // m.groups) * kh * kw
        self.calls += 1
        self.fwd += 2.0 * macs
        if torch.is_grad_enabled():
            w = getattr(m, "weight_orig", m.weight)                             # spectral norm keeps the Parameter as weight_orig
            if w.requires_grad:
                self.bwd += 2.0 * macs
            if x.requires_grad:
                self.bwd += 2.0 * macs

    @property
    def total(self):
        return self.fwd + self.bwd

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def count_step(modules, step_fn):
    """FLOPs of one call of `step_fn` over the conv / linear layers of `modules` -> dict(fwd, bwd, total, calls)."""
    c = FlopCounter(modules)
    try:
        step_fn()
    finally:
        c.remove()
    return {"fwd": c.fwd, "bwd": c.bwd, "total": c.total, "calls": c.calls}
