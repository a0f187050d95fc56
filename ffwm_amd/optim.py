"""FlatAdam: torch.optim.Adam's update (models/ffwm_model.py:46-49, models/flownet_model.py:33: no weight decay, no
amsgrad) as ONE streaming kernel per step over flat arrays (csrc/adam.hip).

The gradients already live in the data-parallel reducer's flat array (dp.BucketedGradReducer.flat); this class
moves the parameters of one optimizer into a flat array laid out the same way (``param.data`` becomes a view of
it -- the Parameter objects, their names and state dicts are untouched) and keeps both moment estimates flat.
Like the reference's checkpoints (models/base_model.py:save_networks), optimizer state is not saved.
"""
import torch

from . import _lib


class FlatAdam(object):
    def __init__(self, params, reducer, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        params = [p for p in params if p.requires_grad]
        if reducer.flat is None or not params:
            raise ValueError("FlatAdam needs a reducer with one flat gradient array and at least one parameter")
        spans = sorted(reducer.offset[p] for p in params)
        self.lo, self.hi = spans[0][0], spans[-1][1]
        inside = [p for p, (a, b) in reducer.offset.items() if a >= self.lo and b <= self.hi]
        if len(inside) != len(params) or set(map(id, inside)) != set(map(id, params)):
            raise ValueError("FlatAdam: the parameters are not a contiguous range of the reducer's flat layout")
        if self.lo % 4:
            raise ValueError("FlatAdam: range start is not 16-byte aligned")
        self.reducer = reducer
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        n = self.hi - self.lo
        dev, dt = reducer.flat.device, reducer.flat.dtype
        if dt != torch.float32 or dev.type != "cuda":
            raise NotImplementedError("FlatAdam runs on float32 CUDA parameters only (csrc/adam.hip)")
        self.params = torch.zeros(n, device=dev, dtype=dt)
        self.exp_avg = torch.zeros(n, device=dev, dtype=dt)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=dt)
        with torch.no_grad():
            for p in params:
                a, b = reducer.offset[p]
                view = self.params[a - self.lo:b - self.lo].view_as(p)
                view.copy_(p.data)
                p.data = view
                p._ffwm_flat_adam = True       # updated by a raw-pointer kernel: the version counter does not see it (conv.frozen_cache)
        self.n = n
        self.steps = 0
        # capturable: the step counter lives on the device (ffwm_adam_step_device), so that step() can sit inside a captured hipGraph
        self.capturable = bool(capturable)
        # state[0] step counter, state[1..2] scratch, state[3] the current learning rate (csrc/adam.hip: it overrides the kernel argument
        # a captured graph has baked in)
        self.state = torch.zeros(4, device=dev, dtype=torch.float64) if self.capturable else None
        if self.state is not None:
            self.state[3] = -1.0          # no learning-rate override yet (0.0 is a legitimate rate: the sentinel is negative)
        # torch.optim's interface for schedulers (base_model.update_learning_rate: StepLR / lambda rules write param_groups[i]['lr'])
        self.param_groups = [{"capturable": self.capturable, "lr": self.lr, "params": params}]
        self._lr_on_device = None
        self._lr_seen = self.lr

    def sync_lr(self):
        """The learning rate of the next step: whichever of ``self.lr`` (direct assignment) and ``param_groups[0]['lr']`` (torch.optim's
        interface: base_model.update_learning_rate, StepLR / lambda rules) CHANGED since the last call -- the scheduler's value when both
        did.  Eager: used by the next step() directly.  Capturable: written to the device state, so it also reaches the REPLAYS of a
        captured step -- call it (or step()) from the host after the scheduler ran; a replay itself runs no Python.  A rate of 0.0 is a
        rate like any other (csrc/adam.hip takes the device value whenever it is >= 0; the state starts at -1 = "no override")."""
        g = float(self.param_groups[0]["lr"])
        if g != self._lr_seen:
            lr = g
        elif float(self.lr) != self._lr_seen:
            lr = float(self.lr)
        else:
            lr = self._lr_seen
        self.lr = self._lr_seen = lr
        self.param_groups[0]["lr"] = lr
        if self.capturable and lr != self._lr_on_device and not torch.cuda.is_current_stream_capturing():
            self.state[3:4].fill_(lr)
            self._lr_on_device = lr
        return lr

    def step(self):
        self.sync_lr()
        self.steps += 1
        g = self.reducer.flat[self.lo:self.hi]
        dev = self.params.device.index
        cur = torch.cuda.current_device()
        if cur != dev:
            torch.cuda.set_device(dev)
        try:
            if self.capturable:
                _lib.check(_lib.load().ffwm_adam_step_device(self.params.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                                             self.exp_avg_sq.data_ptr(), self.n, self.lr, self.betas[0], self.betas[1],
                                                             self.eps, self.state.data_ptr(), _lib.F32,
                                                             torch.cuda.current_stream(dev).cuda_stream), "ffwm_adam_step_device")
            else:
                _lib.check(_lib.load().ffwm_adam_step(self.params.data_ptr(), g.data_ptr(), self.exp_avg.data_ptr(),
                                                      self.exp_avg_sq.data_ptr(), self.n, self.lr, self.betas[0], self.betas[1],
                                                      self.eps, self.steps, _lib.F32, torch.cuda.current_stream(dev).cuda_stream),
                           "ffwm_adam_step")
        finally:
            if cur != dev:
                torch.cuda.set_device(cur)

    def zero_grad(self, set_to_none=False):      # the reducer owns the gradients
        self.reducer.flat[self.lo:self.hi].zero_()
