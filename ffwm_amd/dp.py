"""Data parallelism for the FFWM train step: one process per GPU, gradients summed over RCCL/xGMI.

The reference has no working multi-GPU path; this is new.  Design for MI355X:
  * every parameter set that steps together (netD | netG + flowNetF + flowNetB) gets its own
    ``BucketedGradReducer``;
  * gradients live in a few LARGE flat fp32 buffers (``param.grad`` is a view into its bucket), so a
    bucket is reduced with ONE collective and no pack/unpack copies.  xGMI is point-to-point
    (7 links x ~153 GB/s per GPU): ring collectives are per-link bound, so buckets are big (default
    64 MiB) rather than DDP's 25 MiB NVSwitch-era default;
  * a bucket's all-reduce is launched from an autograd post-accumulate hook the moment its last
    gradient is written, i.e. it overlaps with the rest of backward; ``finish()`` waits for all of
    them (and launches any bucket whose parameters got no gradient this step);
  * parameters that never receive gradients (FlowNet's ``inter_conv_occ*``) are left out;
  * ``gather=True`` (the eager default on the GPU): ``param.grad`` is None during backward, so autograd hands every
    parameter its freshly produced gradient without a kernel, and a bucket's gradients are packed into the flat
    array by ONE multi-tensor copy when its last gradient arrives (world > 1) or in ``finish()`` (world 1) --
    instead of one ``grad += new`` launch per parameter (~600 tiny kernels per train step).  ``gather=False`` keeps
    the views installed as ``param.grad`` all the time (static addresses: what a captured hipGraph needs).
  * ``groups``: parameter lists that must not share a bucket (the trainer passes one per network).  A network's backward
    runs on ONE stream (autograd replays a node on its forward's stream), so a bucket never mixes gradients produced on
    different streams; should it happen anyway, the packing stream waits for every stream a gradient of the bucket arrived on.
    Per-network buckets are also what a segmented (captured) backward hands to ``launch_group`` segment by segment.
  * under a hipGraph capture the hooks still run (capture executes the Python once): with RCCL the hook-launched
    all-reduces are captured INTO the graph on RCCL's own stream, forked from and joined to the step's streams by the
    events torch.distributed records -- one graph then holds backward and its overlapped collectives (trainer.capture,
    mode "ingraph").
Works with any ``torch.distributed`` backend: ``nccl`` (= RCCL) on GPUs, ``gloo`` in the CPU tests.
"""
import torch
import torch.distributed as dist


def _pad4(n):
    return (n + 3) & ~3


class BucketedGradReducer(object):
    def __init__(self, params, process_group=None, bucket_bytes=64 << 20, average=True, gather=False, groups=None,
                 force_collectives=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        # force_collectives: issue the all-reduces even in a one-rank group (a one-rank RCCL all-reduce runs through the same
        # launch / capture machinery: the one-GPU box's test of capturing the collectives into the step's graph)
        self.active = self.world > 1 or (bool(force_collectives) and dist.is_available() and dist.is_initialized())
        self.average = average
        if groups is not None:
            groups_in = [[p for p in g if p.requires_grad] for g in groups]
            params = [p for g in groups_in for p in g]
            gid = {p: i for i, g in enumerate(groups_in) for p in g}
        else:
            params = [p for p in params if p.requires_grad]
            gid = {p: 0 for p in params}
        self._gid = gid
        # autograd produces gradients roughly in reverse registration order: fill buckets that way
        # so the first bucket completes early in backward
        ordered = list(reversed(params))
        self.buckets = []          # dicts: flat, params, pending, launched, handle
        self._bucket_of = {}
        self._view = {}            # param -> its slice of the bucket
        groups, cur, cur_bytes = [], [], 0
        for p in ordered:
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype or cur[0].device != p.device
                        or gid[cur[0]] != gid[p]):
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        # ONE flat gradient array for all buckets (a bucket is a slice of it) when dtype and device are uniform:
        # zero_grad is one memset, and a flat optimizer (ffwm_amd/optim.py:FlatAdam) can sweep a contiguous
        # range of parameters.  Every parameter starts on a 16-byte boundary (padding elements stay zero).
        self.flat = None
        self.offset = {}           # param -> (begin, end) in elements of self.flat
        uniform = bool(ordered) and all(p.dtype == ordered[0].dtype and p.device == ordered[0].device for p in ordered)
        if uniform:
            total = sum(_pad4(p.numel()) for p in ordered)
            self.flat = torch.zeros(total, dtype=ordered[0].dtype, device=ordered[0].device)
        base = 0
        for plist in groups:
            base = self._seal(plist, base)
        self.overlap = True
        self.gather = False
        self.launch_log = []       # (bucket index, "hook" | "finish") of the current step, in launch order
        self.set_gather(gather)
        self._hooks = []
        if self.active:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _seal(self, plist, base):
        total = sum(_pad4(p.numel()) for p in plist)
        if self.flat is not None:
            flat = self.flat[base:base + total]
        else:
            flat = torch.zeros(total, dtype=plist[0].dtype, device=plist[0].device)
        off = 0
        for p in plist:
            n = p.numel()
            view = flat[off:off + n].view_as(p)
            p.grad = view                             # autograd accumulates in place into the bucket
            self._view[p] = view
            self.offset[p] = (base + off, base + off + n)
            off += _pad4(n)
        b = {"flat": flat, "params": plist, "pending": len(plist), "launched": False, "handle": None, "index": len(self.buckets),
             "group": self._gid[plist[0]], "streams": []}
        for p in plist:
            self._bucket_of[p] = b
        self.buckets.append(b)
        return base + total

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """Replaces optimizer.zero_grad(): keeps the bucket views alive."""
        if self.flat is not None:
            self.flat.zero_()
        self.launch_log = []
        for b in self.buckets:
            if self.flat is None:
                b["flat"].zero_()
            b["pending"] = len(b["params"])
            b["launched"] = False
            b["packed"] = False
            b["handle"] = None
            b["streams"] = []
            if self.gather:
                for p in b["params"]:
                    p.grad = None             # autograd will install its own tensor: no accumulation kernel

    def begin_replay(self):
        """A replayed (captured) step runs no Python: zero_grad() inside the graph does not reset the host-side launch state.
        Call before the replays of a step whose all-reduces are issued from the host (between the graphs)."""
        self.launch_log = []
        for b in self.buckets:
            b["launched"] = False
            b["handle"] = None

    def set_gather(self, on):
        """Switch between packing the gradients after backward (True) and accumulating in place into the views."""
        self.gather = bool(on)
        for b in self.buckets:
            b["packed"] = False
            for p in b["params"]:
                p.grad = None if self.gather else self._view[p]

    def _pack(self, b):
        """gather mode: one multi-tensor copy of the bucket's fresh gradients into the flat array; the views become
        ``param.grad`` (what the optimizers read)."""
        if b.get("packed"):
            return
        b["packed"] = True
        if b["streams"] and b["flat"].is_cuda:
            # gradients of this bucket arrived on more than one stream (does not happen with per-network groups): the stream
            # that packs and reduces waits for the others' work issued so far -- all of the bucket's gradients are among it
            cur = torch.cuda.current_stream(b["flat"].device)
            for st in b["streams"]:
                if st != cur:
                    cur.wait_stream(st)
        src, dst = [], []
        for p in b["params"]:
            g, view = p.grad, self._view[p]
            if g is not None and g.data_ptr() != view.data_ptr():
                src.append(g if g.shape == view.shape else g.reshape(view.shape))
                dst.append(view)
            p.grad = view
        if src:
            torch._foreach_copy_(dst, src)

    def pack_all(self, group=None):
        """gather mode: pack every bucket (of parameter group `group`) now (a captured step calls it at the end of a backward
        segment, so that the copies are part of the graph and only the all-reduces are left outside the graphs)."""
        if self.gather:
            for b in self.buckets:
                if group is None or b["group"] == group:
                    self._pack(b)

    def launch_group(self, group, where="segment"):
        """Start the all-reduce of every bucket of parameter group `group` that has not gone out yet (asynchronous: the next
        backward segment is issued while it runs); finish() waits.  -> number of buckets launched."""
        n = 0
        if not self.active:
            return n
        for b in self.buckets:
            if b["group"] == group and not b["launched"]:
                if self.gather:
                    self._pack(b)
                self._launch(b, where)
                n += 1
        return n

    def _launch(self, b, where="hook"):
        b["launched"] = True
        # (bucket index, who launched it): what the dry-run tests assert the overlap on -- a bucket reduced from an autograd
        # hook went out DURING backward, one reduced from finish() did not
        self.launch_log.append((b["index"], where))
        b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def set_overlap(self, on):
        """on: buckets are reduced from the autograd hooks, overlapping with backward (default).
        off: nothing is launched from the hooks; finish() reduces every bucket (used when backward is
        replayed from a captured hipGraph, where no Python hook runs)."""
        self.overlap = bool(on)

    def _on_grad(self, p):
        if not self.overlap:
            return
        b = self._bucket_of[p]
        if b["flat"].is_cuda:
            st = torch.cuda.current_stream(b["flat"].device)
            if st not in b["streams"]:
                b["streams"].append(st)
        if self.gather:
            b["pending"] -= 1
            if b["pending"] == 0 and not b["launched"]:
                self._pack(b)
                self._launch(b)
            return
        view = self._view[p]
        if p.grad is not view and (p.grad is None or p.grad.data_ptr() != view.data_ptr()):
            # someone replaced .grad (e.g. optimizer.zero_grad(set_to_none=True)): fold it back
            if p.grad is not None:
                view.copy_(p.grad)
            p.grad = view
        b["pending"] -= 1
        if b["pending"] == 0 and not b["launched"]:
            self._launch(b)

    def wait_launched(self, host=False):
        """Wait for the all-reduces in flight (no averaging: finish() does that once).  host=True also blocks the host until the
        device has caught up -- nothing of a collective may still be running on some backend thread when a capture begins."""
        for b in self.buckets:
            if b["launched"] and b["handle"] is not None:
                b["handle"].wait()
        if host and self.buckets and self.buckets[0]["flat"].is_cuda:
            torch.cuda.synchronize(self.buckets[0]["flat"].device)

    def finish(self):
        """Block (stream-wise) until every bucket is reduced; call before optimizer.step()."""
        if self.gather:
            for b in self.buckets:
                if not b["launched"]:
                    self._pack(b)
        if not self.active:
            return
        for b in self.buckets:
            if not b["launched"]:
                self._launch(b, "finish")       # parameters without a gradient this step still take part
        for b in self.buckets:
            b["handle"].wait()
        if self.average and self.world > 1:
            if self.flat is not None:
                self.flat.div_(self.world)      # one launch for every bucket (padding elements stay zero)
            else:
                for b in self.buckets:
                    b["flat"].div_(self.world)

    def time_buckets(self, reps=3):
        """Stand-alone duration of every bucket's all-reduce (ms, max over `reps` excluded: the median), measured OUTSIDE a step with
        the device idle: what the collective costs when nothing overlaps it.  -> list of dicts (bucket MiB, ms, algorithm
        bandwidth and ring bus bandwidth 2 (n - 1) / n x bytes / time in GB/s).  The flat arrays are summed in place `reps` times:
        call it after the measurement, or zero_grad() afterwards."""
        out = []
        if not self.active:
            return out
        cuda = self.buckets[0]["flat"].is_cuda
        import time
        for i, b in enumerate(self.buckets):
            ts = []
            for _ in range(reps + 1):
                if cuda:
                    torch.cuda.synchronize()
                dist.barrier(group=self.group)
                t0 = time.perf_counter()
                dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
                if cuda:
                    torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[1:])
            t = ts[len(ts) // 2]
            nbytes = b["flat"].numel() * b["flat"].element_size()
            out.append({"bucket": i, "MiB": round(nbytes / 2 ** 20, 2), "ms": round(t * 1e3, 3),
                        "algbw_GBps": round(nbytes / t / 1e9, 1),
                        "busbw_GBps": round(2.0 * (self.world - 1) / self.world * nbytes / t / 1e9, 1)})
        return out

    def grad_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_module_state(modules, src=0, process_group=None):
    """Identical weights AND buffers on every rank at start-up (BN running stats, spectral-norm u/v).
    After that, buffers evolve per rank (per-GPU BatchNorm statistics, exactly what DDP does by
    default); gradients -- hence weights -- stay in lock-step through the reducers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=process_group)
