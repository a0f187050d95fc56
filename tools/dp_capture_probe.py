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

TRACE = []
if os.environ.get("FFWM_PROBE_TRACE") == "1":
    # snapshots around every sliced call of the tiled weight-gradient kernel (captured copy nodes: they replay with the graph), so that
    # after a replay one can tell a non-finite RESULT of finite operands (the kernel / its zero-fill) from non-finite OPERANDS (upstream)
    from ffwm_amd import ops as _ops
    _orig = _ops.conv2d_wgrad_tiled

    ARENA = {}          # FFWM_PROBE_ARENA=1: the result buffers of the traced calls are static tensors outside the graphs' pool

    def _arena_call(rows, gathered, kernel, stride, pad, want_bias, slot):
        from ffwm_amd import _lib as L
        B, K, Ho, Wo = rows.shape
        _, C, H, W = gathered.shape
        n = K * C * kernel * kernel
        buf = ARENA[slot]
        gw, gb = buf[:n].view(K, C, kernel, kernel), (buf[n:n + K] if want_bias else None)
        L.check(L.load().ffwm_conv2d_wgrad_tiled(rows.data_ptr(), gathered.data_ptr(), gw.data_ptr(), gb.data_ptr() if gb is not None else None,
                                                  B, K, Ho, Wo, C, H, W, kernel, stride, pad, 0, 0, torch.cuda.current_stream().cuda_stream), "wgrad")
        return gw, gb

    def _traced(rows, gathered, kernel, stride, pad, want_bias=False):
        take = rows.shape[1] >= 256 and rows.shape[2] == 32 and kernel == 3
        pre = (rows.clone(), gathered.clone()) if take else None
        if take and os.environ.get("FFWM_PROBE_ARENA") == "1" and torch.cuda.is_current_stream_capturing():
            gw, gb = _arena_call(rows, gathered, kernel, stride, pad, want_bias, len(TRACE) % 8)
        else:
            gw, gb = _orig(rows, gathered, kernel, stride, pad, want_bias=want_bias)
        if take:
            TRACE.append({"rows": pre[0], "gathered": pre[1], "rows_after": rows.clone(), "gathered_after": gathered.clone(), "gw": gw.clone(),
                          "gb": gb.clone() if gb is not None else None, "stream": torch.cuda.current_stream().cuda_stream,
                          "capturing": torch.cuda.is_current_stream_capturing(), "shape": (tuple(rows.shape), tuple(gathered.shape), stride, pad)})
        return gw, gb
    _ops.conv2d_wgrad_tiled = _traced


def report_trace(tag):
    for i, t in enumerate(TRACE):
        if not t["capturing"]:
            continue
        fin = {k: bool(torch.isfinite(t[k]).all()) for k in ("rows", "gathered", "rows_after", "gathered_after", "gw", "gb") if t[k] is not None}
        same = bool(torch.equal(t["rows"], t["rows_after"])) and bool(torch.equal(t["gathered"], t["gathered_after"]))
        ref = torch.ops.aten.convolution_backward(t["rows"].double(), t["gathered"].double(), t["gw"].double(), None, [t["shape"][2]] * 2, [t["shape"][3]] * 2,
                                                  [1, 1], False, [0, 0], 1, [False, True, False])[1]
        err = float((t["gw"].double() - ref).abs().max()) if fin["gw"] and fin["rows"] and fin["gathered"] else float("nan")
        nbad = int((~torch.isfinite(t["gw"])).sum())
        K = t["gw"].shape[0]
        g2 = t["gw"].reshape(K, -1).double()
        wrong = (~torch.isfinite(g2)) | ((g2 - ref.reshape(K, -1)).abs() > 1e-3 * (1 + ref.abs().max()))
        rows_bad = wrong.any(1).nonzero().flatten().tolist()
        cols_bad = wrong.any(0).nonzero().flatten().tolist()
        def runs(v):
            out, i = [], 0
            while i < len(v):
                j = i
                while j + 1 < len(v) and v[j + 1] == v[j] + 1:
                    j += 1
                out.append((v[i], v[j]))
                i = j + 1
            return out[:8]
        print("      wrong elements %d; rows (k) runs %s; cols (n = c * 9 + rs) runs %s" % (int(wrong.sum()), runs(rows_bad), runs(cols_bad)), flush=True)
        print("    trace", tag, i, t["shape"][0], "stream %x" % t["stream"], "finite:", fin, "operands unchanged over the call:", same,
              "gw non-finite elements: %d of %d" % (nbad, t["gw"].numel()), "max |gw - fp64 of the snapshot operands| = %.3g" % err, flush=True)


if mode == "probe":
    print("probe_collective_capture:", trainer.probe_collective_capture(dev), flush=True)
    sys.exit(0)
t = trainer.FFWMTrainer(dev, world_size=WORLD, seed=40 + RANK, ngf=16, bucket_bytes=8 << 20, capturable=not mode.startswith("eager"), force_collectives=True,
                        segmented_backward=mode == "eager-segmented")
if mode == "eager-segmented":
    t.red_G.set_overlap(False)
batch = trainer.synthetic_batch(2, dev, seed=800 + RANK)
if os.environ.get("FFWM_PROBE_ARENA") == "1":
    for _i in range(8):
        ARENA[_i] = torch.full((384 * 384 * 9 + 384,), float("nan"), device=dev)
if os.environ.get("FFWM_PROBE_NOCOLL") == "1":
    # the collectives replaced by nothing (a one-rank sum is the identity): is a backend thread's traffic part of the problem?
    class _Done(object):
        def wait(self, *a, **k):
            return True

        def is_completed(self):
            return True
    dist.all_reduce = lambda *a, **k: _Done()
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
    if TRACE and i < 3:
        report_trace("step %d" % i)
    if os.environ.get("FFWM_PROBE_LIST") == "1":
        print("   netG per-param non-finite:", [(n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()) for n, p in t.netG.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())], flush=True)
dist.destroy_process_group()
