"""Stand-alone stress of the tiled MFMA weight-gradient kernel (csrc/conv_bwd.hip, conv_wgrad_tile_kernel) under the conditions in
which round 4 saw non-finite weight gradients: sliced launches (zero-fill + float atomics) of the 256 -> 256 and 384 -> 384 3x3 layers
at 32 x 32, issued concurrently on several streams inside several hipGraphs that share one memory pool, replayed many times, beside a
FlowNet forward + backward on a side stream, with the pool's blocks NaN-poisoned before the capture.

    python tools/wgrad_stress.py [replays=200] [graphs=3] [streams=3] [batch=2]

Every replay's results are compared with the float64 weight gradient of the same operands (the operands change between replays:
static input buffers refilled from a seeded generator); prints one line per configuration and exits non-zero on any mismatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ffwm_amd import nets, ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(256, 256, 32), (384, 384, 32), (256, 384, 32), (128, 128, 64)]          # (K, C, plane)


def fp64_wgrad(rows, gathered):
    K, C = rows.shape[1], gathered.shape[1]
    w = torch.zeros(K, C, 3, 3, device=rows.device, dtype=torch.float64)
    return torch.ops.aten.convolution_backward(rows.double(), gathered.double(), w, [K], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, True])[1:]


def poison_pool(pool, streams):
    """NaN-fill blocks of the graphs' private pool on every stream the captures will allocate on (a pool's free blocks are kept per
    stream), then free them: the captures' allocations are served from poisoned memory."""
    for st in streams:
        with torch.cuda.stream(st), torch.cuda.use_mem_pool(pool):
            bufs = [torch.full((n,), float("nan"), device=dev) for n in (1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 1 << 14) for _ in range(6)]
            del bufs
    torch.cuda.synchronize()


def run(replays, n_graphs, n_streams, batch, flownet_beside=True, poison=True, seed=0):
    torch.manual_seed(seed)
    main = torch.cuda.Stream(dev)
    sides = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    flow_stream = torch.cuda.Stream(dev)
    pool = torch.cuda.MemPool()
    # static operands, one set per (graph, stream)
    ops_in, outs = [], []
    for gi in range(n_graphs):
        for si in range(n_streams):
            K, C, P = SHAPES[(gi * n_streams + si) % len(SHAPES)]
            ops_in.append((torch.empty(batch, K, P, P, device=dev), torch.empty(batch, C, P, P, device=dev)))
    fnet = nets.FlowNet(16).to(dev).train() if flownet_beside else None
    fimg = torch.rand(2, 3, 128, 128, device=dev)
    if fnet is not None:
        from ffwm_amd import conv
        conv.route_conv_winograd(fnet)          # the flow nets' routes of the trainer (FlowNet has no spectral norm: not route_training_kernels)
        conv.route_conv_fwd(fnet)
        conv.route_conv_bwd(fnet)
        for _ in range(2):          # warm-up outside the capture (solver selection)
            sum(o.square().mean() for o in fnet(fimg)).backward()
        fnet.zero_grad(set_to_none=True)
    for r, g in ops_in:             # warm-up of the kernel (LDS attribute, code object load)
        r.normal_(); g.normal_()
        ops.conv2d_wgrad_tiled(r, g, 3, 1, 1, want_bias=True)
    torch.cuda.synchronize()
    if poison:
        poison_pool(pool, [main, flow_stream] + sides)
    graphs = []
    for gi in range(n_graphs):
        g = torch.cuda.CUDAGraph()
        res = []
        with torch.cuda.graph(g, pool=pool.id, stream=main):
            if fnet is not None and gi % 2 == 0:
                flow_stream.wait_stream(main)
                with torch.cuda.stream(flow_stream):
                    fl = sum(o.square().mean() for o in fnet(fimg))
                    fl.backward()
            for si, st in enumerate(sides):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    r, x = ops_in[gi * n_streams + si]
                    # the result buffers are fresh allocations of the shared pool inside the capture, as in the train step; a second
                    # call right behind reuses what the first one's temporaries freed
                    gw, gb = ops.conv2d_wgrad_tiled(r, x, 3, 1, 1, want_bias=True)
                    gw2, gb2 = ops.conv2d_wgrad_tiled(r, x, 3, 1, 1, want_bias=True)
                    res.append((gw.clone(), gb.clone(), gw2.clone(), gb2.clone()))
                    del gw, gb, gw2, gb2
            for st in sides:
                main.wait_stream(st)
            if fnet is not None and gi % 2 == 0:
                main.wait_stream(flow_stream)
        graphs.append(g)
        outs.append(res)
    gen = torch.Generator(device=dev).manual_seed(seed + 1)
    worst, bad = 0.0, 0
    with torch.cuda.stream(main):
        for it in range(replays):
            for r, x in ops_in:
                r.normal_(generator=gen)
                x.normal_(generator=gen)
            for g in graphs:
                g.replay()
            if it < 5 or it % 10 == 0 or it == replays - 1:          # the float64 check costs more than the replay
                main.synchronize()
                for gi in range(n_graphs):
                    for si in range(n_streams):
                        r, x = ops_in[gi * n_streams + si]
                        rw, rb = fp64_wgrad(r, x)
                        for gw, gb in (outs[gi][si][:2], outs[gi][si][2:]):
                            if not (bool(torch.isfinite(gw).all()) and bool(torch.isfinite(gb).all())):
                                bad += 1
                                print("   replay %d graph %d stream %d: non-finite (gw %d, gb %d elements)" % (
                                    it, gi, si, int((~torch.isfinite(gw)).sum()), int((~torch.isfinite(gb)).sum())), flush=True)
                                continue
                            e = max(float((gw.double() - rw).abs().max() / (1 + rw.abs().max())), float((gb.double() - rb).abs().max() / (1 + rb.abs().max())))
                            worst = max(worst, e)
                            if e > 1e-4:
                                bad += 1
                                print("   replay %d graph %d stream %d: error %.3g" % (it, gi, si, e), flush=True)
            else:
                # cheap check on the other replays: finite sums
                s = torch.stack([t.sum() for res in outs for tup in res for t in tup])
                if not bool(torch.isfinite(s).all()):
                    bad += 1
                    print("   replay %d: a non-finite sum" % it, flush=True)
    torch.cuda.synchronize()
    print("wgrad_stress: %d replays x %d graphs x %d streams, batch %d, FlowNet beside %s, poisoned pool %s: %s (worst error %.3g of the scale)"
          % (replays, n_graphs, n_streams, batch, flownet_beside, poison, "OK" if bad == 0 else "%d FAILURES" % bad, worst), flush=True)
    return bad


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:]]
    replays = a[0] if len(a) > 0 else 200
    ng = a[1] if len(a) > 1 else 3
    ns = a[2] if len(a) > 2 else 3
    batch = a[3] if len(a) > 3 else 2
    failures = 0
    failures += run(replays, ng, ns, batch)
    failures += run(max(20, replays // 4), 5, 4, 8, flownet_beside=True)
    failures += run(max(20, replays // 4), 2, 2, 2, flownet_beside=False)
    sys.exit(1 if failures else 0)
