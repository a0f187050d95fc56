"""world_size-2 CPU (gloo) tests of the data-parallel path: ffwm_amd.dp.BucketedGradReducer must give
every rank the average of the per-rank gradients (== the gradient of the global-batch mean loss),
with buckets launched from backward hooks, parameters without gradients included, and the GAN-style
alternation of two reducers with requires_grad toggling (ffwm_model.py:151-160)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy(seed):
    torch.manual_seed(seed)
    g = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.LeakyReLU(0.2), nn.Conv2d(8, 3, 3, padding=1))
    d = nn.Sequential(nn.Conv2d(3, 4, 3, stride=2, padding=1), nn.LeakyReLU(0.2), nn.Conv2d(4, 1, 1))
    unused = nn.Conv2d(3, 3, 1)       # never part of forward, like FlowNet.inter_conv_occ*
    return g, d, unused


def _gan_step(g, d, red_g, red_d, x, y):
    """D step then G step with the reference's requires_grad toggling."""
    fake = g(x)
    for p in d.parameters():
        p.requires_grad = True
    red_d.zero_grad()
    loss_d = 0.5 * ((d(fake.detach()) ** 2).mean() + ((d(y) - 1) ** 2).mean())
    loss_d.backward()
    red_d.finish()
    for p in d.parameters():
        p.requires_grad = False
    red_g.zero_grad()
    loss_g = (fake - y).abs().mean() + 0.1 * ((d(fake) - 1) ** 2).mean()
    loss_g.backward()
    red_g.finish()


def _worker(rank, world, port, ret, gather=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd.dp import BucketedGradReducer, broadcast_module_state
        g, d, unused = _toy(100 + rank)               # different init per rank on purpose
        broadcast_module_state([g, d, unused])
        ref_g, ref_d, _ = _toy(100)                   # rank 0's init
        for a, b in zip(list(g.parameters()) + list(d.parameters()),
                        list(ref_g.parameters()) + list(ref_d.parameters())):
            assert torch.equal(a, b), "broadcast did not synchronise the weights"
        gparams = list(g.parameters()) + list(unused.parameters())
        red_g = BucketedGradReducer(gparams, bucket_bytes=512, gather=gather)       # tiny buckets: several per set
        red_d = BucketedGradReducer(d.parameters(), bucket_bytes=1 << 20, gather=gather)
        assert len(red_g.buckets) > 1 and len(red_d.buckets) == 1
        gen = torch.Generator().manual_seed(7)
        xs = torch.rand(world, 2, 3, 8, 8, generator=gen)
        ys = torch.rand(world, 2, 3, 8, 8, generator=gen)
        _gan_step(g, d, red_g, red_d, xs[rank], ys[rank])
        # hooks must have launched the G buckets that got gradients during backward
        grads = {"g": [p.grad.clone() for p in g.parameters()], "d": [p.grad.clone() for p in d.parameters()],
                 "unused": [p.grad.clone() for p in unused.parameters()]}
        # single-process reference on the global batch
        from ffwm_amd.dp import BucketedGradReducer as R   # world-size-agnostic API
        fake_all = ref_g(xs.flatten(0, 1))
        y_all = ys.flatten(0, 1)
        loss_d = 0.5 * ((ref_d(fake_all.detach()) ** 2).mean() + ((ref_d(y_all) - 1) ** 2).mean())
        gd = torch.autograd.grad(loss_d, list(ref_d.parameters()))
        for p in ref_d.parameters():
            p.requires_grad = False
        loss_g = (fake_all - y_all).abs().mean() + 0.1 * ((ref_d(fake_all) - 1) ** 2).mean()
        gg = torch.autograd.grad(loss_g, list(ref_g.parameters()))
        for a, b in zip(grads["g"], gg):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
        for a, b in zip(grads["d"], gd):
            assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
        for a in grads["unused"]:
            assert float(a.abs().max()) == 0.0
        # a second step keeps the bucket views alive (zero_grad does not detach them)
        _gan_step(g, d, red_g, red_d, xs[rank], ys[rank])
        for p in g.parameters():
            assert p.grad.data_ptr() == red_g._view[p].data_ptr()
        # optimizer.zero_grad(set_to_none=True) by a careless caller is folded back by the hook
        for p in g.parameters():
            p.grad = None
        red_g.zero_grad()
        (g(xs[rank]) - ys[rank]).abs().mean().backward()
        red_g.finish()
        flat = torch.cat([p.grad.flatten() for p in g.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert torch.equal(gathered[0], gathered[1]), "ranks disagree after all-reduce"
        ret[rank] = "ok"
    except Exception as e:   # surface the failure in the parent
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather", [False, True])
def test_bucketed_reducer_world2_gloo(gather):
    """gather=False: gradients accumulate in place into the bucket views; gather=True: autograd installs fresh
    gradients and a bucket is packed by one multi-tensor copy when its last gradient arrives."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, gather), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_reducer_is_a_noop_without_process_group():
    from ffwm_amd.dp import BucketedGradReducer
    net = nn.Linear(4, 3)
    red = BucketedGradReducer(net.parameters())
    assert red.world == 1
    red.zero_grad()
    net(torch.ones(2, 4)).sum().backward()
    red.finish()
    assert torch.allclose(net.weight.grad, torch.full((3, 4), 2.0))
    assert red.grad_bytes() == (12 + 4) * 4          # every parameter starts on a 16-byte boundary: the 3-element bias takes 4


# ------------------------------------------------------------------------------------------------ 8 ranks (BASELINE configs[3]'s world size)
def _worker8(rank, world, port, ret):
    """The reducer at the world size of BASELINE configs[3] (8 ranks, gloo on CPU): buckets go out from the autograd hooks in
    the order backward completes them (overlap), finish() launches only what got no gradient, every rank ends with the
    global-batch gradient and -- after ten SGD steps on per-rank data -- bit-identical weights."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd.dp import BucketedGradReducer, broadcast_module_state
        torch.set_num_threads(1)
        g, d, unused = _toy(300 + rank)
        broadcast_module_state([g, d, unused])
        gparams = list(g.parameters()) + list(unused.parameters())
        red_g = BucketedGradReducer(gparams, bucket_bytes=512, gather=True)
        red_d = BucketedGradReducer(d.parameters(), bucket_bytes=64, gather=True)
        nb = len(red_g.buckets)
        assert nb >= 3 and len(red_d.buckets) >= 2
        # buckets holding a never-used parameter cannot complete from the hooks: finish() reduces them
        dead = [i for i, b in enumerate(red_g.buckets) if any(any(p is q for q in unused.parameters()) for p in b["params"])]
        live = [i for i in range(nb) if i not in dead]
        opt = torch.optim.SGD(list(g.parameters()) + list(d.parameters()), lr=0.05)
        gen = torch.Generator().manual_seed(1000 + rank)
        for step in range(10):
            x, y = torch.rand(2, 3, 8, 8, generator=gen), torch.rand(2, 3, 8, 8, generator=gen)
            fake = g(x)
            for p in d.parameters():
                p.requires_grad = True
            red_d.zero_grad()
            (0.5 * ((d(fake.detach()) ** 2).mean() + ((d(y) - 1) ** 2).mean())).backward()
            log_before_finish = list(red_d.launch_log)
            red_d.finish()
            assert [w for _, w in log_before_finish] == ["hook"] * len(red_d.buckets), log_before_finish      # all during backward
            for p in d.parameters():
                p.requires_grad = False
            red_g.zero_grad()
            ((fake - y).abs().mean() + 0.1 * ((d(fake) - 1) ** 2).mean()).backward()
            before = list(red_g.launch_log)
            red_g.finish()
            after = list(red_g.launch_log)
            # overlap: every bucket whose parameters all get gradients was reduced from a hook, i.e. DURING backward and before
            # finish(), in the order backward completed them; finish() launches exactly the rest
            assert sorted(i for i, _ in before) == live and all(w == "hook" for _, w in before), (before, live)
            assert sorted(i for i, w in after[len(before):]) == dead and all(w == "finish" for _, w in after[len(before):]), after
            assert len(live) >= 2
            for p in d.parameters():
                p.requires_grad = True
            opt.step()
        flat = torch.cat([p.detach().flatten() for p in list(g.parameters()) + list(d.parameters())])
        got = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(got, flat)
        assert all(torch.equal(t, got[0]) for t in got), "weights diverged across the 8 ranks"
        times = red_g.time_buckets(reps=1)
        assert len(times) == nb and all(t["ms"] > 0 for t in times)
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_bucketed_reducer_world8_gloo_launch_order_overlap_and_lockstep():
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


# ------------------------------------------------------------------------------------------------ segmented backward_G (capture mode "segments")
def _worker_segments(rank, world, port, ret):
    """FFWMTrainer(segmented_backward=True) on CPU, two gloo ranks: backward_G runs as three autograd segments cut at ALIASES of the
    generated images / flow fields (a coarse output feeds the next level, so the tensors themselves are no cut); every finished
    network's buckets go out before the next segment is issued.  The result must be the unsegmented data-parallel step's, bit for
    bit, and the ranks stay in lock-step."""
    import sys
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import torch_refs
        from ffwm_amd import trainer
        torch.set_num_threads(4)
        out = []
        for seg in (False, True):
            t = trainer.FFWMTrainer("cpu", world_size=world, seed=3, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat, ngf=8,
                                    bucket_bytes=1 << 20, segmented_backward=seg)
            groups = [b["group"] for b in t.red_G.buckets]
            assert set(groups) == {0, 1, 2} and groups == sorted(groups, reverse=True), groups      # a bucket never spans two networks
            batch = trainer.synthetic_batch(2, "cpu", seed=50 + rank)                                # per-rank data
            if seg:
                t.red_G.set_overlap(False)       # the regime of a replayed graph: no hook runs, the segments hand the buckets over
            t.step(batch)
            log = list(t.red_G.launch_log)
            assert sorted(i for i, _ in log) == list(range(len(groups))), log
            if seg:
                # launch order = completion order of the segments: flowNetB's buckets, netG's, flowNetF's; none left to finish()
                assert [groups[i] for i, _ in log] == sorted((groups[i] for i, _ in log), key=lambda g: {1: 0, 2: 1, 0: 2}[g]), log
                assert all(w == "segment" for _, w in log), log
                last_seg = [i for i, _ in log if groups[i] == 0]
                assert len(log) - len(last_seg) >= 2            # in flight before the last backward segment is issued
            t.step(batch)
            w = torch.cat([p.detach().flatten() for m in (t.flowNetF, t.flowNetB, t.netG, t.netD) for p in m.parameters()])
            out.append((w, t.red_G.flat.clone(), t.loss_values()))
            got = [torch.zeros_like(w) for _ in range(world)]
            dist.all_gather(got, w)
            assert torch.equal(got[0], got[1]), "weights differ across the ranks (segmented=%s)" % seg
        assert torch.equal(out[0][1], out[1][1]), "segmented backward changed the reduced gradients: %g" % (out[0][1] - out[1][1]).abs().max()
        assert torch.equal(out[0][0], out[1][0])
        assert out[0][2] == out[1][2]
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_segmented_backward_G_two_ranks_gloo_equals_the_unsegmented_step():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_segments, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
