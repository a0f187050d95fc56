"""The data-parallel train step end to end on the GPU box: two ranks share the single MI355X of the test box
(backend gloo -- RCCL refuses two ranks on one device; the collective itself is torch.distributed's business,
what is under test is everything around it): FFWMTrainer with the HIP kernels, fused spectral norm, flat
gradient buckets, hook-launched all-reduces overlapping backward, the identity pre-fit broadcast, and the
three-graph capture path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd import trainer
        torch.backends.cudnn.benchmark = False
        dev = torch.device("cuda", 0)
        t = trainer.FFWMTrainer(dev, world_size=world, seed=10 + rank, ngf=16, bucket_bytes=4 << 20)
        assert t.red_G.world == world and len(t.red_G.buckets) > 1

        def flat(mods):
            return torch.cat([p.detach().flatten() for m in mods for p in m.parameters()])

        def same_on_all_ranks(v, tol=0.0):
            got = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(got, v)
            return all((g - got[0]).abs().max().item() <= tol for g in got)

        nets_ = [t.flowNetF, t.flowNetB, t.netG, t.netD]
        assert same_on_all_ranks(flat(nets_)), "weights differ after the start-up broadcast"
        batch = trainer.synthetic_batch(2, dev, seed=100 + rank)          # different data per rank
        t.pretrain_flow_identity(batch, steps=3)
        assert same_on_all_ranks(flat([t.flowNetF, t.flowNetB])), "flow nets differ after the pre-fit broadcast"
        # ten steps on per-rank data: BatchNorm statistics and the spectral-norm power-iteration vectors evolve PER RANK
        # (dp.py: DDP's default behaviour), the weights must stay in lock-step through the averaged gradients
        for i in range(10):
            t.step(batch if i % 2 == 0 else trainer.synthetic_batch(2, dev, seed=200 + 10 * i + rank))
        torch.cuda.synchronize()
        vals = t.loss_values()
        assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), vals
        # identical averaged gradients -> identical Adam updates on every rank
        assert same_on_all_ranks(flat(nets_), tol=1e-6), "weights diverged across ranks after ten DP steps"
        assert same_on_all_ranks(t.red_G.buckets[0]["flat"], tol=0.0), "gradient buckets differ across ranks"
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def _worker8(rank, world, port, ret):
    """BASELINE configs[3]'s world size as a dry run: 8 gloo ranks sharing this box's GPU, narrow nets (ngf 16), batch 1 per
    rank.  Asserted per step: the G buckets completed by backward went out from the autograd hooks (overlap) and finish()
    launched the rest; after ten steps: weights in lock-step (<= 1e-6), and the spectral-norm power-iteration vectors -- which
    evolve per rank, dp.py -- within 1e-4 of rank 0's (they depend on the weights alone, so identical weights keep them together;
    BatchNorm running statistics are data-dependent and do differ)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd import trainer
        torch.backends.cudnn.benchmark = False
        dev = torch.device("cuda", 0)
        t = trainer.FFWMTrainer(dev, world_size=world, seed=20 + rank, ngf=16, bucket_bytes=2 << 20)
        nb = len(t.red_G.buckets)
        assert t.red_G.world == world and nb >= 4

        def gathered(v):
            got = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(got, v)
            return got

        def flat(ts):
            return torch.cat([p.detach().flatten().float() for p in ts])
        params = [p for m in (t.flowNetF, t.flowNetB, t.netG, t.netD) for p in m.parameters()]
        sn_u = [b for m in (t.netG, t.netD) for n, b in m.named_buffers() if n.endswith("weight_u")]
        assert len(sn_u) >= 50
        hooked = 0
        for i in range(10):
            t.step(trainer.synthetic_batch(1, dev, seed=500 + 10 * i + rank))
            log = t.red_G.launch_log
            assert sorted(b for b, _ in log) == list(range(nb)), log            # every bucket reduced exactly once
            hooked += sum(1 for _, w in log if w == "hook")
            # the hook launches come first (during backward), finish() only adds what backward could not complete
            first_finish = next((k for k, (_, w) in enumerate(log) if w == "finish"), len(log))
            assert all(w == "finish" for _, w in log[first_finish:]), log
        assert hooked >= 10 * (nb - 2), (hooked, nb)                             # nearly all buckets overlap with backward
        torch.cuda.synchronize()
        got = gathered(flat(params))
        assert all((g - got[0]).abs().max().item() <= 1e-6 for g in got), "weights diverged across the 8 ranks"
        gu = gathered(flat(sn_u))
        drift = max((g - gu[0]).abs().max().item() for g in gu)
        assert drift <= 1e-4, "spectral-norm u drifted by %.3e across ranks" % drift
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def _worker_graph(rank, world, port, ret):
    """What `bench.py --gpus N` runs for N > 1: the step replayed from THREE captured graphs with the two gradient all-reduces between
    them (gloo here: two ranks share the box's GPU).  Per-rank data; after the replays the weights must be in lock-step across the
    ranks, the losses finite, and equal (to fp32 reduction noise) to an eager data-parallel trainer fed the same batches."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd import trainer
        torch.backends.cudnn.benchmark = False
        dev = torch.device("cuda", 0)
        tg = trainer.FFWMTrainer(dev, world_size=world, seed=30 + rank, ngf=16, bucket_bytes=4 << 20, capturable=True)
        te = trainer.FFWMTrainer(dev, world_size=world, seed=30 + rank, ngf=16, bucket_bytes=4 << 20)
        batch = trainer.synthetic_batch(2, dev, seed=700 + rank)

        def flat(t):
            return torch.cat([p.detach().flatten().float() for m in (t.flowNetF, t.flowNetB, t.netG, t.netD) for p in m.parameters()])

        def spread(v):
            got = [torch.zeros_like(v) for _ in range(world)]
            dist.all_gather(got, v)
            return max((g - got[0]).abs().max().item() for g in got)
        for _ in range(2):                         # capture() runs 2 eager warm-up steps
            te.step(batch)
        tg.capture(batch, warmup=2, mode="serial")
        assert len(tg._graphs) == 3
        for _ in range(3):
            te.step(batch)
            tg.step(batch)
        torch.cuda.synchronize()
        vg, ve = tg.loss_values(), te.loss_values()
        assert all(torch.isfinite(torch.tensor(v)) for v in vg.values()), vg
        assert spread(flat(tg)) <= 1e-6, "captured data-parallel step: weights diverged across the ranks"
        # float atomics make two runs of the same GAN step differ in the last bits and five optimisation steps of these narrow nets
        # amplify that: two IDENTICAL eager data-parallel trainers end 0.3-2.8 % apart in their losses (measured); the sharp
        # criterion is the lock-step above
        for k in ("G", "D", "l1", "illu"):
            assert abs(vg[k] - ve[k]) <= 5e-2 * (1 + abs(ve[k])), (k, vg[k], ve[k])
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def _worker_ingraph(rank, world, port, ret):
    """Capture mode "ingraph" on the one-GPU box: a ONE-rank RCCL process group with force_collectives=True -- every bucket's
    all-reduce is really issued (torch.distributed's NCCL backend: its own stream, its events, its watchdog thread) and CAPTURED
    into the step's single graph from the reducers' autograd hooks.  Asserted: the probe graph replays correctly; ONE graph holds the
    step; at capture time >= n - 2 of the n G buckets were launched from a hook, i.e. while backward was still being issued (they
    are nodes of the graph ahead of the rest of backward); netD's forward / backward sit on the D side branch and its buckets are
    reduced at the join, on the step's stream; ten replays on changing
    batches leave finite losses and the weights of a plain one-GPU captured trainer fed the same batches (a one-rank sum is the
    identity) to fp32 noise."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from ffwm_amd import trainer
        torch.backends.cudnn.benchmark = False
        assert trainer.probe_collective_capture(dev), "RCCL all-reduce could not be captured into a hipGraph"
        tg = trainer.FFWMTrainer(dev, world_size=1, seed=50, ngf=16, bucket_bytes=4 << 20, capturable=True, force_collectives=True)
        tp = trainer.FFWMTrainer(dev, world_size=1, seed=50, ngf=16, bucket_bytes=4 << 20, capturable=True)
        assert tg.red_G.active and tg.dp_active and not tp.red_G.active
        batch = trainer.synthetic_batch(2, dev, seed=1000)
        tg.capture(batch, warmup=2, mode="ingraph")          # opt-in since round 5 (the default for several ranks is "serial")
        tp.capture(batch, warmup=2)
        assert tg.capture_mode == "ingraph" and len(tg._graphs) == 1
        log_D, log_G = tg._captured_launch_log
        n = len(tg.red_G.buckets)
        assert sorted(b for b, _ in log_G) == list(range(n)), log_G
        assert sum(1 for _, w in log_G if w == "hook") >= n - 2, log_G
        assert tg._d_side == "reduce_on_main" and log_D and all(w == "finish" for _, w in log_D), log_D

        # Two trainers of one seed, fed the same batches.  Float atomics make their gradients differ in the last bits and the GAN step of
        # these narrow nets (Adam normalises noise-level gradients to full-size updates) amplifies that from the first warm-up step on:
        # over 10 processes x 10 replays (profiles/r05_ingraph_repeat.txt) the relative loss difference between the two was 0.2-1.2 %
        # at the first replay and wandered between 0.3 % and 14 % afterwards, without trend.  Round 4 asserted 5 % on the values after
        # the tenth replay and failed 1 run in 15 -- that assertion, not the captured collectives, was the failure (the retry it was
        # wrapped in is gone).  Asserted now: finite losses and gradients at every replay, the first replay within 3 %, every replay
        # within 30 % (an all-reduce that was lost, doubled or applied to the wrong bucket moves the losses by far more: the D / G
        # gradients would be zero or twice their size).
        diag = []
        for i in range(10):
            bi = batch if i % 2 == 0 else trainer.synthetic_batch(2, dev, seed=1100 + i)
            tg.step(bi)
            tp.step(bi)
            torch.cuda.synchronize()
            vg, vp = tg.loss_values(), tp.loss_values()
            assert all(torch.isfinite(torch.tensor(v)) for v in vg.values()), (i, vg)
            gg, gp = tg.red_G.flat, tp.red_G.flat
            assert bool(torch.isfinite(gg).all()) and bool(torch.isfinite(tg.red_D.flat).all()), "non-finite gradients at replay %d" % i
            rel_g = float((gg - gp).norm() / (gp.norm() + 1e-30))
            rel_l = max(abs(vg[k] - vp[k]) / (1 + abs(vp[k])) for k in ("G", "D", "l1", "illu"))
            diag.append((round(rel_l, 5), round(rel_g, 5)))
            if i == 0:
                assert rel_l <= 3e-2, ("first replay", diag)
        ret["diag"] = diag
        assert max(d[0] for d in diag) <= 0.30, ("losses drifted apart", diag)
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def test_dp_collectives_captured_into_the_step_graph_one_rank_rccl():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_ingraph, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret.get(0) == "ok", dict(ret)


def test_dp_captured_three_graph_step_two_ranks_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_graph, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_dp_train_step_eight_ranks_dry_run_on_one_gpu():
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker8, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, dict(ret)


def test_dp_train_step_two_ranks_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_bench_launch_path_two_ranks_gloo():
    """`bench.py --gpus 2` exactly as the driver launches it (python -m torch.distributed.run, one rank per process,
    RANK / LOCAL_RANK / WORLD_SIZE from the environment), with FFWM_DIST_BACKEND=gloo so the two ranks may share
    this box's single GPU: init_dist, per-rank batches, the reducers, the barrier + max-over-ranks timing and the JSON
    lines from rank 0 (the last one is the line the driver parses).  (RCCL itself needs one device per rank: that run is the driver's.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FFWM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--flow-init", "random", "--no-kernels", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 3, r.stdout[-2000:]          # rank 0 alone prints: kernels, detail, and LAST the compact judged line
    assert len(lines[-1]) < 4096
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 16 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["grad_bytes_per_step"] > 400e6 and out["config"]["grad_buckets"] >= 2


def _worker_segments(rank, world, port, ret):
    """The five-graph ("segments") replay WITH its side streams -- the configuration in which round 4 saw non-finite weight gradients
    of netG's 256 / 384-channel 3x3 layers.  Cause (round 5, profiles/r05_wgrad_nan_root_cause.txt): hipMemsetAsync nodes whose fill
    pattern the runtime corrupted; the library's own zero-fill kernel replaced them.  Two gloo ranks share the GPU; asserted over 12
    replays: finite losses, weights and gradient buckets, and the ranks' weights in lock step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd import trainer
        torch.backends.cudnn.benchmark = False
        dev = torch.device("cuda", 0)
        t = trainer.FFWMTrainer(dev, world_size=world, seed=60 + rank, ngf=16, bucket_bytes=8 << 20, capturable=True)
        batch = trainer.synthetic_batch(2, dev, seed=1300 + rank)
        t.capture(batch, warmup=2, mode="segments")
        assert len(t._graphs) == 5 and t.flow_stream is not None and t.loss_streams is not None        # the side streams stayed
        for i in range(12):
            t.step(batch if i % 2 == 0 else trainer.synthetic_batch(2, dev, seed=1400 + 10 * i + rank))
            torch.cuda.synchronize()
            vals = t.loss_values()
            assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), (i, vals)
            assert bool(torch.isfinite(t.red_G.flat).all()) and bool(torch.isfinite(t.red_D.flat).all()), "non-finite gradients at replay %d" % i
        spread = t.rank_spread()
        assert all(v <= 1e-6 for v in spread.values()), spread
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def test_dp_five_graph_replay_with_side_streams_two_ranks_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_segments, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def _worker_full_size(rank, world, port, ret):
    """BASELINE configs[3]'s per-rank workload with world > 1 for the first time: ngf 64 flow nets, batch 8 per
    rank, 64 MiB buckets (428 MB of G gradients), the capture mode bench.py uses for several ranks ("serial": three graphs, the two
    all-reduces between them) -- two gloo ranks share the GPU.  Three replays on per-rank data; asserted: finite losses, every bucket
    reduced once per step, the ranks' weights in lock step (identical averaged gradients -> identical Adam updates)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ffwm_amd import miopen_tuning, trainer
        torch.backends.cudnn.benchmark = False
        miopen_tuning.install()
        dev = torch.device("cuda", 0)
        t = trainer.FFWMTrainer(dev, world_size=world, seed=0, bucket_bytes=64 << 20, capturable=True)
        assert t.red_G.grad_bytes() > 400e6 and len(t.red_G.buckets) >= 6
        batch = trainer.synthetic_batch(8, dev, seed=1 + rank)
        t.capture(batch, warmup=2)
        assert t.capture_mode == "serial" and len(t._graphs) == 3          # the default for several ranks
        nb = len(t.red_G.buckets)
        for i in range(3):
            t.step(batch if i != 1 else trainer.synthetic_batch(8, dev, seed=50 + rank))
            torch.cuda.synchronize()
            assert sorted(b for b, _ in t.red_G.launch_log) == list(range(nb)), t.red_G.launch_log
            vals = t.loss_values()
            assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), (i, vals)
        spread = t.rank_spread()
        assert all(v <= 1e-6 for v in spread.values()), spread
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc()[-1800:])
    finally:
        dist.destroy_process_group()


def test_dp_full_size_step_bs8_per_rank_ngf64_two_ranks_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_full_size, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
