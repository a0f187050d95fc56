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
        for _ in range(2):
            t.step(batch)
        torch.cuda.synchronize()
        vals = t.loss_values()
        assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), vals
        # identical averaged gradients -> identical Adam updates on every rank
        assert same_on_all_ranks(flat(nets_), tol=1e-6), "weights diverged across ranks after two DP steps"
        assert same_on_all_ranks(t.red_G.buckets[0]["flat"], tol=0.0), "gradient buckets differ across ranks"
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "FAIL: %s\n%s" % (e, traceback.format_exc())
    finally:
        dist.destroy_process_group()


def test_dp_train_step_two_ranks_on_one_gpu():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)
