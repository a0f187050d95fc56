#!/usr/bin/env python
"""bench.py -- FFWM flow-warp hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|flownet|flowtrain|warp|ops]

Default workload = BASELINE.json configs[2]: the full FFWM train step (netG + netD + flowNetF +
flowNetB, all losses, three Adam optimizers) on synthetic MultiPIE-shaped 128x128 tensors, batch 8
PER GPU (weak scaling; N > 1 is configs[3]: DP with RCCL all-reduce of gradient buckets overlapped
with backward).  One "step" = one optimisation step on one batch resident in HBM.  fp32 throughout
(the reference's dtype).  Rank 0 prints ONE JSON line:

  metric/value/unit : train img/s, whole job
  roofline          : the hand-written HIP kernel that moves the most algorithmic bytes inside the
                      timed region, timed live with HIP events on its launch stream (ffwm_prof_*),
                      algorithmic bytes per launch / average duration vs the 8 TB/s HBM peak;
                      traffic = PMC-measured HBM-side bytes per launch (profiles/r01_pmc_traffic.json)
  kernels           : the same figures for every hand-written kernel seen in the timed region, and
                      for the stand-alone operator shapes of configs[0]/[4] (cfg-1 resample2d, cfg-5
                      block_extractor / local_attn_reshape), measured right after the timed region
  cpu_baseline      : (N=1, rank 0) the same train step on the host CPU cores -- this repo's PyTorch
                      modules with the oracle's C/OpenMP warp -- on a bounded sample (batch 8, 2 steps after a warm-up step)

Launch for N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                   --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
FP32_PEAK = 157.3e12       # FLOP/s, fp32 vector/MFMA
TRAIN_FLOP_PER_IMG = 444e9  # ~222 GMAC per image for the full train step
FLOWNET_FLOP_PER_IMG = 4.35e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train", choices=["train", "flownet", "flowtrain", "warp", "ops"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 8 train, 6 flownet)")
    ap.add_argument("--titers", type=int, default=0, help="0 = warm-up branch (<20000), 20000 = guided-filter branch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernels", action="store_true")
    ap.add_argument("--bucket-mb", type=int, default=64)
    ap.add_argument("--flow-init", default="fit-identity", choices=["fit-identity", "random"],
                    help="train workload: stand-in for the reference's PRETRAINED flow nets -- fit flowNetF/B to the identity "
                         "sampling grid for 80 untimed Adam steps (default), or leave them randomly initialised")
    ap.add_argument("--mfma-wgrad", default="on", choices=["on", "off"],
                    help="weight gradients of netG's large 3x3 convs on the hand-written MFMA kernel (off: vendor library)")
    ap.add_argument("--graph", default="off", choices=["on", "off"],
                    help="train workload: replay the step from captured hipGraphs, or run it eagerly (default: measured faster on ROCm 7.2)")
    return ap.parse_args()


def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # FFWM_DIST_BACKEND=gloo lets the launch path be exercised on a box with fewer GPUs than ranks
        # (ranks then share devices; RCCL refuses that).  The driver's runs use the default: nccl = RCCL.
        backend = os.environ.get("FFWM_DIST_BACKEND", "nccl")
        local = local % torch.cuda.device_count() if backend != "nccl" else local
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    return world, rank, local


def timed(step_fn, steps, warmup, world):
    for _ in range(warmup):
        step_fn()
    from ffwm_amd import _lib
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    rows = _lib.prof_collect()
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, rows


def pmc_traffic():
    """HBM-side bytes per dispatch measured with rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    runs of this same command, FETCH_SIZE doubled per the gfx950 calibration) -- profiles/r01_pmc_traffic.json.
    PMC counters cannot be read from inside the process, so the committed measurement is attached."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    except (OSError, ValueError):
        return {}


def kernel_rows(rows, tag):
    out = []
    for name, r in sorted(rows.items(), key=lambda kv: -kv[1]["total_ms"]):
        if r["avg_ms"] <= 0:
            continue
        bps = r["bytes_per_launch"] / (r["avg_ms"] * 1e-3)
        row = {"kernel": name, "where": tag, "launches": r["launches"], "avg_us": round(r["avg_ms"] * 1e3, 2),
               "total_ms": round(r["total_ms"], 3), "alg_MB": round(r["bytes_per_launch"] / 1e6, 3),
               "GBps": round(bps / 1e9, 1), "frac_hbm_peak": round(bps / HBM_PEAK, 4)}
        if r.get("flops_per_launch", 0) > 0:      # the MFMA kernels (conv3x3_wgrad, correlation_colmax)
            fps = r["flops_per_launch"] / (r["avg_ms"] * 1e-3)
            row.update({"alg_GFLOP": round(r["flops_per_launch"] / 1e9, 3), "TFLOPs": round(fps / 1e12, 2),
                        "frac_mfma_fp32_peak": round(fps / FP32_PEAK, 4)})
        out.append(row)
    return out


def standalone_kernels(reps=10):
    """cfg-1 / cfg-5 operator shapes, HIP-event timed through the library profiler."""
    from ffwm_amd import _lib, ops
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    rows = []

    def run(tag, fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        rows.extend(kernel_rows(_lib.prof_collect(), tag))

    src = torch.rand(4, 128, 256, 256, generator=g).to(dev)
    flow = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(dev)
    out = torch.empty(4, 128, 768, 768, device=dev)
    run("cfg5/GPU block_extractor k=3 src[4,128,256,256] flow~U[-2,2)", lambda: ops.block_extractor_forward(src, flow, 3, out=out))
    gs, gf = torch.zeros_like(src), torch.zeros_like(flow)
    run("cfg5/GPU block_extractor k=3 backward", lambda: ops.block_extractor_backward(src, flow, out, 3, gs, gf))
    del src, out, gs, gf
    attn = torch.rand(4, 9, 256, 256, generator=g).to(dev)
    o = torch.empty(4, 1, 768, 768, device=dev)
    gi = torch.empty_like(attn)
    run("cfg5/GPU local_attn_reshape k=3 [4,9,256,256]", lambda: ops.local_attn_reshape_forward(attn, 3, out=o))
    run("cfg5/GPU local_attn_reshape k=3 backward", lambda: ops.local_attn_reshape_backward(o, 3, gi))
    in1 = torch.rand(1, 64, 128, 128, generator=g).to(dev)
    in2 = torch.cat((torch.rand(1, 2, 128, 128, generator=g) * 6 - 3, torch.full((1, 1, 128, 128), 2.0)), 1).to(dev)
    o = torch.empty_like(in1)
    go = torch.rand(1, 64, 128, 128, generator=g).to(dev)
    g1, g2 = torch.zeros_like(in1), torch.empty_like(in2)
    run("cfg1 resample2d ks=4 [1,64,128,128]", lambda: ops.resample2d_forward(in1, in2, 4, 1, out=o))
    run("cfg1 resample2d ks=4 backward", lambda: ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2))
    return rows


def host_threads():
    """Threads for the CPU leg: the cores this process may run on, capped at 16 -- PyTorch's CPU
    convolutions at batch 2 stop scaling (and then collapse) well before that on a many-core host."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_train_baseline(titers):
    """The same train step on the host cores: this repo's modules on CPU, warps through the oracle's
    C/OpenMP restatement (test infrastructure, used here only as the CPU comparison leg)."""
    import oracle
    from ffwm_amd import trainer
    oracle.build()
    cores = host_threads()
    torch.set_num_threads(cores)

    def warp(images, flow, mode="bilinear"):
        return oracle.WarpOracleFn.apply(images, flow, False)

    def warp_flipcat(feat, flow):
        return oracle.WarpOracleFn.apply(feat, flow, True)

    bs, steps = 8, 2
    t = trainer.FFWMTrainer("cpu", seed=0, titers=titers, warp=warp, warp_flipcat=warp_flipcat)
    batch = trainer.synthetic_batch(bs, "cpu", seed=1)
    t.step(batch)                                  # untimed: allocator / oneDNN primitive warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        t.step(batch)
    dt = time.perf_counter() - t0
    return {"value": round(bs * steps / dt, 4), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%d full FFWM train steps, batch %d, after 1 warm-up step, %d torch/OpenMP threads, %.1f s"
                      % (steps, bs, cores, dt)}


def cpu_ops_baseline():
    import oracle
    oracle.build()
    cores = host_threads()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    g = torch.Generator().manual_seed(0)
    src = torch.rand(1, 32, 256, 256, generator=g)
    flow = torch.rand(1, 2, 256, 256, generator=g) * 4 - 2
    t0 = time.perf_counter()
    oracle.block_extractor_forward(src, flow, 3)
    dt = time.perf_counter() - t0
    nbytes = 4.0 * (32 * 65536 + 2 * 65536 + 32 * 9 * 65536)
    return {"value": round(nbytes / dt / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "oracle block_extractor forward, src [1,32,256,256] (1/16 of cfg-5 per GPU), %.2f s" % dt}


def main():
    args = parse()
    world, rank, local = init_dist(args)
    dev = torch.device("cuda", local)
    # MIOpen ships no gfx950 kernel database in this image: every conv kernel is JIT-compiled on a
    # fresh box.  Exhaustive find mode multiplies that start-up cost by the number of candidate
    # solvers (~10 min), so it is opt-in; the default is MIOpen's heuristic ("immediate") choice.
    torch.backends.cudnn.benchmark = os.environ.get("FFWM_MIOPEN_FIND", "0") == "1"
    from ffwm_amd import _lib, miopen_tuning
    # solver selection from the in-tree find-db (ffwm_amd/miopen_db, 245 KB of MIOpen's own text records for the
    # convolutions of this workload on gfx950 / 256 CUs): immediate mode, no find pass at start-up
    miopen_db = miopen_tuning.install()
    _lib.load()                                # fail loudly if the HIP library is missing

    result = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}

    if args.workload == "train":
        from ffwm_amd import trainer
        bs = args.batch or 8
        t = trainer.FFWMTrainer(dev, world_size=world, seed=0, titers=args.titers,
                                bucket_bytes=args.bucket_mb << 20, capturable=args.graph == "on",
                                mfma_wgrad=args.mfma_wgrad == "on")
        batch = trainer.synthetic_batch(bs, dev, seed=1 + rank)
        flow_fit = t.pretrain_flow_identity(batch) if args.flow_init == "fit-identity" else None
        graphed = args.graph == "on"
        if graphed:
            t.capture(batch, warmup=max(2, args.warmup))
        dt, rows = timed(lambda: t.step(batch, batch_increment=0), args.steps, args.warmup, world)
        if graphed:
            # HIP events cannot bracket kernels inside a replayed graph: time the hand-written kernels
            # in two EAGER steps of the same trainer right after the timed region
            t.release_graphs()
            _, rows = timed(lambda: t.step(batch, batch_increment=0), 2, 1, world)
        imgs = bs * world * args.steps
        result.update({"metric": "train img/s (128x128, full FFWM GAN step)", "value": round(imgs / dt, 2),
                       "unit": "img/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
                       "config": {"workload": "BASELINE configs[2]: full FFWM train step (netG+netD+flowNetF+flowNetB, "
                                              "all losses, 3x Adam), synthetic MultiPIE-shaped 128x128",
                                  "batch_per_gpu": bs, "global_batch": bs * world,
                                  "parallelism": "dp%d" % world, "launch": "hipGraph replay" if graphed else "eager",
                                  "miopen": "immediate mode%s" % (" + in-tree find-db (ffwm_amd/miopen_db)" if miopen_db else ", heuristic solver choice"),
                                  "conv_wgrad": ("MFMA kernel for %d netG layers" % getattr(t, "mfma_wgrad_layers", 0))
                                  if args.mfma_wgrad == "on" else "vendor library",
                                  "titers_branch": "<20000" if args.titers < 20000 else ">=20000",
                                  "weights": "seeded random init (no pretrained VGG19/LightCNN/FlowNet offline)",
                                  "flow_nets": ("fitted to the identity grid for 80 untimed steps (stand-in for the reference's "
                                                "pretrained flowNetF/B checkpoints), final L1 %s" % [round(v, 3) for v in flow_fit])
                                  if flow_fit else "randomly initialised"},
                       "img_per_s_per_gpu": round(imgs / dt / world, 2),
                       "fp32_flop_frac": round(imgs / dt / world * TRAIN_FLOP_PER_IMG / FP32_PEAK, 4),
                       "losses": {k: round(v, 5) for k, v in t.loss_values().items()}})
        if world > 1:
            result["config"]["grad_bytes_per_step"] = t.red_G.grad_bytes() + t.red_D.grad_bytes()
    elif args.workload == "flownet":
        from ffwm_amd import nets
        bs = args.batch or 6
        torch.manual_seed(0)
        net = nets.FlowNet(64).to(dev).eval()
        x = torch.rand(bs, 3, 128, 128, generator=torch.Generator().manual_seed(1 + rank)).to(dev)

        def step():
            with torch.no_grad():
                net(x)
        dt, rows = timed(step, args.steps, args.warmup, world)
        imgs = bs * world * args.steps
        result.update({"metric": "FlowNetF forward img/s (128x128)", "value": round(imgs / dt, 2), "unit": "img/s",
                       "ms_per_step": round(dt / args.steps * 1e3, 3),
                       "config": {"workload": "BASELINE configs[1]: FlowNetF forward-only, bs=%d" % bs,
                                  "batch_per_gpu": bs, "parallelism": "dp%d" % world},
                       "fp32_flop_frac": round(imgs / dt / world * FLOWNET_FLOP_PER_IMG / FP32_PEAK, 5)})
    elif args.workload == "flowtrain":
        # FlowNet pre-training step (train_flow.py / models/flownet_model.py:57-78): the only trainer of the
        # reference that runs the custom ops; README.md:105,116 trains it with batch 6
        from ffwm_amd import trainer
        bs = args.batch or 6
        t = trainer.FlowNetTrainer(dev, world_size=world, seed=0, bucket_bytes=args.bucket_mb << 20)
        batch = trainer.synthetic_batch(bs, dev, seed=1 + rank)
        dt, rows = timed(lambda: t.step(batch), args.steps, args.warmup, world)
        imgs = bs * world * args.steps
        result.update({"metric": "FlowNet pre-training img/s (128x128)", "value": round(imgs / dt, 2), "unit": "img/s",
                       "ms_per_step": round(dt / args.steps * 1e3, 3),
                       "config": {"workload": "FlowNetModel train step (correctness + affine regularisation + landmark "
                                              "losses, Adam), synthetic 128x128", "batch_per_gpu": bs,
                                  "parallelism": "dp%d" % world, "weights": "seeded random init"},
                       "losses": {k: round(v, 5) for k, v in t.loss_values().items()}})
    elif args.workload == "warp":
        # the warp + flip + cat sub-path of netG's warp-attention, forward + backward, bs images
        from ffwm_amd.external_function import WarpFlipCat
        bs = args.batch or 8
        g = torch.Generator().manual_seed(1 + rank)
        feats = [torch.rand(bs, c, s, s, generator=g).to(dev).requires_grad_(True) for c, s in ((128, 32), (64, 64), (64, 128))]
        def smooth(s):      # identity grid + ~3 px of low-frequency displacement: what a trained FlowNet produces
            lin = (torch.arange(s, dtype=torch.float32) + 0.5) / s * 2 - 1
            yy, xx = torch.meshgrid(lin, lin, indexing="ij")
            amp = 6.0 / s
            fx = xx + amp * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx)
            fy = yy + amp * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)
            return torch.stack((fx, fy), 0).unsqueeze(0).repeat(bs, 1, 1, 1).contiguous()
        flows = [smooth(s).to(dev).requires_grad_(True) for s in (32, 64, 128)]
        gos = [torch.rand(bs, 2 * c, s, s, generator=g).to(dev) for c, s in ((128, 32), (64, 64), (64, 128))]
        mod = WarpFlipCat()

        def step():
            for f, fl, go in zip(feats, flows, gos):
                f.grad = fl.grad = None
                mod(f, fl).backward(go)
        dt, rows = timed(step, args.steps, args.warmup, world)
        imgs = bs * world * args.steps
        result.update({"metric": "warp+flip+cat path img/s (3 netG levels, fwd+bwd)", "value": round(imgs / dt, 2),
                       "unit": "img/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
                       "config": {"workload": "netG warp-attention warp sub-path (warp + flip + cat), 3 levels, fwd+bwd, smooth flows", "batch_per_gpu": bs,
                                  "parallelism": "dp%d" % world}})
    else:   # ops: cfg-5 per GPU block_extractor forward + backward
        from ffwm_amd import ops
        g = torch.Generator().manual_seed(1 + rank)
        bs = args.batch or 4
        src = torch.rand(bs, 128, 256, 256, generator=g).to(dev)
        flow = (torch.rand(bs, 2, 256, 256, generator=g) * 4 - 2).to(dev)
        out = torch.empty(bs, 128, 768, 768, device=dev)
        gs, gf = torch.zeros_like(src), torch.zeros_like(flow)

        def step():
            ops.block_extractor_forward(src, flow, 3, out=out)
            gs.zero_()
            gf.zero_()
            ops.block_extractor_backward(src, flow, out, 3, gs, gf)
        dt, rows = timed(step, args.steps, args.warmup, world)
        imgs = bs * world * args.steps
        result.update({"metric": "block_extractor fwd+bwd img/s (256x256, C=128, k=3)", "value": round(imgs / dt, 2),
                       "unit": "img/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
                       "config": {"workload": "BASELINE configs[4] per GPU: block_extractor fwd+bwd", "batch_per_gpu": bs,
                                  "parallelism": "dp%d" % world}})

    if rank == 0:
        inrun = kernel_rows(rows, "timed region")
        if inrun:
            # the roofline kernel of an HBM-bound path = the hand-written kernel that moves the most
            # algorithmic bytes in a step (launch-latency-bound helpers on a few hundred KB -- guided filter
            # on 24 planes, spectral norm on 61 small matrices -- are listed in `kernels` with the rest)
            def hbm_roofline(top):
                pmc = pmc_traffic().get(top["kernel"]) if args.workload == "train" else None     # measured on the train workload
                return {"bound": "hbm", "kernel": top["kernel"], "achieved": top["GBps"], "peak": HBM_PEAK / 1e9,
                        "unit": "GB/s", "frac": top["frac_hbm_peak"],
                        "traffic": pmc["traffic_bytes"] if pmc else None,
                        "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                          "of this command; bytes per launch, FETCH_SIZE x2 per the gfx950 calibration)"
                        if pmc else None,
                        "avg_us": top["avg_us"], "alg_MB_per_launch": top["alg_MB"], "launches": top["launches"],
                        "note": "hand-written HIP kernel moving the most algorithmic bytes per step; HIP events "
                                "on its launch stream" + (" (eager steps run right after the graph-replayed "
                                "timed region)" if args.workload == "train" and args.graph == "on" else "")}
            hbm_rows = [r for r in inrun if "TFLOPs" not in r]
            mfma_rows = [r for r in inrun if "TFLOPs" in r]
            dominant = max(inrun, key=lambda r: r["total_ms"])
            if "TFLOPs" in dominant:
                # the dominant hand-written kernel of the step is a contraction on the matrix cores: its roofline
                # is the fp32 MFMA peak (157.3 TFLOP/s dense); algorithmic flops = 2 * 9 * B H W C K per launch
                pmc = pmc_traffic().get(dominant["kernel"]) if args.workload == "train" else None
                result["roofline"] = {"bound": "mfma", "kernel": dominant["kernel"], "achieved": dominant["TFLOPs"],
                                      "peak": FP32_PEAK / 1e12, "unit": "TFLOP/s", "frac": dominant["frac_mfma_fp32_peak"],
                                      "traffic": pmc["traffic_bytes"] if pmc else None,
                                      "avg_us": dominant["avg_us"], "alg_GFLOP_per_launch": dominant["alg_GFLOP"],
                                      "launches": dominant["launches"],
                                      "note": "hand-written kernel with the largest share of the step (fp32-in / fp32-accumulate "
                                              "v_mfma_f32_32x32x2_f32); flops and duration averaged over the layer shapes of "
                                              "the step; HIP events on its launch stream"}
                if hbm_rows:
                    result["roofline_hbm"] = hbm_roofline(max(hbm_rows, key=lambda r: r["alg_MB"] * r["launches"]))
            else:
                result["roofline"] = hbm_roofline(max(hbm_rows or mfma_rows, key=lambda r: r["alg_MB"] * r["launches"]))
        else:
            result["roofline"] = None
        result["kernels"] = inrun
        if not args.no_kernels:
            result["kernels"] = inrun + standalone_kernels()
        traffic = pmc_traffic() if args.workload == "train" else {}
        for row in result["kernels"]:
            if row["kernel"] in traffic and (row["where"] == "timed region" or not row["kernel"].startswith("warp")):
                row["pmc_traffic_MB"] = round(traffic[row["kernel"]]["traffic_bytes"] / 1e6, 3)
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_train_baseline(args.titers) if args.workload in ("train",) \
                    else cpu_ops_baseline()
            except Exception as e:      # the baseline leg must never take the measurement down
                result["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
