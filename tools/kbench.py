"""Per-kernel microbenchmark of the HIP ops (HIP-event timing through the library's own profiler).

    python tools/kbench.py [--reps 20] [--only be_fwd,warp_fwd,...] [--opt key=value ...]

Prints one line per kernel: avg ms, algorithmic GB/s, fraction of 8 TB/s.  Shapes are the
BASELINE.json configs (cfg-5 per-GPU for block_extractor / local_attn_reshape, cfg-1 for resample2d,
the three netG levels at bs=8 for the fused warp).
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ffwm_amd import _lib, ops  # noqa: E402

PEAK = 8.0e12


def run(name, fn, reps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    rows = _lib.prof_collect()
    _lib.prof_enable(False)
    out = [{"case": name, "kernel": "(whole call, stream time incl. torch kernels)", "avg_ms": round(e0.elapsed_time(e1) / reps, 5)}]
    for k, r in rows.items():
        gbs = r["bytes_per_launch"] / (r["avg_ms"] * 1e-3) / 1e9 if r["avg_ms"] > 0 else 0.0
        out.append({"case": name, "kernel": k, "avg_ms": round(r["avg_ms"], 5), "GBps": round(gbs, 1),
                    "frac_hbm_peak": round(gbs * 1e9 / PEAK, 4), "launches": r["launches"],
                    "MB": round(r["bytes_per_launch"] / 1e6, 2)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--flow", type=float, default=2.0, help="block_extractor flow ~ U[-f, f) px")
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--warp-flow", default="random", choices=["random", "smooth", "const"],
                    help="warp sampling grid: random = U[-1.1,1.1) per pixel (worst case: every lane its own cache line), "
                         "smooth = identity grid + a few pixels of low-frequency displacement (a trained FlowNet), "
                         "const = every pixel samples the image centre (an untrained FlowNet, tanh(~0))")
    args = ap.parse_args()
    for kv in args.opt:
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    only = set(filter(None, args.only.split(",")))
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    results = []

    def want(n):
        return not only or n in only

    if want("calib"):
        # achievable HBM rates on this box for plain streaming (torch kernels), for calibration
        a = torch.empty(4, 128, 768, 768, device=dev)
        b = torch.empty_like(a)
        for name, fn, nbytes in (("fill 1.2GB", lambda: a.fill_(1.0), a.numel() * 4),
                                 ("copy 1.2GB->1.2GB", lambda: b.copy_(a), 2 * a.numel() * 4)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            results.append({"case": "calibration " + name, "kernel": "torch", "avg_ms": round(ms, 4),
                            "GBps": round(nbytes / ms / 1e6, 1), "frac_hbm_peak": round(nbytes / ms / 1e6 / 8000, 4)})
        del a, b
    # cfg-5 per GPU: src [4,128,256,256], flow [4,2,256,256], k=3
    B = args.B
    if want("be_fwd") or want("be_bwd"):
        src = torch.rand(B, 128, 256, 256, generator=g).to(dev)
        flow = ((torch.rand(B, 2, 256, 256, generator=g) * 2 - 1) * args.flow).to(dev)
        out = torch.empty(B, 128, 768, 768, device=dev)
        if want("be_fwd"):
            results += run("cfg5 block_extractor fwd flow=U[-%g,%g)" % (args.flow, args.flow),
                           lambda: ops.block_extractor_forward(src, flow, 3, out=out), args.reps)
        if want("be_bwd"):
            go = torch.rand(B, 128, 768, 768, device=dev)
            gs = torch.zeros_like(src)
            gf = torch.zeros_like(flow)
            results += run("cfg5 block_extractor bwd", lambda: ops.block_extractor_backward(src, flow, go, 3, gs, gf), args.reps)
            del go, gs, gf
        del src, flow, out
    if want("attn"):
        # the fused extractor + attention consumer on the cfg-5 shape, next to the composition it replaces
        import torch.nn.functional as F
        src = torch.rand(B, 128, 256, 256, generator=g).to(dev)
        flow = ((torch.rand(B, 2, 256, 256, generator=g) * 2 - 1) * args.flow).to(dev)
        w = torch.softmax(torch.randn(B, 9, 256, 256, generator=g), 1).to(dev)
        out = torch.empty(B, 128, 256, 256, device=dev)
        go = torch.rand(B, 128, 256, 256, device=dev)
        gs, gf, gw = torch.zeros_like(src), torch.zeros_like(flow), torch.zeros_like(w)
        results += run("cfg5 block_attention fwd", lambda: ops.block_attention_forward(src, flow, w, 3, out=out), args.reps)
        results += run("cfg5 block_attention bwd",
                       lambda: ops.block_attention_backward(src, flow, w, go, 3, gs, gf, gw), args.reps)

        def composition():
            ext = ops.block_extractor_forward(src, flow, 3)
            return F.avg_pool2d(ext * ops.local_attn_reshape_forward(w, 3), 3, 3)
        results += run("cfg5 composition fwd (extract, reshape, mul, avg_pool)", composition, max(3, args.reps // 4))
        del src, flow, w, out, go, gs, gf, gw
    if want("lar"):
        attn = torch.rand(B, 9, 256, 256, generator=g).to(dev)
        o = torch.empty(B, 1, 768, 768, device=dev)
        results += run("cfg5 local_attn_reshape fwd", lambda: ops.local_attn_reshape_forward(attn, 3, out=o), args.reps)
        gi = torch.empty_like(attn)
        results += run("cfg5 local_attn_reshape bwd", lambda: ops.local_attn_reshape_backward(o, 3, gi), args.reps)
    if want("rs"):
        in1 = torch.rand(1, 64, 128, 128, generator=g).to(dev)
        in2 = torch.cat((torch.rand(1, 2, 128, 128, generator=g) * 6 - 3, torch.full((1, 1, 128, 128), 2.0)), 1).to(dev)
        o = torch.empty(1, 64, 128, 128, device=dev)
        go = torch.rand(1, 64, 128, 128, device=dev)
        g1, g2 = torch.zeros_like(in1), torch.empty_like(in2)
        for ks in (2, 4):
            results += run("cfg1 resample2d fwd ks=%d" % ks, lambda: ops.resample2d_forward(in1, in2, ks, 1, out=o), args.reps)
            results += run("cfg1 resample2d bwd ks=%d" % ks, lambda: ops.resample2d_backward(in1, in2, go, ks, 1, g1, g2), args.reps)
        # same op at an HBM-sized shape
        in1 = torch.rand(8, 64, 512, 512, generator=g).to(dev)
        in2 = torch.cat((torch.rand(8, 2, 512, 512, generator=g) * 6 - 3, torch.full((8, 1, 512, 512), 2.0)), 1).to(dev)
        o = torch.empty_like(in1)
        results += run("big resample2d fwd ks=4 [8,64,512,512]", lambda: ops.resample2d_forward(in1, in2, 4, 1, out=o), args.reps)
        del in1, in2, o
    def make_flow(b, size):
        if args.warp_flow == "random":
            return (torch.rand(b, 2, size, size, generator=g) * 2.2 - 1.1).to(dev)
        if args.warp_flow == "const":
            return torch.zeros(b, 2, size, size, device=dev) + 1e-3
        lin = (torch.arange(size, dtype=torch.float32) + 0.5) / size * 2 - 1
        yy, xx = torch.meshgrid(lin, lin, indexing="ij")
        amp = 6.0 / size                                          # ~3 pixels of displacement
        fx = xx + amp * torch.sin(3.1 * yy + 0.3) * torch.cos(2.3 * xx)
        fy = yy + amp * torch.cos(2.7 * xx - 0.2) * torch.sin(1.9 * yy)
        return torch.stack((fx, fy), 0).unsqueeze(0).repeat(b, 1, 1, 1).contiguous().to(dev)

    if want("warp"):
        for (C, S) in ((128, 32), (64, 64), (64, 128)):
            feat = torch.rand(8, C, S, S, generator=g).to(dev)
            fl = make_flow(8, S)
            o = torch.empty(8, 2 * C, S, S, device=dev)
            go = torch.rand(8, 2 * C, S, S, device=dev)
            gfe, gfl = torch.zeros_like(feat), torch.zeros_like(fl)
            results += run("netG warp+flip+cat fwd [8,%d,%d,%d] flow=%s" % (C, S, S, args.warp_flow), lambda: ops.warp_forward(feat, fl, True, out=o), args.reps)
            results += run("netG warp+flip+cat bwd [8,%d,%d,%d]" % (C, S, S), lambda: ops.warp_backward(feat, fl, go, True, gfe, gfl), args.reps)
        feat = torch.rand(32, 64, 256, 256, generator=g).to(dev)
        fl = make_flow(32, 256)
        o = torch.empty(32, 128, 256, 256, device=dev)
        results += run("big warp+flip+cat fwd [32,64,256,256] flow=%s" % args.warp_flow, lambda: ops.warp_forward(feat, fl, True, out=o), args.reps)
        del feat, fl, o
    if want("corr"):
        # correlation column maximum of the correctness loss at relu1_1 of 128 x 128 images, batch 6 (flownet_model.py:67):
        # fp32 MFMA kernel vs torch.bmm + max (which materialises 6 x 1 GiB)
        import time
        for (N, C) in ((16384, 64), (4096, 128), (1024, 256)):
            s_ = torch.randn(6, N, C, generator=g).to(dev)
            t_ = torch.randn(6, C, N, generator=g).to(dev)
            s_ = s_ / (s_.norm(dim=2, keepdim=True) + 1e-8)
            t_ = t_ / (t_.norm(dim=1, keepdim=True) + 1e-8)
            rows = run("correlation colmax B=6 N=%d C=%d" % (N, C), lambda: ops.correlation_colmax(s_, t_), args.reps)
            for r in rows:
                r["TFLOPs"] = round(2.0 * 6 * N * N * C / (r["avg_ms"] * 1e-3) / 1e12, 1)
                r["frac_fp32_mfma_peak"] = round(r["TFLOPs"] / 157.3, 3)
            results += rows
            for _ in range(3):
                torch.bmm(s_, t_).max(dim=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                torch.bmm(s_, t_).max(dim=1)
            torch.cuda.synchronize()
            results.append({"case": "torch.bmm + max B=6 N=%d C=%d" % (N, C), "kernel": "wall",
                            "avg_ms": round((time.perf_counter() - t0) / args.reps * 1e3, 4)})
            del s_, t_
    if want("affine"):
        # MultiAffineRegularizationLoss at the reference's FlowNet pre-training sizes (bs 6, kz 7/5/3 on 128/64/32 px):
        # the reference's op composition on the HIP kernels vs the fused kernel, forward + backward, wall time
        import time
        from ffwm_amd.losses import MultiAffineRegularizationLoss
        flows = [(torch.rand(6, 2, sz, sz, generator=g) * 2 - 1).to(dev).requires_grad_(True) for sz in (32, 64, 128)]
        for fused in (False, True):
            m = MultiAffineRegularizationLoss({1: 7, 2: 5, 3: 3}, fused=fused)

            def step():
                for f in flows:
                    f.grad = None
                m(flows).backward()
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.reps * 1e3
            results.append({"case": "MultiAffineRegularizationLoss fwd+bwd bs=6 (3 scales), %s" % ("fused kernel" if fused else "op composition"),
                            "kernel": "wall", "avg_ms": round(ms, 4)})
    for r in results:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
