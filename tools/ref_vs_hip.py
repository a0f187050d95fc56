"""Live three-way check on the MI355X: the REFERENCE's kernels (oracle/_ref, hipcc build of cuda/*/*_kernel.cu),
this library's HIP kernels, and the CPU oracle -- on the seeded cases of tests/golden/ref_ops_cases.py; then the
reference's kernels timed next to ours at the BASELINE shapes (configs[0], configs[4] per GPU).

    gpurun -- 'python tools/ref_vs_hip.py'       -> gpurun_out/ref_vs_hip.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import ref_ops_cases as cases  # noqa: E402
import oracle  # noqa: E402
from oracle import build_ref  # noqa: E402
from ffwm_amd import ops  # noqa: E402

DEV = "cuda:0"


def md(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / (1.0 + float(b.abs().max()))


def time_ms(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    mods = build_ref.load()
    assert mods is not None
    rs, be, lar = mods["resample2d"], mods["block_extractor"], mods["local_attn_reshape"]
    oracle.build()
    rows = []
    for name in cases.RS_CASES:
        for dn, dt in cases.DTYPES.items():
            in1, in2, go, ks, dil = cases.rs_inputs(name, dt)
            a, b, g = in1.to(DEV), in2.to(DEV), go.to(DEV)
            o = a.new_zeros(b.shape[0], a.shape[1], b.shape[2], b.shape[3])
            rs.forward(a, b, o, ks, dil)
            g1, g2 = torch.zeros_like(a), torch.zeros_like(b)
            rs.backward(a, b, g, g1, g2, ks, dil)
            ho = ops.resample2d_forward(a, b, ks, dil)
            h1, h2 = torch.zeros_like(a), torch.zeros_like(b)
            ops.resample2d_backward(a, b, g, ks, dil, h1, h2)
            co = oracle.resample2d_forward(in1, in2, ks, dil)
            c1, c2 = oracle.resample2d_backward(in1, in2, go, ks, dil)
            rows.append({"op": "resample2d", "case": name, "dtype": dn,
                         "hip_vs_ref": [md(ho, o), md(h1, g1), md(h2, g2)],
                         "oracle_vs_ref": [md(co, o), md(c1, g1), md(c2, g2)]})
    for name in cases.BE_CASES:
        for dn, dt in cases.DTYPES.items():
            src, flow, go, k = cases.be_inputs(name, dt)
            s, f, g = src.to(DEV), flow.to(DEV), go.to(DEV)
            o = f.new_zeros(s.shape[0], s.shape[1], k * f.shape[2], k * f.shape[3])
            be.forward(s, f, o, k)
            gs, gf = torch.zeros_like(s), torch.zeros_like(f)
            be.backward(s, f, g, gs, gf, k)
            ho = ops.block_extractor_forward(s, f, k)
            hs, hf = torch.zeros_like(s), torch.zeros_like(f)
            ops.block_extractor_backward(s, f, g, k, hs, hf)
            co = oracle.block_extractor_forward(src, flow, k)
            cs, cf = oracle.block_extractor_backward(src, flow, go, k)
            rows.append({"op": "block_extractor", "case": name, "dtype": dn,
                         "hip_vs_ref": [md(ho, o), md(hs, gs), md(hf, gf)],
                         "oracle_vs_ref": [md(co, o), md(cs, gs), md(cf, gf)]})
    for name in cases.LAR_CASES:
        for dn, dt in cases.DTYPES.items():
            x, go, k = cases.lar_inputs(name, dt)
            a, g = x.to(DEV), go.to(DEV)
            o = a.new_zeros(a.shape[0], 1, k * a.shape[2], k * a.shape[3])
            lar.forward(a, o, k)
            gi = torch.zeros_like(a)
            lar.backward(a, g, gi, k)
            ho = ops.local_attn_reshape_forward(a, k)
            hi = ops.local_attn_reshape_backward(g, k)
            rows.append({"op": "local_attn_reshape", "case": name, "dtype": dn,
                         "hip_vs_ref": [md(ho, o), md(hi, gi)],
                         "oracle_vs_ref": [md(oracle.local_attn_reshape_forward(x, k), o),
                                           md(oracle.local_attn_reshape_backward(go, k), gi)]})
    worst = {}
    for r in rows:
        for key in ("hip_vs_ref", "oracle_vs_ref"):
            k2 = (r["op"], r["dtype"], key)
            worst[k2] = max(worst.get(k2, 0.0), max(r[key]))
    for k2 in sorted(worst):
        print("%-20s %s %-14s worst rel diff %.3e" % (k2[0], k2[1], k2[2], worst[k2]))

    # ---- timing: the reference's kernels next to ours, same GPU, same tensors
    timing = []
    g = torch.Generator().manual_seed(0)
    in1 = torch.rand(1, 64, 128, 128, generator=g).to(DEV)
    in2 = torch.cat((torch.rand(1, 2, 128, 128, generator=g) * 6 - 3, torch.full((1, 1, 128, 128), 2.0)), 1).to(DEV)
    go = torch.rand(1, 64, 128, 128, generator=g).to(DEV)
    o, g1, g2 = torch.zeros_like(in1), torch.zeros_like(in1), torch.zeros_like(in2)
    timing.append({"shape": "cfg1 resample2d ks=4 fwd", "ref_ms": time_ms(lambda: rs.forward(in1, in2, o, 4, 1)),
                   "hip_ms": time_ms(lambda: ops.resample2d_forward(in1, in2, 4, 1, out=o))})

    def ref_bwd():
        g1.zero_(); g2.zero_(); rs.backward(in1, in2, go, g1, g2, 4, 1)

    def hip_bwd():
        g1.zero_(); ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2)
    timing.append({"shape": "cfg1 resample2d ks=4 bwd (+memsets)", "ref_ms": time_ms(ref_bwd), "hip_ms": time_ms(hip_bwd)})
    big1 = torch.rand(8, 64, 512, 512, generator=g).to(DEV)
    big2 = torch.cat((torch.rand(8, 2, 512, 512, generator=g) * 6 - 3, torch.full((8, 1, 512, 512), 2.0)), 1).to(DEV)
    bo = torch.zeros_like(big1)
    timing.append({"shape": "resample2d ks=4 fwd [8,64,512,512]", "ref_ms": time_ms(lambda: rs.forward(big1, big2, bo, 4, 1), 5),
                   "hip_ms": time_ms(lambda: ops.resample2d_forward(big1, big2, 4, 1, out=bo), 5)})
    del big1, big2, bo
    src = torch.rand(4, 128, 256, 256, generator=g).to(DEV)
    flow = (torch.rand(4, 2, 256, 256, generator=g) * 4 - 2).to(DEV)
    out = torch.zeros(4, 128, 768, 768, device=DEV)
    timing.append({"shape": "cfg5/GPU block_extractor k=3 fwd", "ref_ms": time_ms(lambda: be.forward(src, flow, out, 3), 5),
                   "hip_ms": time_ms(lambda: ops.block_extractor_forward(src, flow, 3, out=out), 5)})
    gs, gf = torch.zeros_like(src), torch.zeros_like(flow)

    def ref_bb():
        gs.zero_(); gf.zero_(); be.backward(src, flow, out, gs, gf, 3)

    def hip_bb():
        gs.zero_(); gf.zero_(); ops.block_extractor_backward(src, flow, out, 3, gs, gf)
    timing.append({"shape": "cfg5/GPU block_extractor k=3 bwd (+memsets)", "ref_ms": time_ms(ref_bb, 3), "hip_ms": time_ms(hip_bb, 3)})
    for t in timing:
        t["speedup"] = t["ref_ms"] / t["hip_ms"]
        print("%-48s reference %.4f ms   ours %.4f ms   x%.1f" % (t["shape"], t["ref_ms"], t["hip_ms"], t["speedup"]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ref_vs_hip.json"), "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "cases": rows, "timing": timing,
                   "worst": {"/".join(k): v for k, v in worst.items()}}, f, indent=1)


if __name__ == "__main__":
    main()
