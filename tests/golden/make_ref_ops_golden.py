"""Golden vectors of the three native ops, produced BY THE REFERENCE'S OWN KERNELS on an MI355X.

    gpurun -- 'python tests/golden/make_ref_ops_golden.py'        (writes gpurun_out/reference_ops_gfx950.pt)
    cp gpurun_out/reference_ops_gfx950.pt tests/golden/

oracle/_ref/*.so are the reference's `resample2d_cuda`, `block_extractor_cuda`, `local_attn_reshape_cuda` extensions
(cuda/*/*_kernel.cu + *_cuda.cc) compiled for gfx950 by oracle/build_ref.py.  They are called exactly the way the
reference's autograd Functions call them (models/external_function.py:19-56, 69-101, 111-144): the caller allocates
zero-filled outputs / gradients, `forward(...)` / `backward(...)` fill them in place.  Inputs come from
tests/golden/ref_ops_cases.py (seeded CPU generators).  This script needs a GPU and oracle/_ref; it reads nothing
from /root/reference at run time.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_ops_cases as cases  # noqa: E402
from oracle import build_ref  # noqa: E402


def checksum(*ts):
    return [float(t.double().sum()) for t in ts]


def main():
    assert torch.cuda.is_available(), "needs the MI355X"
    mods = build_ref.load()
    assert mods is not None, "oracle/_ref is not built (python oracle/build_ref.py where /root/reference exists)"
    rs, be, lar = mods["resample2d"], mods["block_extractor"], mods["local_attn_reshape"]
    dev = "cuda:0"
    out = {"_meta": {"device": torch.cuda.get_device_name(0), "torch": torch.__version__,
                     "arch": torch.cuda.get_device_properties(0).gcnArchName,
                     "what": "outputs of the reference's CUDA kernels (hipcc, gfx950) on the seeded inputs of ref_ops_cases.py",
                     "sample_stride": cases.SAMPLE_STRIDE},
           "resample2d": {}, "block_extractor": {}, "local_attn_reshape": {}}

    for name in cases.RS_CASES:
        for dn, dt in cases.DTYPES.items():
            in1, in2, go, ks, dil = cases.rs_inputs(name, dt)
            a, b, g = in1.to(dev), in2.to(dev), go.to(dev)
            # Resample2dFunction.forward (external_function.py:111-128)
            o = a.new_zeros(b.shape[0], a.shape[1], b.shape[2], b.shape[3])
            rs.forward(a, b, o, ks, dil)
            # Resample2dFunction.backward (external_function.py:130-144)
            g1, g2 = torch.zeros_like(a), torch.zeros_like(b)
            rs.backward(a, b, g, g1, g2, ks, dil)
            torch.cuda.synchronize()
            out["resample2d"][name + "/" + dn] = {"inputs_sum": checksum(in1, in2, go), "out": cases.pack(o),
                                                  "g1": cases.pack(g1), "g2": cases.pack(g2)}

    for name in cases.BE_CASES:
        for dn, dt in cases.DTYPES.items():
            src, flow, go, k = cases.be_inputs(name, dt)
            s, f, g = src.to(dev), flow.to(dev), go.to(dev)
            # BlockExtractorFunction.forward / backward (external_function.py:19-56)
            o = f.new_zeros(s.shape[0], s.shape[1], k * f.shape[2], k * f.shape[3])
            be.forward(s, f, o, k)
            gs, gf = torch.zeros_like(s), torch.zeros_like(f)
            be.backward(s, f, g, gs, gf, k)
            torch.cuda.synchronize()
            out["block_extractor"][name + "/" + dn] = {"inputs_sum": checksum(src, flow, go), "out": cases.pack(o),
                                                       "g_src": cases.pack(gs), "g_flow": cases.pack(gf)}

    for name in cases.LAR_CASES:
        for dn, dt in cases.DTYPES.items():
            x, go, k = cases.lar_inputs(name, dt)
            a, g = x.to(dev), go.to(dev)
            # LocalAttnReshapeFunction.forward / backward (external_function.py:69-101)
            o = a.new_zeros(a.shape[0], 1, k * a.shape[2], k * a.shape[3])
            lar.forward(a, o, k)
            gi = torch.zeros_like(a)
            lar.backward(a, g, gi, k)
            torch.cuda.synchronize()
            out["local_attn_reshape"][name + "/" + dn] = {"inputs_sum": checksum(x, go), "out": cases.pack(o),
                                                          "g_in": cases.pack(gi)}

    # the reference's known answer (test_local_attn_reshape.py:29-43): range(9) -> [[0,1,2],[3,4,5],[6,7,8]]
    x = torch.arange(9, dtype=torch.float32).view(1, 9, 1, 1).to(dev)
    o = x.new_zeros(1, 1, 3, 3)
    lar.forward(x, o, 3)
    out["local_attn_reshape"]["range9"] = o.cpu()

    dst = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dst, exist_ok=True)
    path = os.path.join(dst, "reference_ops_gfx950.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path), "bytes;",
          {k: len(v) for k, v in out.items() if k != "_meta"})


if __name__ == "__main__":
    main()
