import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_parity as T
import oracle
oracle.build()
from ffwm_amd import ops, _lib
DEV = "cuda"
for case in T.BA_LIN_CASES:
    src, flow, _, k = T._be_inputs(case, torch.float32)
    B, C, Hf, Wf = src.shape[0], src.shape[1], flow.shape[2], flow.shape[3]
    g = T._gen(200 + case[8])
    w = torch.randn(B, k * k, Hf, Wf, generator=g)
    go = torch.randn(B, C, Hf, Wf, generator=g)
    _, gs_ref, gf_ref, gw_ref = T._attention_reference(oracle, src, flow, w, k, go)
    for fused in (0, 1):
        for zero in (True, False):
            base = [torch.randn(src.shape, generator=g), torch.randn(flow.shape, generator=g), torch.randn(w.shape, generator=g)]
            if zero:
                base = [torch.zeros_like(b) for b in base]
            gs, gf, gw = (t.to(DEV) for t in base)
            _lib.set_option("ba_bwd_fused", fused)
            ops.block_attention_backward(src.to(DEV), flow.to(DEV), w.to(DEV), go.to(DEV), k, gs, gf, gw)
            errs = [float(((got.cpu() - b0) - ref).abs().max() / ref.abs().max()) for got, ref, b0 in ((gs, gs_ref, base[0]), (gf, gf_ref, base[1]), (gw, gw_ref, base[2]))]
            print(case, "fused", fused, "zero base", zero, ["%.1e" % e for e in errs])
_lib.set_option("ba_bwd_fused", 1)
