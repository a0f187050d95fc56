"""Generates the evaluation / checkpoint-interchange fixtures by running the REFERENCE's own
Python modules (imported from /root/reference, CPU) -- build container only; what travels is data:

  tests/golden/ckpt/7_net_flowNetF.pth   a checkpoint file written FROM THE REFERENCE'S FlowNet(4) exactly as BaseModel.save_networks
                                          does (models/base_model.py:172-191: torch.save(net.cpu().state_dict(), '<epoch>_net_<name>.pth'))
  tests/golden/reference_eval.pt          * per network (netG = FFWM(sn=True), netD = MSDiscriminator(128), flowNetF = FlowNet(4)): the
                                            state dict's key list, shapes and float64 (sum, abs-sum) per tensor of the reference module
                                            filled by tests/golden/fill.py -- netG's 65 MB of weights do not travel, the test re-derives
                                            them with the same closed form and must meet these checksums key by key;
                                          * the composed FFWMModel.test_forward (models/ffwm_model.py:183-189: flowNetF -> WarpNet -> netG
                                            -> GuidedFilter(32)) on closed-form inputs: strided samples + float64 sums of fake_F128,
                                            img_GF128, img_S_warp, the attention map and the three flows.

    python tests/golden/make_eval_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path.insert(0, REF)
np.int = int                                   # base_networks.py:366 uses the removed alias
tv = types.ModuleType("torchvision")           # losses.py:3 imports torchvision.models (never used here)
tv.models = types.ModuleType("torchvision.models")
sys.modules["torchvision"] = tv
sys.modules["torchvision.models"] = tv.models

import fill  # noqa: E402
from models import base_networks, external_function  # noqa: E402

torch.set_num_threads(8)
EPOCH = 7


def sums(sd):
    return {k: (tuple(v.shape), float(v.double().sum()), float(v.double().abs().sum())) for k, v in sd.items()}


def packed(t, step):
    return {"shape": tuple(t.shape), "sample": t[..., ::step, ::step].contiguous().clone(), "step": step,
            "sum": float(t.double().sum()), "abs_sum": float(t.double().abs().sum())}


out = {}
with torch.no_grad():
    flowNetF = fill.fill_module(base_networks.FlowNet(4)).eval()
    netG = fill.boost_output_gain(fill.fill_module(base_networks.FFWM(sn=True))).eval()      # image heads x 150: an image of std 0.14 (fill.py)
    netD = fill.fill_module(base_networks.MSDiscriminator(128, sigmoid=False)).eval()
    os.makedirs(os.path.join(HERE, "ckpt"), exist_ok=True)
    # models/base_model.py:172-191 (the CPU branch): torch.save(net.cpu().state_dict(), save_path)
    torch.save(flowNetF.cpu().state_dict(), os.path.join(HERE, "ckpt", "%s_net_%s.pth" % (EPOCH, "flowNetF")))
    out["state"] = {"flowNetF": sums(flowNetF.state_dict()), "netG": sums(netG.state_dict()), "netD": sums(netD.state_dict())}
    out["epoch"] = EPOCH
    # models/ffwm_model.py:183-189
    img_S, img_F = fill.image(2, 3, 128, 128, "eval_img_S"), fill.image(2, 3, 128, 128, "eval_img_F")
    warpNet = base_networks.WarpNet()
    gf128 = external_function.GuidedFilter(32)
    flow_F128, flow_F64, flow_F32 = flowNetF(img_S)
    img_S_warp = warpNet(img_S, flow_F128)
    _, _, fake_F128, att = netG(img_S, flow=[flow_F32, flow_F64, flow_F128], return_att=True)
    att = torch.mean(att[:, :64, :, :], (1,), keepdim=True)
    img_GF128 = gf128(fake_F128, img_F)
    # Round 5's fixture had a nearly constant generated image (std 0.006): the guided filter's a = cov / (var + 1e-8) divided by ~3e-5 and
    # the reference's OWN fp32 result was 1.3e-2 from its float64 evaluation -- the composed img_GF128 could only be bounded to 0.13.
    # With the boosted image heads (fill.boost_output_gain) the generated image has a std of 0.14 and the filter is well conditioned.  What
    # remains is the filter's own fp32 arithmetic: GuidedFilter(32) is cumsum -> window difference over 128-pixel rows / columns, and the
    # reference's fp32 result is ~2e-4 from its float64 evaluation on ANY image (also on the input pair).  The fixture therefore holds the
    # float64 output as well: a test can ask for "as close to float64 as the reference's fp32 is" instead of bit-chasing its rounding.
    gf64 = external_function.GuidedFilter(32).double()(fake_F128.double(), img_F.double())
    out["img_GF128_ref_fp32_vs_fp64"] = float((img_GF128.double() - gf64).abs().max())
    out["img_GF128_fp64"] = packed(gf64, 2)
    out["fake_F128_std"] = float(fake_F128.std())
    assert out["fake_F128_std"] >= 0.1 and out["img_GF128_ref_fp32_vs_fp64"] <= 3e-4, (out["fake_F128_std"], out["img_GF128_ref_fp32_vs_fp64"])
    out["gf128_on_images"] = packed(gf128(img_S, img_F), 2)
    out["test_forward"] = {"fake_F128": packed(fake_F128, 2), "img_GF128": packed(img_GF128, 2), "img_S_warp": packed(img_S_warp, 2),
                           "att": packed(att, 2), "flow_F128": packed(flow_F128, 2), "flow_F64": packed(flow_F64, 1),
                           "flow_F32": packed(flow_F32, 1)}
    # the discriminator's score of the generated image (netD is a checkpointed network too: models/ffwm_model.py:20-24)
    out["netD_score_of_fake"] = netD(fake_F128)
out["_meta"] = {"what": "the reference's modules (models/base_networks.py, models/external_function.py) imported on CPU, state from tests/golden/fill.py",
                "torch": torch.__version__}
torch.save(out, os.path.join(HERE, "reference_eval.pt"))
print("wrote reference_eval.pt:", {k: (v["shape"], round(v["sum"], 4)) for k, v in out["test_forward"].items()})
print("ckpt:", os.listdir(os.path.join(HERE, "ckpt")), os.path.getsize(os.path.join(HERE, "ckpt", "%s_net_flowNetF.pth" % EPOCH)), "bytes")
