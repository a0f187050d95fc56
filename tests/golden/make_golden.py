"""Generates tests/golden/*.pt by running the REFERENCE's own Python modules (imported from
/root/reference, CPU) on closed-form inputs/state (tests/golden/fill.py).  Runs only in the build
container; the fixtures (data: inputs are re-derived, outputs are stored) are what travels.

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

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
from models import base_networks, external_function, losses  # noqa: E402
from lightcnn import light_cnn  # noqa: E402

torch.set_num_threads(8)
out = {}


def sub(t, step):
    return t[..., ::step, ::step].contiguous()


with torch.no_grad():
    # ---- FlowNet(ngf=4): eval and train-mode forward (base_networks.py:59-165)
    net = fill.fill_module(base_networks.FlowNet(4))
    x = fill.image(2, 3, 128, 128, "flownet_in")
    net.eval()
    f128, f64, f32 = net(x)
    out["flownet4_eval"] = {"flow128_s2": sub(f128, 2), "flow64": f64, "flow32": f32,
                            "sum128": f128.double().sum(), "abs128": f128.double().abs().sum()}
    net.train()
    f128, f64, f32 = net(x)
    out["flownet4_train"] = {"flow128_s2": sub(f128, 2), "flow64": f64, "flow32": f32,
                             "sum128": f128.double().sum(), "abs128": f128.double().abs().sum(),
                             "bn_mean_conv0": net.conv0[1].running_mean.clone()}
    out["flownet64_keys"] = sorted(base_networks.FlowNet(64).state_dict().keys())

    # ---- FFWM netG (sn=True), eval forward (base_networks.py:274-347)
    netG = fill.fill_module(base_networks.FFWM(sn=True)).eval()
    img = fill.image(1, 3, 128, 128, "netG_in")
    flows = [fill.flow_field(1, s, s, "netG_flow%d" % s) for s in (32, 64, 128)]
    r32, r64, r128, att = netG(img, flow=flows, return_att=True)
    out["ffwm_eval"] = {"rec32": r32, "rec64": r64, "rec128_s2": sub(r128, 2), "sum128": r128.double().sum(),
                        "att_s8": sub(att, 8), "att_sum": att.double().sum()}
    out["ffwm_keys"] = sorted(netG.state_dict().keys())

    # ---- MSDiscriminator(128, sigmoid=False) eval (base_networks.py:354-437)
    netD = fill.fill_module(base_networks.MSDiscriminator(128, sigmoid=False)).eval()
    out["netD_eval"] = {"score": netD(fill.image(2, 3, 128, 128, "netD_in"))}
    out["netD_keys"] = sorted(netD.state_dict().keys())

    # ---- LightCNN-29 eval (lightcnn/light_cnn.py:82-129)
    lc = fill.fill_module(light_cnn.LightCNN_29Layers()).eval()
    _, fc, pool = lc(fill.image(2, 1, 128, 128, "lightcnn_in"))
    out["lightcnn_eval"] = {"fc": fc, "pool_s2": sub(pool, 2), "pool_sum": pool.double().sum()}
    out["lightcnn_keys"] = sorted(lc.state_dict().keys())

    # ---- GuidedFilter (external_function.py:239-277)
    gx, gy = fill.image(2, 3, 64, 64, "gf_x"), fill.image(2, 3, 64, 64, "gf_y")
    out["guided_filter_r8_64"] = external_function.GuidedFilter(8)(gx, gy)
    out["guided_filter_r16_64"] = external_function.GuidedFilter(16)(gx, gy)

    # ---- WarpNet (base_networks.py:168-173) incl. the flip+cat of FFWM.forward (:326-329)
    feat = fill.image(2, 6, 24, 20, "warp_feat")
    fl = fill.flow_field(2, 24, 20, "warp_flow", amp=1.05)
    w = base_networks.WarpNet()(feat, fl)
    out["warpnet"] = {"w": w, "flipcat": torch.cat((w, torch.flip(w, (3,))), 1)}

    # ---- MSL1Loss (losses.py:130-157)
    crit = losses.MSL1Loss(torch.nn.L1Loss())
    fl3 = [fill.flow_field(2, s, s, "msl1_flow%d" % s) for s in (128, 64, 32)]
    im3 = [fill.image(2, 3, s, s, "msl1_img%d" % s) for s in (128, 64, 32)]
    img_F = fill.image(2, 3, 128, 128, "msl1_F")
    mask = (fill.image(2, 1, 128, 128, "msl1_mask") > 0.4).float()
    out["msl1"] = {"masked": crit(fl3, im3, img_F, mask), "plain": crit(fl3, im3, img_F)}

    # ---- IdentityLoss grid + FFWMModel part grids (losses.py:100-112, ffwm_model.py:217-246)
    out["identity_grid98"] = losses.IdentityLoss(lc).build_grid(2, 98)

# part grids: the methods only need `.lm_F` and `.device` on self
from models import ffwm_model  # noqa: E402

dummy = types.SimpleNamespace(device=torch.device("cpu"))
dummy.build_grid = lambda lm, d: ffwm_model.FFWMModel.build_grid(dummy, lm, d)
g = torch.Generator().manual_seed(5)
dummy.lm_F = torch.randint(16, 112, (2, 600, 2), generator=g)
grids = ffwm_model.FFWMModel.get_part_grid(dummy)
out["part_grids"] = {"lm_F": dummy.lm_F, "el": grids[0], "er": grids[1], "n": grids[2], "m": grids[3]}

# ---- AffineRegularizationLoss (losses.py:181-223) with the CUDA ops replaced by their proven
#      CPU identities: LocalAttnReshape == pixel_shuffle, BlockExtractor with the
#      constant flow kz//2 == unfold.
class _Reshape(torch.nn.Module):
    def forward(self, x, k):
        return F.pixel_shuffle(x, k)


class _Extract(torch.nn.Module):
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, grid, f):
        k = self.k
        assert float(f.min()) == float(f.max()) == float(k // 2)
        b, _, h, w = f.shape
        return F.unfold(grid, k).view(b, k, k, h, w).permute(0, 3, 1, 4, 2).reshape(b, 1, h * k, w * k)


reg = {}
with torch.no_grad():
    for kz, s in ((3, 32), (5, 64), (7, 128)):
        m = losses.AffineRegularizationLoss(kz)
        m.reshape, m.extractor = _Reshape(), _Extract(kz)
        flow = fill.flow_field(2, s, s, "reg_flow%d" % s)
        reg["kz%d" % kz] = {"loss": m(flow).double(), "kernel": m.kernel.clone()}
    multi = losses.MultiAffineRegularizationLoss(kz_dic={1: 7, 2: 5, 3: 3})
    for key, inst in multi.method_dic.items():
        inst.reshape, inst.extractor = _Reshape(), _Extract(inst.kz)
    flows = [fill.flow_field(2, s, s, "reg_flow%d" % s) for s in (128, 64, 32)]
    reg["multi"] = multi(flows[::-1]).double()    # flownet_model.py:68 passes flows[::-1]
    reg["multi_layers"] = torch.tensor(multi.layers)
out["affine_reg"] = reg

# ---- MultiScaleLDLoss (losses.py:61-74,114-126) and PerceptualCorrectness.calculate_loss (:342-371, bilinear
#      path).  The reference's VGG19 needs torchvision's pretrained weights: the loss is fed closed-form
#      "features" directly (the class is instantiated without __init__).
g = torch.Generator().manual_seed(1234)
lm_S = torch.randint(16, 112, (2, 40, 2), generator=g)
lm_F = torch.randint(16, 112, (2, 40, 2), generator=g)
gate = (torch.rand(2, 40, 1, generator=g) > 0.3).float()
gate2 = torch.cat((gate, gate), 2)
ld_flows = [fill.flow_field(2, s, s, "ld_flow%d" % s) for s in (128, 64, 32)]
with torch.no_grad():
    try:
        ld = losses.MultiScaleLDLoss()(ld_flows, lm_S, lm_F, gate2)
        out["ld_loss"] = {"lm_S": lm_S, "lm_F": lm_F, "gate": gate, "loss": ld.double(), "mode": "reference"}
    except Exception as e:      # torch >= 1.6: lm_F.div(scale) is float and torch.gather refuses it; the reference
        # (PyTorch 1.5) floor-divides integer tensors.  Pin the per-scale LandmarkLoss instead, with the division
        # done the 1.5 way, and the weighted sum the class would have formed.
        crit = losses.LandmarkLoss()
        tot = 0
        for i, fl in enumerate(ld_flows):
            sc = 128 // fl.size(3)
            tot = tot + [1000, 1000, 1500][i] * crit(fl, lm_S // sc if sc > 1 else lm_S, lm_F // sc if sc > 1 else lm_F, gate2)
        out["ld_loss"] = {"lm_S": lm_S, "lm_F": lm_F, "gate": gate, "loss": tot.double(), "mode": "per-scale (torch>=1.6: %s)" % type(e).__name__}

    pc = object.__new__(losses.PerceptualCorrectness)
    torch.nn.Module.__init__(pc)
    pc.eps = 1e-8
    pc.l1_loss = torch.nn.L1Loss()
    tv_, sv_ = fill.image(2, 8, 16, 16, "pc_target") + 0.1, fill.image(2, 8, 16, 16, "pc_source") + 0.1
    pc.target_vgg, pc.source_vgg = {"relu1_1": tv_}, {"relu1_1": sv_}
    pflow = fill.flow_field(2, 32, 32, "pc_flow")
    pmask = (fill.image(2, 1, 32, 32, "pc_mask") > 0.4).float()
    out["correctness"] = {"masked": pc.calculate_loss(pflow, "relu1_1", pmask, True).double(),
                          "unmasked": pc.calculate_loss(pflow, "relu1_1", None, True).double()}

torch.save(out, os.path.join(HERE, "reference_modules.pt"))
tot = os.path.getsize(os.path.join(HERE, "reference_modules.pt"))
print("wrote reference_modules.pt  %.1f KiB" % (tot / 1024.0))
