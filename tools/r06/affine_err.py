import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import fill
from ffwm_amd.losses import AffineRegularizationLoss, MultiAffineRegularizationLoss
gold = torch.load("tests/golden/reference_modules.pt")["affine_reg"]
for kz, s in ((3, 32), (5, 64), (7, 128)):
    f = fill.flow_field(2, s, s, "reg_flow%d" % s)
    l64 = float(AffineRegularizationLoss(kz, fused=False)(f.double().cuda()))
    l64f = float(AffineRegularizationLoss(kz, fused=True)(f.double().cuda()))
    lf = float(AffineRegularizationLoss(kz, fused=True)(f.cuda()))
    lc = float(AffineRegularizationLoss(kz, fused=False)(f.cuda()))
    ref = float(gold["kz%d" % kz]["loss"])
    print(kz, "fp64", l64, l64f, "fused32 err", abs(lf - l64) / abs(l64), "composed32 err", abs(lc - l64) / abs(l64), "ref32 err", abs(ref - l64) / abs(l64))
flows = [fill.flow_field(2, s, s, "reg_flow%d" % s) for s in (128, 64, 32)]
m64 = float(MultiAffineRegularizationLoss({1: 7, 2: 5, 3: 3}, fused=False)([x.double().cuda() for x in flows[::-1]]))
for fused in (True, False):
    m = float(MultiAffineRegularizationLoss({1: 7, 2: 5, 3: 3}, fused=fused)([x.cuda() for x in flows[::-1]]))
    print("multi fused", fused, abs(m - m64) / abs(m64), "ref", abs(float(gold["multi"]) - m64) / abs(m64))
