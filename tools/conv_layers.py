"""Per-layer timing of csrc/conv_fwd.hip on FlowNet(64)'s layers at batch 6 (HIP events, warm), next to MIOpen."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from ffwm_amd import flownet_eval as fe, _lib
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B = int(os.environ.get("B", 6))
if os.environ.get("FFWM_SPLIT_TARGET"):
    _lib.set_option("conv_fwd_split_target", int(os.environ["FFWM_SPLIT_TARGET"]))
layers = [("conv1", 64, 128, 64, 3, 2, False), ("conv2", 128, 64, 128, 3, 2, False), ("conv3", 128, 32, 256, 3, 2, False),
          ("conv4", 256, 16, 512, 3, 2, False), ("conv4_1", 512, 8, 512, 3, 1, False), ("conv5", 512, 8, 512, 3, 2, False),
          ("conv5_1", 512, 4, 512, 3, 1, False), ("conv6", 512, 4, 1024, 3, 2, False), ("conv6_1", 1024, 2, 1024, 3, 1, False),
          ("deconv5", 1024, 2, 512, 4, 2, True), ("inter5", 1026, 4, 512, 3, 1, False), ("deconv4", 1026, 4, 256, 4, 2, True),
          ("inter4", 770, 8, 256, 3, 1, False), ("deconv3", 770, 8, 128, 4, 2, True), ("inter3", 386, 16, 128, 3, 1, False),
          ("deconv2", 386, 16, 64, 4, 2, True), ("inter2", 66, 32, 64, 3, 1, False), ("deconv1", 66, 32, 32, 4, 2, True),
          ("inter1", 34, 64, 32, 3, 1, False), ("deconv0", 34, 64, 16, 4, 2, True), ("inter0", 18, 128, 16, 3, 1, False),
          ("conv1_1", 64, 64, 128, 3, 1, False), ("conv2_1", 128, 32, 128, 3, 1, False), ("conv3_1", 256, 16, 256, 3, 1, False)]
tot_m = tot_v = 0
for name, C, H, K, k, s, tr in layers:
    x = torch.randn(B, C, H, H, device="cuda")
    w = torch.randn(*((C, K, k, k) if tr else (K, C, k, k)), device="cuda") * 0.01
    b = torch.randn(K, device="cuda")
    mine = t(lambda: fe.conv_mfma(x, w, b, s, 1, tr, fe.LRELU, 0.2))
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(10): fe.conv_mfma(x, w, b, s, 1, tr, fe.LRELU, 0.2)
    torch.cuda.synchronize(); _lib.prof_enable(False)
    kr = {k: v for k, v in _lib.prof_collect().items() if k.startswith("conv_fwd")}
    kus = list(kr.values())[0]["avg_ms"] * 1e3
    vend = t(lambda: fe.bias_act((F.conv_transpose2d if tr else F.conv2d)(x, w, None, s, 1), b, fe.LRELU))
    Ho = 2 * H if tr else (H + 2 - k) // s + 1
    gf = 2.0 * B * Ho * Ho * K * C * (4 if tr else k * k) / 1e9
    print("%-8s C=%4d H=%3d K=%4d k%d s%d %s  kernel %6.1f us (%5.1f TF)  call %7.1f us (%5.1f TF)   MIOpen+epilogue %7.1f us   %.2f GFLOP" % (name, C, H, K, k, s, "T" if tr else " ", kus, gf / kus * 1e3, mine, gf / mine * 1e3, vend, gf))
