"""One shape of the Winograd-domain weight gradient (netG's 192 -> 192 body at 128 x 128, batch 8) for counter passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ffwm_amd import ops
C = int(os.environ.get("CV_C", 192)); K = int(os.environ.get("CV_K", 192)); H = int(os.environ.get("CV_H", 128))
x = torch.randn(8, C, H, H, device="cuda"); go = torch.randn(8, K, H, H, device="cuda")
dw = torch.zeros(K, C, 3, 3, device="cuda")
for _ in range(5):
    dw.zero_()
    ops.conv3x3_wgrad(x, go, dw, None)
torch.cuda.synchronize()
