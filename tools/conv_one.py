import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import flownet_eval as fe
B, C, H, K, k, s, tr = 6, int(os.environ.get("CV_C", 64)), int(os.environ.get("CV_H", 128)), int(os.environ.get("CV_K", 64)), 3, int(os.environ.get("CV_S", 2)), False
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, k, k, device="cuda") * 0.01; b = torch.randn(K, device="cuda")
for _ in range(5): fe.conv_mfma(x, w, b, s, 1, tr, fe.LRELU, 0.2)
torch.cuda.synchronize()
