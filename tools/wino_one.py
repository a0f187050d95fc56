import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops
B, C, H, K = 8, int(os.environ.get("CV_C", 192)), int(os.environ.get("CV_H", 128)), int(os.environ.get("CV_K", 192))
x = torch.randn(B, C, H, H, device="cuda"); w = torch.randn(K, C, 3, 3, device="cuda") * 0.05; b = torch.randn(K, device="cuda")
for _ in range(5): ops.conv3x3_winograd(x, w, b)
torch.cuda.synchronize()
