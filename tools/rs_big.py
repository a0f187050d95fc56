"""resample2d at an HBM-resident shape ([8,64,512,512], ks=4, flow ~ U[-3,3) px): the kernel the PMC passes look at."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ffwm_amd import ops
g = torch.Generator().manual_seed(0)
B = int(os.environ.get("RS_B", "8"))
in1 = torch.rand(B, 64, 512, 512, generator=g).cuda()
in2 = torch.cat((torch.rand(B, 2, 512, 512, generator=g) * 6 - 3, torch.full((B, 1, 512, 512), 2.0)), 1).cuda()
o = torch.empty_like(in1)
for _ in range(int(os.environ.get("RS_REPS", "5"))):
    ops.resample2d_forward(in1, in2, 4, 1, out=o)
torch.cuda.synchronize()
if os.environ.get("RS_BWD"):
    go = torch.rand(B, 64, 512, 512, generator=g).cuda()
    g1, g2 = torch.zeros_like(in1), torch.zeros_like(in2)
    for _ in range(3):
        ops.resample2d_backward(in1, in2, go, 4, 1, g1, g2)
    torch.cuda.synchronize()
