import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from ffwm_amd import ops
torch.manual_seed(0)
B,C,H,W,K = 1,66,8,12,65
x = torch.randn(B,C,H,W,device="cuda"); w = torch.randn(K,C,3,3,device="cuda")*0.1
y = ops.conv3x3_winograd(x,w,None)[:, 64:65].double()
xd, wd = x.double(), w.double()
cands = {"ref": F.conv2d(xd, wd[64:65], None, 1, 1), "flip": F.conv2d(xd, wd[64:65].flip(2,3), None, 1, 1),
         "k0": F.conv2d(xd, wd[0:1], None, 1, 1), "transposed": F.conv2d(xd, wd[64:65].transpose(2,3), None, 1, 1)}
for q in range(4):
    m = torch.zeros(C, dtype=torch.bool); m[q::4] = True
    cands["chan%d" % q] = F.conv2d(xd[:, m.cuda()], wd[64:65][:, m.cuda()], None, 1, 1)
for n, r in cands.items():
    print(n, (y - r).abs().max().item())
print(y[0,0,:2,:6]); print(cands["ref"][0,0,:2,:6])
