"""Stock-PyTorch stand-ins used ONLY by tests to run the conv stacks on CPU (the product ops
refuse CPU tensors by design).  Identities from SURVEY D6."""
import torch
import torch.nn.functional as F


def warp(images, flow, mode="bilinear"):
    return F.grid_sample(images, flow.permute(0, 2, 3, 1), mode=mode, padding_mode="zeros",
                         align_corners=False)


def warp_flipcat(feat, flow):
    w = warp(feat, flow)
    return torch.cat((w, torch.flip(w, (3,))), 1)
