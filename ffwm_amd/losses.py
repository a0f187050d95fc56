"""The reference's callers of the custom ops, restated on top of the HIP operator API.

``AffineRegularizationLoss`` / ``MultiAffineRegularizationLoss`` (/root/reference/models/losses.py:163-223)
are the only place the reference really runs BlockExtractor + LocalAttnReshape (FlowNet pre-training,
models/flownet_model.py:30-31,67-68): on 1-channel pixel-coordinate grids, with the constant flow kz//2,
kz = 3 / 5 / 7 on the 32 / 64 / 128 px flows.  Same constructor arguments, same call, same arithmetic;
the two ops are the gfx950 kernels behind ``ffwm_amd.external_function``.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .external_function import BlockExtractor, LocalAttnReshape


class AffineRegularizationLoss(nn.Module):
    """Penalises the deviation of every kz x kz window of the sampling grid from an affine map
    (losses.py:181-223).  kernel = K^T K with K = A (A^T A)^-1 A^T - I, A = [row, col, 1]."""

    def __init__(self, kz):
        super().__init__()
        self.kz = kz
        self.extractor = BlockExtractor(kernel_size=kz)
        self.reshape = LocalAttnReshape()
        temp = np.arange(kz)
        A = np.ones([kz * kz, 3])
        A[:, 0] = temp.repeat(kz)
        A[:, 1] = temp.repeat(kz).reshape((kz, kz)).transpose().reshape(kz ** 2)
        AH = A.transpose()
        k = np.dot(A, np.dot(np.linalg.inv(np.dot(AH, A)), AH)) - np.identity(kz ** 2)
        kernel = np.dot(k.transpose(), k)
        self.kernel = torch.from_numpy(kernel).unsqueeze(1).view(kz ** 2, kz, kz).unsqueeze(1)

    def forward(self, flow_fields):
        grid = self.flow2grid(flow_fields)
        grid_x = grid[:, 0, :, :].unsqueeze(1)
        grid_y = grid[:, 1, :, :].unsqueeze(1)
        weights = self.kernel.type_as(flow_fields)
        return self.calculate_loss(grid_x, weights) + self.calculate_loss(grid_y, weights)

    def calculate_loss(self, grid, weights):
        results = F.conv2d(grid, weights)                            # K^T K patch: [b, kz*kz, h, w]
        b, c, h, w = results.size()
        kernels_new = self.reshape(results, self.kz)                 # HIP local_attn_reshape
        f = torch.zeros(b, 2, h, w).type_as(kernels_new) + float(int(self.kz / 2))
        grid_H = self.extractor(grid, f)                             # HIP block_extractor
        result = F.avg_pool2d(grid_H * kernels_new, self.kz, self.kz)
        return torch.mean(result) * self.kz ** 2

    @staticmethod
    def flow2grid(flow_field):
        return flow_field.add(1.0).div(2.0).mul(128.0)


class MultiAffineRegularizationLoss(nn.Module):
    """losses.py:163-179: one AffineRegularizationLoss per flow scale; ``kz_dic`` maps layer -> kz and the
    flows are matched to the layers in DESCENDING layer order (flownet_model.py:31 builds {1: 7, 2: 5, 3: 3}
    and :68 passes the flows smallest first)."""

    def __init__(self, kz_dic):
        super().__init__()
        self.kz_dic = kz_dic
        self.method_dic = {key: AffineRegularizationLoss(kz_dic[key]) for key in kz_dic}
        self.layers = sorted(kz_dic, reverse=True)

    def forward(self, flow_fields):
        loss = 0
        for i in range(len(flow_fields)):
            loss = loss + self.method_dic[self.layers[i]](flow_fields[i])
        return loss
