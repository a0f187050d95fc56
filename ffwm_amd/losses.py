"""The reference's callers of the custom ops, restated on top of the HIP operator API.

``AffineRegularizationLoss`` / ``MultiAffineRegularizationLoss`` (/root/reference/models/losses.py:163-223)
are the only place the reference really runs BlockExtractor + LocalAttnReshape (FlowNet pre-training,
models/flownet_model.py:30-31,67-68): on 1-channel pixel-coordinate grids, with the constant flow kz//2,
kz = 3 / 5 / 7 on the 32 / 64 / 128 px flows.  Same constructor arguments, same call, same arithmetic;
the two ops are the gfx950 kernels behind ``ffwm_amd.external_function``.  ``fused=True`` evaluates the
same loss and its gradient with ONE kernel per scale (csrc/affine_reg.hip) instead of
6 launches forward + ~10 backward per coordinate grid.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .external_function import BlockExtractor, LocalAttnReshape


class _FusedAffineReg(Function):
    """loss and d(loss)/d(flow) from one launch (csrc/affine_reg.hip)."""

    @staticmethod
    def forward(ctx, flow, ktk, kz):
        loss, grad = ops.affine_regularization(flow, ktk, kz, want_grad=flow.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_output):
        (grad,) = ctx.saved_tensors
        return (grad * grad_output if grad is not None else None), None, None


class AffineRegularizationLoss(nn.Module):
    """Penalises the deviation of every kz x kz window of the sampling grid from an affine map
    (losses.py:181-223).  kernel = K^T K with K = A (A^T A)^-1 A^T - I, A = [row, col, 1]."""

    def __init__(self, kz, fused=False):
        super().__init__()
        self.kz = kz
        self.fused = fused        # True: the whole loss (and its gradient) as one HIP kernel per scale
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
        self._cache = {}

    def _weights(self, like):
        """self.kernel.type_as(flow) of the reference, cached per (device, dtype): the reference re-uploads
        the CPU tensor on every call (a synchronous host-to-device copy)."""
        key = (like.device, like.dtype)
        w = self._cache.get(key)
        if w is None:
            w = self._cache[key] = self.kernel.to(device=like.device, dtype=like.dtype)
        return w

    def forward(self, flow_fields):
        weights = self._weights(flow_fields)
        if self.fused:
            return _FusedAffineReg.apply(flow_fields, weights.view(self.kz ** 2, self.kz ** 2), self.kz)
        grid = self.flow2grid(flow_fields)
        grid_x = grid[:, 0, :, :].unsqueeze(1)
        grid_y = grid[:, 1, :, :].unsqueeze(1)
        return self.calculate_loss(grid_x, weights) + self.calculate_loss(grid_y, weights)

    def calculate_loss(self, grid, weights):
        results = F.conv2d(grid, weights)                            # K^T K patch: [b, kz*kz, h, w]
        b, c, h, w = results.size()
        kernels_new = self.reshape(results, self.kz)                 # HIP local_attn_reshape
        f = torch.full((b, 2, h, w), float(int(self.kz / 2)), dtype=kernels_new.dtype, device=kernels_new.device)
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

    def __init__(self, kz_dic, fused=False):
        super().__init__()
        self.kz_dic = kz_dic
        self.method_dic = {key: AffineRegularizationLoss(kz_dic[key], fused=fused) for key in kz_dic}
        self.layers = sorted(kz_dic, reverse=True)

    def forward(self, flow_fields):
        loss = 0
        for i in range(len(flow_fields)):
            loss = loss + self.method_dic[self.layers[i]](flow_fields[i])
        return loss
