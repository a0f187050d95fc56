"""The reference's callers of the custom ops, restated on top of the HIP operator API.

``AffineRegularizationLoss`` / ``MultiAffineRegularizationLoss`` (/root/reference/models/losses.py:163-223)
are the only place the reference really runs BlockExtractor + LocalAttnReshape (FlowNet pre-training,
models/flownet_model.py:30-31,67-68): on 1-channel pixel-coordinate grids, with the constant flow kz//2,
kz = 3 / 5 / 7 on the 32 / 64 / 128 px flows.  Same constructor arguments, same call, same arithmetic;
the two ops are the gfx950 kernels behind ``ffwm_amd.external_function``.  ``fused=True`` evaluates the
same loss and its gradient with ONE kernel per scale (csrc/affine_reg.hip) instead of
6 launches forward + ~10 backward per coordinate grid.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import ops
from .external_function import BlockExtractor, LocalAttnReshape, Resample2d


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


def affine_residual_kernels(kz):
    """The kz^2 filters [kz^2, 1, kz, kz] (float64) whose responses, dotted with a window of the sampling grid, give the squared
    distance of that window from the nearest affine map (what losses.py:192-199 builds).  With the design matrix D of the window's
    kz^2 cells -- one row (y, x, 1) per cell, y-major -- the least-squares affine fit of a window g is P g, P = D (D^T D)^-1 D^T,
    and the residual is R g with R = I - P.  R is a symmetric projector, so the reference's K^T K (K = P - I) is R itself up to
    rounding; it is formed as R^T R here to follow the reference's arithmetic to the last bits the fixtures pin."""
    ys, xs = torch.meshgrid(torch.arange(kz, dtype=torch.float64), torch.arange(kz, dtype=torch.float64), indexing="ij")
    design = torch.stack((ys.reshape(-1), xs.reshape(-1), torch.ones(kz * kz, dtype=torch.float64)), 1)          # [kz^2, 3]
    fit = design @ torch.linalg.solve(design.T @ design, design.T)                                                # P
    resid = fit - torch.eye(kz * kz, dtype=torch.float64)                                                         # -R
    return (resid.T @ resid).reshape(kz * kz, 1, kz, kz)


class AffineRegularizationLoss(nn.Module):
    """Penalises the deviation of every kz x kz window of the sampling grid from an affine map
    (losses.py:181-223).  kernel = K^T K with K = A (A^T A)^-1 A^T - I, A = [row, col, 1]."""

    def __init__(self, kz, fused=False):
        super().__init__()
        self.kz = kz
        self.fused = fused        # True: the whole loss (and its gradient) as one HIP kernel per scale
        self.extractor = BlockExtractor(kernel_size=kz)
        self.reshape = LocalAttnReshape()
        self.kernel = affine_residual_kernels(kz)
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

    @staticmethod
    def _patch_products(grid, weights):
        """F.conv2d(grid, weights) of the reference (models/losses.py:196: ONE input channel, kz^2 output channels) as im2col +
        matmul on the GPU: with one input channel the vendor library's heuristics pick an NHWC implicit-GEMM data-gradient kernel
        (igemm_bwd_gtcx35_nhwc_fp32 ... bt256x32x4) that reads out of bounds -- a GPU memory fault whenever the neighbouring
        page is unmapped (seen under rocgdb after a trainer had shaped the allocator).  Same sums, rocBLAS instead."""
        if not grid.is_cuda:
            return F.conv2d(grid, weights)
        b, _, h, w = grid.shape
        kz = weights.shape[-1]
        cols = F.unfold(grid, kz)                                    # [b, kz*kz, (h - kz + 1) (w - kz + 1)]
        return torch.matmul(weights.reshape(weights.shape[0], -1), cols).view(b, weights.shape[0], h - kz + 1, w - kz + 1)

    def calculate_loss(self, grid, weights):
        results = self._patch_products(grid, weights)                # K^T K patch: [b, kz*kz, h, w]
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


# ------------------------------------------------------------------------------------------------
# The other two losses of FlowNet pre-training (models/flownet_model.py:64-72)
class LandmarkLoss(nn.Module):
    """MSE between the flow sampled at the frontal landmarks and the (normalised) profile landmarks
    (losses.py:61-74).  flow [B,2,s,s]; lm_S / lm_F integer pixel coordinates [B,L,2]; gate [B,L,2]."""

    def forward(self, flow, lm_S, lm_F, gate):
        side = flow.size(-1)
        # the flow vectors at the frontal landmarks: pixel (x, y) of the row-major plane is element y * side + x
        cell = (lm_F[..., 1] * side + lm_F[..., 0]).unsqueeze(1).expand(-1, 2, -1)             # [B, 2, L]
        picked = flow.flatten(2).gather(2, cell).transpose(1, 2)                                # [B, L, 2] = (flow_x, flow_y) per landmark
        wanted = lm_S.to(flow.dtype) * (2.0 / side) - 1                                         # profile landmarks on the [-1, 1] grid
        return F.mse_loss(picked * gate, wanted * gate)


class MultiScaleLDLoss(nn.Module):
    """losses.py:114-126: LandmarkLoss on the 128 / 64 / 32 px flows, weights 1000 / 1000 / 1500.  The integer
    landmarks are divided by the scale with INTEGER division: that is what ``lm.div(scale)`` did for integer
    tensors in the reference's pinned PyTorch 1.5.0 (README.md:14); from 1.6 on ``div`` is true division and
    the reference's own ``torch.gather`` call rejects the float index."""

    def __init__(self):
        super().__init__()
        self.criterionLD = LandmarkLoss()
        self.weights = [1000, 1000, 1500]
        self.img_size = 128

    def forward(self, flows, lm_S, lm_F, gate):
        ld_loss = 0
        for i, flow in enumerate(flows):
            scale = self.img_size // flow.size(3)
            lms = torch.div(lm_S, scale, rounding_mode="floor") if scale > 1 else lm_S
            lmf = torch.div(lm_F, scale, rounding_mode="floor") if scale > 1 else lm_F
            ld_loss = ld_loss + self.weights[i] * self.criterionLD(flow, lms, lmf, gate)
        return ld_loss


class PerceptualCorrectness(nn.Module):
    """losses.py:322-396 (the sampling-correctness loss of Global-Flow-Local-Attention) on the bilinear
    path the reference really takes (``use_bilinear_sampling=True``): for every flow scale,
    exp(-cos(warp(source_vgg, flow), target_vgg) / max_j cos(source_vgg_j, target_vgg)) averaged under the
    mask.  ``vgg`` maps an image to {'relu1_1': ..., ...}; ``warp`` is WarpNet (HIP) or a stand-in.

    Result-preserving difference: the VGG features and the N^2 x N^2 correlation maximum do not depend on
    the flow, and neither the images nor VGG are trained, so they are evaluated under no_grad -- the
    reference back-propagates through a [B, N^2, N^2] matrix (1 GiB per sample at relu1_1) for nothing."""

    def __init__(self, vgg, warp, layer=("relu1_1", "relu2_1", "relu3_1", "relu4_1"), resample=None):
        super().__init__()
        self.vgg = vgg
        self.warp = warp
        self.layer = list(layer)
        self.eps = 1e-8
        # losses.py:329: the Gaussian-weighted resampler of the `use_bilinear_sampling=False` branch (the one call site
        # of resample2d in the reference; its flow argument is handed over unscaled, exactly as the reference does)
        self.resample = resample if resample is not None else Resample2d(4, 1, sigma=2)

    def forward(self, target, source, flow_list, used_layers, norm_mask=None, use_bilinear_sampling=True):
        used_layers = sorted(used_layers, reverse=True)
        with torch.no_grad():
            self.target_vgg, self.source_vgg = self.vgg(target), self.vgg(source)
        loss = 0
        for i in range(len(flow_list)):
            loss = loss + self.calculate_loss(flow_list[i], self.layer[used_layers[i]], norm_mask, use_bilinear_sampling)
        return loss

    def calculate_loss(self, flow, layer, norm_mask=None, use_bilinear_sampling=False):       # the reference's default (:342)
        target_vgg = self.target_vgg[layer]
        source_vgg = self.source_vgg[layer]
        b, c, h, w = target_vgg.shape
        flow = F.interpolate(flow, [h, w])
        target_all = target_vgg.reshape(b, c, -1)                       # [b, C, N2]
        with torch.no_grad():
            source_all = source_vgg.reshape(b, c, -1).transpose(1, 2)   # [b, N2, C]
            source_norm = source_all / (source_all.norm(dim=2, keepdim=True) + self.eps)
            target_norm = target_all / (target_all.norm(dim=1, keepdim=True) + self.eps)
            # (the MFMA kernel wants >= 192 workgroups of 128 columns; below that rocBLAS + max is as fast)
            if source_norm.is_cuda and source_norm.dtype == torch.float32 and c in (64, 128, 256) and b * ((h * w + 127) // 128) >= 192:
                correction_max = ops.correlation_colmax(source_norm, target_norm)   # MFMA, no [b, N2, N2] matrix
            else:
                correction_max = torch.bmm(source_norm, target_norm).max(dim=1)[0]  # [b, N2]
        if use_bilinear_sampling:                                                  # losses.py:356-357
            input_sample = self.warp(source_vgg, flow).reshape(b, c, -1)
        else:                                                                      # losses.py:358-359 -> resample2d
            input_sample = self.resample(source_vgg.contiguous(), flow.contiguous()).reshape(b, c, -1)
        correction_sample = F.cosine_similarity(input_sample, target_all)          # [b, N2]
        loss_map = torch.exp(-correction_sample / (correction_max + self.eps))
        e1 = _exp_minus_one(loss_map.dtype)      # exp(-1) rounded in the map's dtype, as a host constant (no H2D copy: capturable)
        if norm_mask is None:
            return torch.mean(loss_map) - e1
        norm_mask = F.interpolate(norm_mask, size=(h, w)).reshape(-1, h * w)
        return (torch.sum(norm_mask * loss_map) - e1) / (torch.sum(norm_mask) + self.eps)


_E1 = {}


def _exp_minus_one(dtype):
    if dtype not in _E1:
        _E1[dtype] = float(torch.exp(torch.tensor(-1.0, dtype=dtype)))
    return _E1[dtype]


# ================================================================================= fused L1 terms (csrc/l1_loss.hip)
class _L1Terms(torch.autograd.Function):
    """out[n_slots] = sum over the terms of weight * mean |x m - y m| (models/ffwm_model.py:107-139: the pixel, perceptual, illumination
    and identity terms of backward_G), ONE launch forward and ONE backward for all terms instead of 5-7 element-wise / reduction
    launches per term and direction.  y and the masks are data / detached features in every term of the reference: no gradient."""

    @staticmethod
    def forward(ctx, n_slots, meta, *xs):
        from . import ops
        xs = [x.contiguous() for x in xs]
        terms = []
        for x, (y, m, segments) in zip(xs, meta):
            row = x.numel() // max(x.shape[0], 1)
            terms.append((x, y.contiguous(), None if m is None else m.contiguous(),
                          [(x0, y0, rows, w / float(max(rows * row, 1)), slot) for (x0, y0, rows, w, slot) in segments]))
        # saved through autograd (not parked on ctx): an in-place change of x, y or a mask between forward and backward is then caught by
        # the version check instead of silently producing the gradient of other data
        flat, layout = [], []
        for (x, y, m, segs) in terms:
            layout.append((len(flat), m is not None, segs))
            flat += [x, y] + ([m] if m is not None else [])
        ctx.save_for_backward(*flat)
        ctx.layout = layout
        return ops.l1_multi_forward(terms, n_slots)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import ops
        saved = ctx.saved_tensors
        terms = [(saved[i], saved[i + 1], saved[i + 2] if has_m else None, segs) for (i, has_m, segs) in ctx.layout]
        grads = ops.l1_multi_backward(terms, grad_out, ctx.needs_input_grad[2:])
        return (None, None) + tuple(grads)


def l1_terms(terms, n_slots):
    """terms: [(x, y, mask or None, weight, slot)] or [(x, y, mask, [(x_row0, y_row0, rows, weight, slot), ...])] -> vector [n_slots]:
    out[slot] = sum of weight * F.l1_loss(x[rows] * mask, y[rows] * mask) over the terms / segments routed to that slot."""
    xs, meta = [], []
    for t in terms:
        if len(t) == 5:
            x, y, m, w, slot = t
            segments = [(0, 0, x.shape[0], float(w), int(slot))]
        else:
            x, y, m, segments = t
        xs.append(x)
        meta.append((y, m, segments))
    return _L1Terms.apply(n_slots, meta, *xs)
