"""Drop-in mirror of the reference's operator API for the flow-warp hot path.

Same names, call signatures and error behaviour as /root/reference/models/external_function.py:19-158
(``BlockExtractorFunction``, ``LocalAttnReshapeFunction``, ``Resample2dFunction`` and their ``nn.Module``
shells), so ``models/losses.py:161`` style imports keep working::

    from ffwm_amd.external_function import Resample2d, LocalAttnReshape, BlockExtractor

Underneath, every call goes through the C ABI of libffwm_hip.so (hand-written gfx950 kernels).
Differences from the reference, all deliberate:
  * outputs are allocated uninitialised where the kernel overwrites them (the reference zero-fills
    and then overwrites); gradients that are scattered into are still zero-filled;
  * ``grad_output`` is made contiguous for real in these Functions (the reference calls ``.contiguous()`` and drops
    the result, external_function.py:46-47,93-94,132-133): a copy + the tuned kernels beats reading a view element by
    element.  The C ABI itself takes a strided grad_output since ABI 5 (ffwm_*_backward_strided; ``ops.*_backward`` pass a
    non-contiguous tensor's strides through), which is what ``ffwm_amd.compat`` -- the pybind-name shims -- relies on;
  * gradients nobody asked for (``ctx.needs_input_grad``) are not computed;
  * a device guard: kernels launch on the tensors' device and its current stream.
Plus ``WarpNet`` (models/base_networks.py:168-173) and the fused warp + flip + concat that
FFWM.forward performs at models/base_networks.py:326-329.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops


def _require_cuda(t):
    if not t.is_cuda:
        # the reference raises the same for CPU tensors (external_function.py:37-38,84-85)
        raise NotImplementedError


class BlockExtractorFunction(Function):
    """apply(source[B,C,Hs,Ws], flow_field[B,2,Hf,Wf], kernel_size) -> [B,C,k*Hf,k*Wf]"""

    @staticmethod
    def forward(ctx, source, flow_field, kernel_size):
        assert source.is_contiguous()
        assert flow_field.is_contiguous()
        assert flow_field.size(1) == 2
        _require_cuda(source)
        ctx.save_for_backward(source, flow_field)
        ctx.kernel_size = kernel_size
        return ops.block_extractor_forward(source, flow_field, kernel_size)

    @staticmethod
    def backward(ctx, grad_output):
        source, flow_field = ctx.saved_tensors
        need_src, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_source = torch.zeros_like(source) if need_src else None
        grad_flow = torch.zeros_like(flow_field) if need_flow else None
        if need_src or need_flow:
            ops.block_extractor_backward(source, flow_field, grad_output.contiguous(), ctx.kernel_size,
                                         grad_source, grad_flow)
        return grad_source, grad_flow, None


class BlockExtractor(nn.Module):
    def __init__(self, kernel_size=3):
        super().__init__()
        self.kernel_size = kernel_size

    def forward(self, source, flow_field):
        return BlockExtractorFunction.apply(source.contiguous(), flow_field.contiguous(), self.kernel_size)


class BlockAttentionFunction(Function):
    """apply(source[B,C,Hs,Ws], flow_field[B,2,Hf,Wf], weights[B,k*k,Hf,Wf], kernel_size) -> [B,C,Hf,Wf]

    The fused extractor + attention consumer: equal to
    ``F.avg_pool2d(BlockExtractor(k)(source, flow) * LocalAttnReshape()(weights, k), k, k)`` -- the composition
    of the reference's ops a GFLA-style local attention runs -- without materialising the k^2-fold tensors."""

    @staticmethod
    def forward(ctx, source, flow_field, weights, kernel_size):
        assert source.is_contiguous() and flow_field.is_contiguous() and weights.is_contiguous()
        assert flow_field.size(1) == 2
        assert weights.size(1) == kernel_size * kernel_size
        _require_cuda(source)
        ctx.save_for_backward(source, flow_field, weights)
        ctx.kernel_size = kernel_size
        return ops.block_attention_forward(source, flow_field, weights, kernel_size)

    @staticmethod
    def backward(ctx, grad_output):
        source, flow_field, weights = ctx.saved_tensors
        need = ctx.needs_input_grad
        grad_source = torch.zeros_like(source) if need[0] else None
        grad_flow = torch.zeros_like(flow_field) if need[1] else None
        grad_weights = torch.zeros_like(weights) if need[2] else None
        if need[0] or need[1] or need[2]:
            # the fast path produces d(source) and d(flow) in one pass: ask for d(source) whenever d(flow) is wanted
            gs = grad_source if grad_source is not None or not need[1] else torch.zeros_like(source)
            ops.block_attention_backward(source, flow_field, weights, grad_output.contiguous(), ctx.kernel_size,
                                         gs, grad_flow, grad_weights)
        return grad_source, grad_flow, grad_weights, None


class BlockAttention(nn.Module):
    """``softmax=True`` normalises the k*k attention logits over dim 1 first (a [B,k*k,H,W] torch op: 1/C-th of
    the data the fused kernels move)."""

    def __init__(self, kernel_size=3, softmax=False):
        super().__init__()
        self.kernel_size = kernel_size
        self.softmax = softmax

    def forward(self, source, flow_field, attn):
        weights = torch.softmax(attn, 1) if self.softmax else attn
        return BlockAttentionFunction.apply(source.contiguous(), flow_field.contiguous(), weights.contiguous(),
                                            self.kernel_size)


class MaxFeatureMapFunction(Function):
    """apply(x[B, 2C, ...], bias=None) -> max(x[:, :C] + bias[:C], x[:, C:] + bias[C:]): LightCNN's mfm activation
    (lightcnn/light_cnn.py `mfm.forward`: torch.split + torch.max) with the bias of the layer in front folded in,
    forward and backward one kernel each; ties share the gradient like ATen's maximum."""

    @staticmethod
    def forward(ctx, x, bias=None):
        _require_cuda(x)
        assert x.is_contiguous() and x.size(1) % 2 == 0
        ctx.save_for_backward(x, bias)
        return ops.mfm_forward(x, bias)

    @staticmethod
    def backward(ctx, grad_y):
        x, bias = ctx.saved_tensors
        dx = ops.mfm_backward(x, grad_y.contiguous(), bias)
        db = None
        if bias is not None and ctx.needs_input_grad[1]:
            db = dx.sum(dim=[0] + list(range(2, dx.dim())))
        return dx, db


class BiasReLUFunction(Function):
    """apply(h[B, C, ...], bias[C]) -> relu(h + bias): the bias add and nn.ReLU behind a convolution as one pass."""

    @staticmethod
    def forward(ctx, h, bias):
        _require_cuda(h)
        y = ops.bias_relu_forward(h, bias)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        y, = ctx.saved_tensors
        gh = torch.ops.aten.threshold_backward(grad_y, y, 0)
        gb = gh.sum(dim=[0] + list(range(2, gh.dim()))) if ctx.needs_input_grad[1] else None
        return gh, gb


class LocalAttnReshapeFunction(Function):
    """apply(inputs[B,k*k,H,W], kernel_size) -> [B,1,k*H,k*W]"""

    @staticmethod
    def forward(ctx, inputs, kernel_size):
        assert inputs.is_contiguous()
        assert inputs.size(1) == kernel_size * kernel_size
        _require_cuda(inputs)
        ctx.kernel_size = kernel_size
        return ops.local_attn_reshape_forward(inputs, kernel_size)

    @staticmethod
    def backward(ctx, grad_output):
        if not ctx.needs_input_grad[0]:
            return None, None
        # the map is a bijection: write the inverse permutation, no zero-fill, no atomics
        return ops.local_attn_reshape_backward(grad_output.contiguous(), ctx.kernel_size), None


class LocalAttnReshape(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, inputs, kernel_size=3):
        return LocalAttnReshapeFunction.apply(inputs.contiguous(), kernel_size)


class Resample2dFunction(Function):
    """apply(input1[B1,C,Hi,Wi], input2[B,3,H,W]=(dx,dy,sigma), kernel_size=2, dilation=1) -> [B,C,H,W]

    ``reference_quirk`` (class attribute, default True) keeps the reference's ``int()`` truncation in
    the d_input1 weights (resample2d_kernel.cu:137-138); set it False for the true gradient."""

    reference_quirk = True

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=2, dilation=1):
        assert input1.is_contiguous()
        assert input2.is_contiguous()
        _require_cuda(input1)
        ctx.save_for_backward(input1, input2)
        ctx.kernel_size = kernel_size
        ctx.dilation = dilation
        return ops.resample2d_forward(input1, input2, kernel_size, dilation)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # the reference zero-fills grad_input1 for its atomics (external_function.py:137); here the library is told that the buffer is
        # uninitialised (owned tiles store every cell once; the other paths clear it themselves) -- unless input1 has more samples than
        # input2 (the extra ones get no gradient and must read zero)
        fresh = need1 and input1.size(0) == input2.size(0)
        grad_input1 = (torch.empty_like(input1) if fresh else torch.zeros_like(input1)) if need1 else None
        grad_input2 = torch.empty_like(input2) if need2 else None
        if need1 or need2:
            ops.resample2d_backward(input1, input2, grad_output.contiguous(), ctx.kernel_size, ctx.dilation,
                                    grad_input1, grad_input2, Resample2dFunction.reference_quirk, overwrite_input1=fresh)
        return grad_input1, grad_input2, None, None


class Resample2d(nn.Module):
    """Resample2d(kernel_size, dilation, sigma)(input1, flow[B,2,H,W]): appends the constant sigma
    channel (external_function.py:154-158).  sigma follows the flow's device, so -- unlike the
    reference module -- it also works for GPU tensors."""

    def __init__(self, kernel_size=2, dilation=1, sigma=5):
        super().__init__()
        self.kernel_size = kernel_size
        self.dilation = dilation
        self.sigma = float(sigma)

    def forward(self, input1, input2):
        b, _, h, w = input2.shape
        sigma = input2.new_full((b, 1, h, w), self.sigma)
        packed = torch.cat((input2, sigma), 1)
        return Resample2dFunction.apply(input1.contiguous(), packed, self.kernel_size, self.dilation)


class WarpFunction(Function):
    """apply(feat[B,C,Hi,Wi], flow[B,2,H,W] in [-1,1], flipcat=False): F.grid_sample(feat,
    flow.permute(0,2,3,1)) -- bilinear, zeros, align_corners=False -- and, with flipcat, the
    cat((w, flip(w, 3)), 1) of FFWM.forward in the same kernel."""

    @staticmethod
    def forward(ctx, feat, flow, flipcat=False):
        _require_cuda(feat)
        feat = feat.contiguous()
        flow = flow.contiguous()
        ctx.save_for_backward(feat, flow)
        ctx.flipcat = bool(flipcat)
        return ops.warp_forward(feat, flow, ctx.flipcat)

    @staticmethod
    def backward(ctx, grad_output):
        feat, flow = ctx.saved_tensors
        need_feat, need_flow = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        grad_feat = torch.empty_like(feat) if need_feat else None          # produced whole by the library (overwrite_feat)
        grad_flow = torch.zeros_like(flow) if need_flow else None
        if need_feat or need_flow:
            ops.warp_backward(feat, flow, grad_output.contiguous(), ctx.flipcat, grad_feat, grad_flow, overwrite_feat=True)
        return grad_feat, grad_flow, None


class WarpMultiFunction(Function):
    """apply(flipcat, n, feat_0..feat_{n-1}, flow_0..flow_{n-1}) -> n warped tensors: n independent WarpFunction calls
    as one forward launch, one d(flow) launch and one d(feat) launch per plane size (csrc/warp.hip, multi-problem
    tables).  autograd runs the backward once, when the gradients of all n outputs are known."""

    @staticmethod
    def forward(ctx, flipcat, n, *tensors):
        feats = [t.contiguous() for t in tensors[:n]]
        flows = [t.contiguous() for t in tensors[n:]]
        _require_cuda(feats[0])
        ctx.save_for_backward(*(feats + flows))
        ctx.flipcat, ctx.n = bool(flipcat), n
        return tuple(ops.warp_multi_forward(feats, flows, ctx.flipcat))

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.n
        saved = ctx.saved_tensors
        feats, flows = list(saved[:n]), list(saved[n:])
        need = ctx.needs_input_grad[2:]
        # every wanted gradient is a slice of ONE zero-filled buffer (the kernels accumulate): one fill instead of up to 2 n
        sizes = [feats[i].numel() if need[i] else 0 for i in range(n)] + [flows[i].numel() if need[n + i] else 0 for i in range(n)]
        offs, tot = [], 0
        for sz in sizes:
            offs.append(tot)
            tot += (sz + 63) // 64 * 64                    # 256-byte aligned slices
        buf = torch.zeros(max(tot, 1), device=feats[0].device, dtype=feats[0].dtype)
        gfe = [buf[offs[i]:offs[i] + sizes[i]].view_as(feats[i]) if need[i] else None for i in range(n)]
        gfl = [buf[offs[n + i]:offs[n + i] + sizes[n + i]].view_as(flows[i]) if need[n + i] else None for i in range(n)]
        gos = [g.contiguous() if g is not None else torch.zeros(
            (feats[i].size(0), (2 if ctx.flipcat else 1) * feats[i].size(1)) + tuple(flows[i].shape[2:]),
            device=feats[i].device, dtype=feats[i].dtype) for i, g in enumerate(grads)]
        if any(need):
            ops.warp_multi_backward(feats, flows, gos, ctx.flipcat, gfe, gfl)
        return (None, None) + tuple(gfe) + tuple(gfl)


def warp_many(feats, flows, flipcat=False):
    """[WarpNet()(f, fl) for f, fl in zip(feats, flows)] (or the flip + cat form) as one multi-problem call."""
    feats, flows = list(feats), list(flows)
    return list(WarpMultiFunction.apply(flipcat, len(feats), *(feats + flows)))


class WarpNet(nn.Module):
    """Same call as the reference's WarpNet (models/base_networks.py:168-173); bilinear only."""

    def forward(self, images, flow, mode='bilinear'):
        if mode != 'bilinear':
            raise NotImplementedError("WarpNet: only mode='bilinear' is implemented (every reference caller uses it)")
        return WarpFunction.apply(images, flow, False)


class WarpFlipCat(nn.Module):
    """w = WarpNet(feat, flow); return cat((w, flip(w, (3,))), 1) -- models/base_networks.py:326-329."""

    def forward(self, feat, flow):
        return WarpFunction.apply(feat, flow, True)


class GuidedFilterFunction(Function):
    """apply(x[B,C,H,W], y[B,C,H,W], r, eps) -> GuidedFilter(r, eps)(x, y) of the reference
    (models/external_function.py:239-277), four launches forward and four backward (separable column / row passes over the whole chip).  y is data (FFWM filters
    the generated image against the ground truth, models/ffwm_model.py:81): it gets no gradient."""

    @staticmethod
    def forward(ctx, x, y, r, eps):
        _require_cuda(x)
        x = x.contiguous()
        y = y.contiguous()
        out, saved = ops.guided_filter_forward(x, y, r, eps)
        ctx.save_for_backward(x, y, saved)
        ctx.r = r
        return out

    @staticmethod
    def backward(ctx, grad_output):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("GuidedFilterFunction: no gradient for y (the guidance target is data)")
        x, y, saved = ctx.saved_tensors
        gx = ops.guided_filter_backward(x, y, saved, grad_output.contiguous(), ctx.r) if ctx.needs_input_grad[0] else None
        return gx, None, None, None


class GuidedFilter(nn.Module):
    """Same constructor and call as the reference's GuidedFilter(r, eps=1e-8)(x, y)
    (models/external_function.py:239-277) for c_x == c_y, H, W <= 128."""

    def __init__(self, r, eps=1e-8):
        super().__init__()
        self.r = r
        self.eps = eps

    def forward(self, x, y):
        n_x, c_x, h_x, w_x = x.size()
        n_y, c_y, h_y, w_y = y.size()
        assert n_x == n_y
        assert h_x == h_y and w_x == w_y
        assert h_x > 2 * self.r + 1 and w_x > 2 * self.r + 1
        if c_x != c_y:
            raise NotImplementedError("GuidedFilter: the HIP kernel needs c_x == c_y (FFWM's only usage)")
        return GuidedFilterFunction.apply(x, y, self.r, self.eps)
