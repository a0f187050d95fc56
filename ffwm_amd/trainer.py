"""The FFWM GAN train step (BASELINE.json configs[2]/[3]) on synthetic MultiPIE-shaped tensors.

Restates the computation of /root/reference/models/ffwm_model.py:72-160 (``forward`` ->
``backward_D`` -> ``backward_G``, three Adam optimizers) with the losses of
/root/reference/models/losses.py (GANLoss 'lsgan' :7-58, IdentityLoss :76-112, MSL1Loss :130-157,
PerceptualLoss :293-320).  Every WarpNet call (image warps, part crops, the multi-scale
illumination loss) and netG's warp-attention go through the HIP kernels; conv stacks run on
PyTorch-ROCm.  Data parallelism: ``ffwm_amd.dp`` (one process per GPU, RCCL all-reduce of flat
gradient buckets overlapped with backward).

Deliberate, result-preserving differences from the reference:
  * LightCNN and VGG19 are frozen (``requires_grad=False``).  The reference leaves LightCNN's weights
    trainable although no optimizer owns them (SURVEY 3.2 "wasted weight-grads"); outputs and all
    used gradients are identical.
  * target-side feature extractors (VGG(y), LightCNN(gt)) run under ``no_grad``: the reference
    detaches their results.
  * pretrained VGG19 / LightCNN / FlowNet checkpoints cannot be fetched offline: seeded random
    weights of the same architectures (same FLOPs and bytes; loss values differ).
"""
import contextlib
import itertools
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import conv, nets
from .dp import BucketedGradReducer, broadcast_module_state
from .external_function import WarpNet

# FFWM_GRAD_ARENA=0: every weight-gradient call clears its own buffer again (conv._GradArena)
_GRAD_ARENA_ON = os.environ.get("FFWM_GRAD_ARENA", "1") != "0"

PRC_LAYERS = ("relu1_1", "relu2_1", "relu3_1", "relu4_1", "relu5_1")
PRC_WEIGHTS = (1.0, 1.0 / 2, 1.0 / 4, 1.0 / 4, 1.0 / 8)


def synthetic_batch(batch_size, device, seed=0, n_landmarks=1024):
    """MultiPIE-shaped synthetic inputs: images ~ U[0,1), centred disc masks
    r=48, integer landmarks in [16,111] (>= 580 of them, ffwm_model.py:222-224)."""
    g = torch.Generator().manual_seed(seed)
    img_S = torch.rand(batch_size, 3, 128, 128, generator=g)
    img_F = torch.rand(batch_size, 3, 128, 128, generator=g)
    yy, xx = torch.meshgrid(torch.arange(128.), torch.arange(128.), indexing="ij")
    disc = (((yy - 63.5) ** 2 + (xx - 63.5) ** 2) <= 48.0 ** 2).float().view(1, 1, 128, 128)
    mask = disc.repeat(batch_size, 1, 1, 1)
    lm_F = torch.randint(16, 112, (batch_size, n_landmarks, 2), generator=g)
    lm_S = torch.randint(16, 112, (batch_size, n_landmarks, 2), generator=g)
    gate = (torch.rand(batch_size, n_landmarks, 1, generator=g) > 0.2).float()     # landmark visibility (flownet_model.py:53-54)
    batch = {"img_S": img_S, "img_F": img_F, "mask_S": mask.clone(), "mask_F": mask, "lm_F": lm_F, "lm_S": lm_S,
             "gate": gate}
    return {k: v.to(device) for k, v in batch.items()}


def build_part_grid(lm, d):
    """Sampling grid (normalised, [B,2,d,d]) of a d x d patch centred on landmark ``lm`` [B,1,2]
    (ffwm_model.py:236-246)."""
    b = lm.size(0)
    r = d // 2
    lin = torch.linspace(-r, r, d, device=lm.device)
    base = torch.stack((lin.view(1, d).expand(d, d), lin.view(d, 1).expand(d, d)), 0)   # [2,d,d]: x, y
    centre = lm.float().view(b, 2, 1, 1) - 64.0
    return (base.unsqueeze(0) + centre) / 64.0


def part_grids(lm_F, torch15_integer_division=False):
    """Eye / nose / mouth centres from the landmark table (ffwm_model.py:217-234).

    The mouth centre is `(min + max) / 2` on integer landmarks.  On the PyTorch of this image (>= 1.6) that is TRUE
    division -- what the reference's code computes when imported here, and what tests/golden/reference_modules.pt pins
    (default).  Under the reference's pinned PyTorch 1.5 (README.md:14) the same line floor-divides: the grid moves by
    half a pixel whenever min + max is odd; `torch15_integer_division=True` reproduces that."""
    el, er = lm_F[:, 63:64], lm_F[:, 515:516]
    nc = lm_F[:, 429:430]
    mouth = torch.cat((lm_F[:, 64:128], lm_F[:, 516:580]), 1)
    mc = mouth.min(1, keepdim=True)[0] + mouth.max(1, keepdim=True)[0]
    mc = torch.div(mc, 2, rounding_mode="floor") if (torch15_integer_division and not mc.is_floating_point()) else mc / 2
    return [build_part_grid(c, 32) for c in (el, er, nc, mc)]


def probe_collective_capture(device, group=None):
    """Can this process group's all-reduce be captured into a hipGraph and replayed?  One tiny graph (an asynchronous all-reduce
    and its wait, the shape the reducers use) captured on a side stream, replayed on fresh data and checked; the ranks agree on the
    answer (MIN over ranks), so all of them pick the same capture mode.  Only RCCL is tried: gloo's collectives run on the host."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
        return False
    import time
    device = torch.device(device)
    world = dist.get_world_size(group)
    ok = True
    x = torch.ones(4096, device=device)
    try:
        dist.all_reduce(x, group=group)               # communicator set-up outside the capture
        torch.cuda.synchronize(device)
        time.sleep(0.3)                               # (the watchdog retires the finished work)
        x.fill_(1.0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode=os.environ.get("FFWM_CAPTURE_ERROR_MODE", "thread_local")):
            h = dist.all_reduce(x, group=group, async_op=True)
            h.wait()
            x.mul_(0.5)
        for _ in range(2):
            x.fill_(1.0)
            g.replay()
            torch.cuda.synchronize(device)
            ok = ok and abs(float(x[0]) - 0.5 * world) < 1e-6 and abs(float(x[-1]) - 0.5 * world) < 1e-6
        del g
    except Exception as e:                            # noqa: BLE001 -- any failure means "do not capture the collectives"
        import sys
        print("probe_collective_capture: %r" % (e,), file=sys.stderr)
        ok = False
    try:
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = float(flag.item()) > 0.5
    except Exception:                                 # noqa: BLE001
        ok = False
    return ok


class _FlowGate(torch.autograd.Function):
    """Identity on a flow net's outputs whose BACKWARD orders two streams: the gate of flowNetF (record=True) records an event on its
    stream when its backward runs -- autograd reaches it when d(flows_F) is complete, i.e. after netG's backward --, the gate of
    flowNetB (record=False) makes its stream wait for that event before flowNetB's backward starts.  The autograd engine pops ready
    nodes by descending sequence number: flowNetB's gate is the older node, so the host records the event before it issues the wait
    (hipGraph capture needs that order); if the event is missing (flows_F without gradient) the wait is skipped."""

    @staticmethod
    def forward(ctx, box, record, *flows):
        ctx.box, ctx.record = box, record
        return tuple(f.view_as(f) for f in flows)

    @staticmethod
    def backward(ctx, *grads):
        if grads and any(g is not None for g in grads):
            dev = next(g for g in grads if g is not None).device
            cur = torch.cuda.current_stream(dev)
            if ctx.record:
                ev = torch.cuda.Event()
                ev.record(cur)
                ctx.box["ev"] = ev
            elif ctx.box.get("ev") is not None:
                cur.wait_event(ctx.box["ev"])
        return (None, None) + tuple(grads)


class FFWMTrainer(object):
    def __init__(self, device, world_size=1, seed=0, titers=0, bucket_bytes=64 << 20, warp=None,
                 warp_flipcat=None, ngf=64, capturable=False, fused_spectral_norm=None, batched_losses=True,
                 mfma_wgrad=None, flat_adam=None, fused_bn=None, mfma_fwd=None, fused_l1=None, segmented_backward=False,
                 force_collectives=False):
        self.device = torch.device(device)
        self.titers = titers
        # data parallelism is live when there are several ranks (force_collectives: also in a one-rank process group -- the
        # one-GPU box's way to run RCCL calls through the capture path)
        self.dp_active = world_size > 1 or bool(force_collectives)
        # backward_G in three segments cut where a network's gradients are complete (flowNetB | netG | flowNetF): the all-reduce of a
        # finished network overlaps the next segment even when every segment is a replayed hipGraph (capture mode "segments")
        self.segmented = bool(segmented_backward)
        torch.manual_seed(seed)
        self.warp = warp if warp is not None else WarpNet()
        if warp is None and self.device.type == "cuda":
            # independent warps issued together go out as ONE multi-problem launch (csrc/warp.hip)
            from .external_function import warp_many as _hip_warp_many
            self.warp_many = lambda feats, flows: _hip_warp_many(feats, flows, False)
        else:
            self.warp_many = lambda feats, flows: [self.warp(f, fl) for f, fl in zip(feats, flows)]
        self.flowNetF = nets.FlowNet(ngf).to(self.device)
        if self.device.type == "cuda":
            from . import miopen_tuning
            miopen_tuning.install()      # solver selection for the conv stacks; before the first convolution
        self.flowNetB = nets.FlowNet(ngf).to(self.device)
        self.netG = nets.FFWM(sn=True, warp_flipcat=warp_flipcat).to(self.device)
        self.netD = nets.MSDiscriminator(128, sigmoid=False).to(self.device)
        self.lightCNN = nets.LightCNN29().to(self.device).eval()
        self.vgg = nets.VGG19("relu5_1").to(self.device).eval()
        for p in self.lightCNN.parameters():
            p.requires_grad = False
        if self.device.type == "cuda":      # one launch forward + one backward instead of ~300 (csrc/guided_filter.hip)
            from .external_function import GuidedFilter as HipGuidedFilter
            self.gf = {128: HipGuidedFilter(32), 64: HipGuidedFilter(16), 32: HipGuidedFilter(8)}
        else:
            self.gf = {128: nets.GuidedFilter(32), 64: nets.GuidedFilter(16), 32: nets.GuidedFilter(8)}

        # spectral norm of netG's 52 and netD's 9 convs: one batched launch per forward call instead
        # of ~12 tiny kernels per layer (GPU only -- the CPU baseline keeps PyTorch's per-layer hooks)
        if mfma_wgrad is None:
            mfma_wgrad = self.device.type == "cuda"
        if mfma_wgrad:
            # weight gradients of the large-image 3x3 layers on the hand-written MFMA kernel (conv.py)
            from .conv import route_conv_wgrad
            self.mfma_wgrad_layers = route_conv_wgrad(self.netG)
        self.winograd_layers = 0
        if self.device.type == "cuda":
            # forward + data gradient of the large-plane 3x3 / stride-1 layers: fp32 Winograd on the MFMA units (conv.py),
            # before route_conv_fwd so that the small-plane policy of that route only sees what is left
            from .conv import route_conv_winograd
            self.winograd_layers = sum(route_conv_winograd(net) for net in (self.flowNetF, self.flowNetB, self.netG, self.netD, self.lightCNN))
        if mfma_fwd is None:
            mfma_fwd = self.device.type == "cuda"
        self.mfma_fwd_layers = 0
        if mfma_fwd:
            # forward (and the 4x4 / stride-2 data gradients) of the stride-2 / transposed / small-plane convolutions on the
            # hand-written MFMA kernel instead of the vendor's NHWC implicit GEMM + layout transposes (conv.py)
            from .conv import route_conv_fwd
            self.mfma_fwd_layers = sum(route_conv_fwd(net) for net in (self.flowNetF, self.flowNetB, self.netG, self.netD))
        self.own_bwd_layers = 0
        if mfma_fwd and self.device.type == "cuda":
            # FlowNet's two-channel layers (7 flow heads + 6 flow upsamplers per net): direct kernels forward and backward
            from .conv import route_flow_heads
            self.flow_head_layers = sum(route_flow_heads(net) for net in (self.flowNetF, self.flowNetB))
            # what is left (thin layers, flow heads / upsamplers, image layers): vendor forward, weight gradient on the tiled kernel
            from .conv import route_conv_bwd
            self.own_bwd_layers = sum(route_conv_bwd(net) for net in (self.flowNetF, self.flowNetB, self.netG, self.netD))
        if fused_bn is None:
            fused_bn = self.device.type == "cuda"
        if fused_bn:
            # BatchNorm2d + LeakyReLU pairs of the conv blocks as one kernel per direction (norm.py, csrc/bn_lrelu.hip)
            from .norm import fuse_bn_lrelu
            self.fused_bn_layers = sum(fuse_bn_lrelu(net) for net in (self.flowNetF, self.flowNetB, self.netG, self.netD))
        self.fused_residual_blocks = 0
        if fused_bn and self.device.type == "cuda" and os.environ.get("FFWM_FUSED_RESIDUAL", "1") == "1":
            # add + activation behind every residual block and the warp-attention gate `skip * att_i(skip)` as one kernel per
            # direction (residual.py, csrc/residual.hip)
            from .residual import fuse_residual
            self.fused_residual_blocks = sum(fuse_residual(net) for net in (self.netG, self.netD))
        if fused_spectral_norm is None:
            fused_spectral_norm = self.device.type == "cuda"
        if fused_spectral_norm:
            from .spectral_norm import fuse_spectral_norm
            fuse_spectral_norm(self.netG)
            fuse_spectral_norm(self.netD)

        broadcast_module_state([self.flowNetF, self.flowNetB, self.netG, self.netD, self.lightCNN, self.vgg])

        flow_params = [p for net in (self.flowNetF, self.flowNetB) for n, p in net.named_parameters()
                       if not n.startswith("inter_conv_occ")]
        # Adam hyper-parameters of ffwm_model.py:46-49 (capturable=True keeps the step counter on the
        # device so that an optimizer step can sit inside a captured hipGraph; it costs ~10 ms per
        # eager step, so it is only switched on for capture())
        cap = bool(capturable) and self.device.type == "cuda"
        # fused (one multi-tensor kernel per optimizer instead of ~10 foreach launches) on the GPU; its
        # capturable form keeps the step counter on the device without the per-parameter kernels of the
        # foreach implementation (1275 extra launches per step, measured)
        kw = {"fused": True, "capturable": cap} if self.device.type == "cuda" else {}
        self.world_size = world_size
        # Independent branches of the step on side streams (default: on for a trainer built for capture, FFWM_STREAMS=0 / 1 overrides;
        # in eager mode the host issues the launches one by one and nothing overlaps).  Under hipGraph replay the step is bound by
        # kernel time alone, and most kernels of these branches (2 x 2 ... 64 x 64 planes) fill a fraction of the chip:
        #   * flowNetB beside flowNetF (forward and, because autograd replays a node on its forward's stream, backward);
        #   * the loss networks' passes: VGG19 on the 64 / 32 px scales and LightCNN beside the 128 px VGG19 pass;
        #   * the whole D step (netD forward / backward on 8 ... 64 px planes, its gradient packing and Adam): the generator's loss
        #     networks need nothing from it until the adversarial term.  One GPU only: with several ranks the D gradients'
        #     all-reduce sits between the captured segments.
        # TWO side streams carry all of it (flowNetB, later VGG19 64 / 32 px on the first; the D step, then LightCNN on the second):
        # the HIP runtime maps streams and a graph's branches onto 4 hardware queues, and with five streams (one per branch, the
        # first version) the same binary ran at 45.2-45.9 ms or at 42.4-42.8 ms per step depending on the process -- queues
        # beyond the fourth are time-sliced.  Three streams: 42.5 ms every time (profiles/r03_hw_queues.txt).
        # FFWM_STREAM_LAYOUT=5 restores one stream per branch for an A/B.
        multi = os.environ.get("FFWM_STREAMS", os.environ.get("FFWM_FLOW_STREAMS", "1" if capturable else "0")) == "1"
        multi = multi and self.device.type == "cuda"
        # Round 4: on one GPU the D step gets a THIRD side stream of its own (layout 4: four streams = the four hardware queues; 8 of 8
        # processes 38.9-39.1 ms against 39.3-39.5 with two side streams, profiles/r04_stream_experiments.txt); with several ranks RCCL's
        # stream is the fourth, so the D step keeps sharing LightCNN's (layout 3).
        layout = os.environ.get("FFWM_STREAM_LAYOUT", "3" if self.dp_active else "4")
        five = layout == "5"
        self.loss_streams = [torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)] if multi else None
        self.flow_stream = (torch.cuda.Stream(self.device) if five else self.loss_streams[0]) if multi else None
        self.d_stream = None
        if multi and os.environ.get("FFWM_D_STREAM", "1") == "1":
            # (FFWM_STREAM_LAYOUT=4: the D step on a third side stream of its own -- four streams = the four hardware queues)
            self.d_stream = torch.cuda.Stream(self.device) if (five or layout == "4") else self.loss_streams[1]
        # flowNetF on the SECOND side stream (idle until the D step forks, and idle again once LightCNN's backward is through), beside
        # netG's encoder: e0-e3 need no flow, so the step's stream runs them while both flow nets -- ~3 ms of 2 x 2 ... 64 x 64 plane
        # kernels each -- run beside it, and in the backward flowNetF's pass (the tail of the step: it waits for d(flow) from the
        # warps) runs beside the encoder's.  No new stream: the four hardware queues stay enough (see above).  One GPU only for now:
        # with several ranks the reducers' hooks would launch flowNetF's buckets from that stream.
        # (measured, round 4: 41.3 ms against 41.0-41.1 with flowNetF on the step's stream -- netG's encoder is Winograd kernels that
        # own every CU's LDS and registers while they run, a side stream's kernels wait for them instead of running beside them; what
        # overlaps on this chip is small kernels with OTHER small kernels.  Opt-in: FFWM_FLOWF_STREAM=1.)
        self._event_joins = multi and not five and os.environ.get("FFWM_EVENT_JOINS", "1") == "1"
        self._prefetch_targets = os.environ.get("FFWM_TARGET_PREFETCH", "1") == "1"
        # Opt-in experiment (FFWM_PAIR_FLOW_BWD=1): the flow nets' BACKWARD passes side by side.  flowNetB's gradient is complete long
        # before flowNetF's (it comes straight from the illumination warp), so its backward runs on its side stream in the middle of
        # netG's backward and flowNetF's alone at the tail of the step; a gate (_FlowGate) can hold flowNetB's backward until flowNetF's
        # starts.  Measured neutral (40.47 / 40.63 ms with the gate, 40.48 without, profiles/r04_stream_experiments.txt): flowNetB's
        # backward already finds room between netG's kernels.
        self._pair_flow_backward = os.environ.get("FFWM_PAIR_FLOW_BWD", "0") == "1"
        self._tgt = {}
        self.flowf_stream = None
        if multi and not five and not self.dp_active and os.environ.get("FFWM_FLOWF_STREAM", "0") == "1":
            self.flowf_stream = self.loss_streams[1]
        # the D step on its side stream: always on one GPU; with several ranks only when the collectives are captured into the
        # step's graph (capture mode "ingraph": the D gradients' all-reduce is then a node of the side branch)
        self._d_side = not self.dp_active
        self._d_pending = False
        self.batched_losses = batched_losses
        # the ~25 L1 terms of backward_G as one launch per direction (losses.l1_terms, csrc/l1_loss.hip); needs the batched passes
        self.fused_l1 = (self.device.type == "cuda" and batched_losses) if fused_l1 is None else bool(fused_l1)
        self._graphs = None
        self._static = None
        # eager steps pack the gradients into the flat arrays after backward (no per-parameter accumulation kernel);
        # a captured graph needs static gradient addresses: in-place accumulation into the views
        # (round 3: on ONE GPU the captured step packs too -- inside a capture the fresh gradients come from the graph's private pool,
        # their addresses are as static as the views'; with several ranks the packing runs between the graphs, in Python, where the
        # replayed gradients are not visible as new tensors: those keep the in-place accumulation)
        gather = True          # (several ranks under capture: the buckets are packed at the end of each captured backward segment)
        # one bucket group per network: a network's backward runs on one stream, and a segmented backward completes them one by one
        self._params_F = [p for n, p in self.flowNetF.named_parameters() if not n.startswith("inter_conv_occ")]
        self._params_B = [p for n, p in self.flowNetB.named_parameters() if not n.startswith("inter_conv_occ")]
        self.red_G = BucketedGradReducer(None, groups=[self._params_F, self._params_B, list(self.netG.parameters())],
                                         bucket_bytes=bucket_bytes, gather=gather, force_collectives=force_collectives)
        self.red_D = BucketedGradReducer(self.netD.parameters(), bucket_bytes=bucket_bytes, gather=gather,
                                         force_collectives=force_collectives)
        if flat_adam is None:
            flat_adam = self.device.type == "cuda"
        self.flat_adam = bool(flat_adam)
        if self.flat_adam:
            # parameters, gradients and moments as flat arrays: one streaming kernel per optimizer step (csrc/adam.hip); capturable:
            # the step counter lives on the device
            from .optim import FlatAdam
            self.opt_F = FlatAdam(flow_params, self.red_G, lr=0.00005, betas=(0.5, 0.999), capturable=cap)
            self.opt_G = FlatAdam(list(self.netG.parameters()), self.red_G, lr=0.0004, betas=(0.5, 0.999), capturable=cap)
            self.opt_D = FlatAdam(list(self.netD.parameters()), self.red_D, lr=0.0004, betas=(0.5, 0.999), capturable=cap)
        else:
            self.opt_F = torch.optim.Adam(flow_params, lr=0.00005, betas=(0.5, 0.999), **kw)
            self.opt_G = torch.optim.Adam(self.netG.parameters(), lr=0.0004, betas=(0.5, 0.999), **kw)
            self.opt_D = torch.optim.Adam(self.netD.parameters(), lr=0.0004, betas=(0.5, 0.999), **kw)
        self.losses = {}

    # ------------------------------------------------------------------ stand-in for the pretrained flow nets
    def pretrain_flow_identity(self, b, steps=80, lr=2e-3):
        """The reference never trains FFWM from randomly initialised flow nets: train_ffwm.py loads
        pretrained flowNetF / flowNetB checkpoints (README, models/ffwm_model.py:30-35), whose fields
        are smooth near-identity sampling grids.  Those checkpoints cannot be fetched offline, and an
        untrained FlowNet outputs tanh(~0): every pixel samples the image centre -- an access pattern
        (all lanes on one cache line, all scatter-adds on four cells) that real training never sees.
        This fits both nets to the identity grid for a few Adam steps (own throw-away optimizer;
        the step's optimizers and their state are untouched) so the timed step runs on realistic flows."""
        self._no_eager_while_captured("pretrain_flow_identity")
        def grid(s):
            lin = (torch.arange(s, dtype=torch.float32, device=self.device) + 0.5) / s * 2 - 1
            yy, xx = torch.meshgrid(lin, lin, indexing="ij")
            return torch.stack((xx, yy), 0).unsqueeze(0)
        targets = [grid(128), grid(64), grid(32)]
        last = []
        # the throw-away optimisation must not drive the step's gradient reducer: its post-accumulate hooks would count
        # these backward passes and launch all-reduces nobody waits for
        overlap_before = self.red_G.overlap
        self.red_G.set_overlap(False)
        for net in (self.flowNetF, self.flowNetB):
            params = [p for n, p in net.named_parameters() if not n.startswith("inter_conv_occ")]
            grads_before = [p.grad for p in params]          # views into the reducer's buckets: keep them
            opt = torch.optim.Adam(params, lr=lr)
            for _ in range(steps):
                flows = net(b["img_S"])
                loss = sum((f - g).abs().mean() for f, g in zip(flows, targets))
                for p in params:
                    p.grad = None
                loss.backward()
                opt.step()
            for p, g in zip(params, grads_before):
                p.grad = g
            last.append(float(loss.detach()))
        self.red_G.set_overlap(overlap_before)
        self.red_G.zero_grad()                           # pending / launched counters back to a clean step
        broadcast_module_state([self.flowNetF, self.flowNetB])
        return last

    # ------------------------------------------------------------------ loss pieces
    def perceptual(self, x, y):
        fx = self.vgg(x)
        with torch.no_grad():
            fy = self.vgg(y)
        return sum(w * F.l1_loss(fx[k], fy[k]) for k, w in zip(PRC_LAYERS, PRC_WEIGHTS))

    def perceptual_many(self, pairs):
        """[perceptual(x_i, y_i) for i] for pairs that share one shape, with ONE VGG pass over the
        concatenated x's and one over the y's.  VGG19 has no batch statistics, so every sample's
        features are those of a separate call; only the number of launches changes."""
        n = len(pairs)
        fx = self.vgg(torch.cat([p[0] for p in pairs], 0))
        with torch.no_grad():
            fy = self.vgg(torch.cat([p[1] for p in pairs], 0))
        total = 0
        for k, w in zip(PRC_LAYERS, PRC_WEIGHTS):
            total = total + w * (fx[k] - fy[k]).abs().reshape(n, -1).mean(1)      # the L1 mean of each pair
        return total

    def identity(self, out, gt):
        _, fc_o, pool_o = self.lightCNN(out.mean(1, keepdim=True))
        with torch.no_grad():
            _, fc_g, pool_g = self.lightCNN(gt.mean(1, keepdim=True))
        return F.l1_loss(fc_o, fc_g) + F.l1_loss(pool_o, pool_g)

    def identity_many(self, outs, gt, weights):
        """sum_i weight_i * identity(out_i, gt): the ground-truth features are extracted once, and an
        output that appears several times (warm-up branch: the guided-filter output IS fake128) is run
        through LightCNN once.  LightCNN has no batch statistics either."""
        uniq, wsum = [], []
        for o, w in zip(outs, weights):
            for i, u in enumerate(uniq):
                if u is o:
                    wsum[i] += w
                    break
            else:
                uniq.append(o)
                wsum.append(w)
        with torch.no_grad():
            _, fc_g, pool_g = self.lightCNN(gt.mean(1, keepdim=True))
        n, b = len(uniq), gt.size(0)
        _, fc_o, pool_o = self.lightCNN(torch.cat([u.mean(1, keepdim=True) for u in uniq], 0))
        total = 0
        for i, w in enumerate(wsum):
            sl = slice(i * b, (i + 1) * b)
            total = total + w * (F.l1_loss(fc_o[sl], fc_g) + F.l1_loss(pool_o[sl], pool_g))
        return total

    def illumination(self, flows_B, fakes, img_S, mask_S):
        """MSL1Loss (losses.py:130-157): warp each generated scale back with flowNetB and compare
        with the (resized) profile input."""
        total = 0
        warped = self.warp_many(list(fakes), list(flows_B))            # the three scales: one launch
        for w, flow, back in zip((1, 1, 1.5), flows_B, warped):
            size = flow.shape[2:]
            tgt = F.interpolate(img_S, size, mode="bilinear", align_corners=True)
            m = F.interpolate(mask_S, size, mode="nearest")
            total = total + w * F.l1_loss(back * m, tgt * m)
        return total

    @staticmethod
    def lsgan(pred, real):
        return F.mse_loss(pred, torch.ones_like(pred) if real else torch.zeros_like(pred))

    # ------------------------------------------------------------------ one optimisation step
    def forward(self, b):
        img_S, img_F = b["img_S"], b["img_F"]
        self._tgt = {}
        if (self.flow_stream is not None and self._event_joins and not self.segmented
                and (not self.dp_active or getattr(self, "capture_mode", None) == "ingraph")):
            # ONE graph (one GPU, or several ranks with the collectives captured inside it -- the three-graph split must find every side
            # stream joined where a graph ends): the side branches are joined by EVENTS, not whole-stream waits, so that more work can
            # follow on a side stream behind the point the step's stream waits for (the ground-truth passes below), and the two flow
            # nets' BACKWARD passes can be paired at the tail of the step (the gates, opt-in).
            cur = torch.cuda.current_stream(self.device)
            box = {}
            self.flow_stream.wait_stream(cur)                  # fork: img_S and last step's weights are ready
            with torch.cuda.stream(self.flow_stream):
                flows_B = self.flowNetB(img_S)
                if self._pair_flow_backward:
                    flows_B = _FlowGate.apply(box, False, *flows_B)
                self.flows_B = list(flows_B)
            ev_B = torch.cuda.Event()
            ev_B.record(self.flow_stream)
            fstream = self.flowf_stream                        # (opt-in: flowNetF on the second side stream beside netG's encoder)
            ev_F = None
            if fstream is not None:
                fstream.wait_stream(cur)
            with (torch.cuda.stream(fstream) if fstream is not None else contextlib.nullcontext()):
                flows_F = self.flowNetF(img_S)
                if self._pair_flow_backward:
                    flows_F = _FlowGate.apply(box, True, *flows_F)
            if fstream is not None:
                ev_F = torch.cuda.Event()
                ev_F.record(fstream)
            if self._prefetch_targets and self.fused_l1:
                # the loss networks' passes over the GROUND TRUTH (no_grad, inputs only) behind flowNetB / on the second side stream,
                # beside netG's forward, instead of on the step's stream in the middle of the loss section
                mask_F = b["mask_F"]
                with torch.cuda.stream(self.flow_stream), torch.no_grad():
                    self._tgt["vgg128"] = self.vgg(img_F * mask_F)
                    self._tgt["vgg64"] = self.vgg(F.interpolate(img_F, (64, 64), mode="bilinear") * F.interpolate(mask_F, (64, 64), mode="nearest"))
                self.loss_streams[1].wait_stream(cur)
                with torch.cuda.stream(self.loss_streams[1]), torch.no_grad():
                    self._tgt["light"] = self.lightCNN(img_F.mean(1, keepdim=True))

            def flows_ready():
                # called by netG once its encoder is issued: join the flow nets, warp the images, hand the flows over
                if ev_F is not None:
                    cur.wait_event(ev_F)
                cur.wait_event(ev_B)
                for f in list(self.flows_B) + (list(flows_F) if ev_F is not None else []):
                    f.record_stream(cur)                       # allocated on a side stream, consumed on this one
                self.flows_F = list(flows_F)
                self.img_S_warp, self.img_S_rec = self.warp_many([img_S, img_F], [flows_F[0], self.flows_B[0]])
                return [flows_F[2], flows_F[1], flows_F[0]]
            self.fake32, self.fake64, self.fake128 = self.netG(img_S, flow=flows_ready)
            self.img_GF128 = self.gf[128](self.fake128, img_F)
            grids = part_grids(b["lm_F"])            # eye-l, eye-r, nose, mouth
            crops = self.warp_many([self.img_GF128, img_F] * len(grids), [g for g in grids for _ in (0, 1)])   # 8 crops: one launch
            self.parts = [(crops[2 * i], crops[2 * i + 1]) for i in range(len(grids))]
            return
        if self.flow_stream is not None:
            cur = torch.cuda.current_stream(self.device)
            self.flow_stream.wait_stream(cur)                  # fork: img_S and last step's weights are ready
            with torch.cuda.stream(self.flow_stream):
                self.flows_B = self.flowNetB(img_S)
            flow_F128, flow_F64, flow_F32 = self.flowNetF(img_S)
            cur.wait_stream(self.flow_stream)                  # join: everything downstream runs on the step's stream
            for f in self.flows_B:
                f.record_stream(cur)                           # allocated on the side stream, consumed on this one
        else:
            flow_F128, flow_F64, flow_F32 = self.flowNetF(img_S)
            self.flows_B = self.flowNetB(img_S)
        if self.segmented:
            # a segment boundary must be a CUT of the autograd graph: FlowNet feeds its coarse flows into the finer ones and netG its
            # coarse reconstructions into the next level, so the tensors themselves are not one (a coarse output is an ancestor of a
            # fine one).  Aliases (views: no kernel) consumed only on the far side of the boundary are.
            flow_F128, flow_F64, flow_F32 = (f.view_as(f) for f in (flow_F128, flow_F64, flow_F32))
        self.flows_F = [flow_F128, flow_F64, flow_F32]
        self.img_S_warp, self.img_S_rec = self.warp_many([img_S, img_F], [flow_F128, self.flows_B[0]])
        self.fake32, self.fake64, self.fake128 = self.netG(img_S, flow=[flow_F32, flow_F64, flow_F128])
        if self.segmented:
            self.fake32, self.fake64, self.fake128 = (f.view_as(f) for f in (self.fake32, self.fake64, self.fake128))
        self.img_GF128 = self.gf[128](self.fake128, img_F)
        grids = part_grids(b["lm_F"])            # eye-l, eye-r, nose, mouth
        crops = self.warp_many([self.img_GF128, img_F] * len(grids), [g for g in grids for _ in (0, 1)])   # 8 crops: one launch
        self.parts = [(crops[2 * i], crops[2 * i + 1]) for i in range(len(grids))]

    def backward_D(self, b):
        m = b["mask_F"]
        fake = self.netD(self.img_GF128.detach() * m)
        real = self.netD(b["img_F"] * m)
        loss_D = (self.lsgan(fake, False) + self.lsgan(real, True)) * 0.5
        loss_D.backward()
        self.loss_D = loss_D.detach()

    def _backward_G_fused_l1(self, b):
        """backward_G with every `w * F.l1_loss(x * m, y * m)` term (pixel, perceptual, part crops, illumination, identity:
        ffwm_model.py:107-139) handed to ONE fused launch per direction; the networks see exactly the inputs they see in
        backward_G below.  Slots of the result vector: l1, prc, fc, illu, iden."""
        from .losses import l1_terms
        img_F, mask_F = b["img_F"], b["mask_F"]
        img_F64 = F.interpolate(img_F, (64, 64), mode="bilinear")
        img_F32 = F.interpolate(img_F, (32, 32), mode="bilinear")
        mask64 = F.interpolate(mask_F, (64, 64), mode="nearest")
        mask32 = F.interpolate(mask_F, (32, 32), mode="nearest")
        if self.titers < 20000:
            gf128, gf64, gf32 = self.fake128, self.fake64, self.fake32
        else:
            gf128 = self.img_GF128
            gf64 = self.gf[64](self.fake64, img_F64)
            gf32 = self.gf[32](self.fake32, img_F32)
        L1, PRC, FC, ILLU, IDEN = range(5)
        B = img_F.size(0)
        terms = [(gf128, img_F, mask_F, 5.0, L1), (gf64, img_F64, mask64, 5.0, L1), (gf32, img_F32, mask32, 7.5, L1)]
        # The loss networks' passes are independent of each other until their gradients meet at the generated images: with
        # `loss_streams` the VGG19 passes of the 64 / 32 px scales and the LightCNN passes run on two side streams beside the 128 px
        # VGG19 pass (their kernels -- 16 x 16 ... 64 x 64 planes -- fill a fraction of the chip); autograd runs each backward node
        # on its forward's stream, so the backward overlaps the same way.  Fork here, join in front of the fused L1 launch.
        cur = torch.cuda.current_stream(self.device) if self.loss_streams else None
        side_terms = []

        def branch(i):
            if self.loss_streams is None:
                return contextlib.nullcontext()
            self.loss_streams[i].wait_stream(cur)
            return torch.cuda.stream(self.loss_streams[i])

        pre = getattr(self, "_tgt", None) or {}            # ground-truth features computed at the start of the step (forward())

        def vgg_pair(x, y, m, key):
            fx = self.vgg(x * m)
            if key in pre:
                fy = pre[key]
            else:
                with torch.no_grad():
                    fy = self.vgg(y * m)
            return [(fx[k], fy[k], None, w, PRC) for k, w in zip(PRC_LAYERS, PRC_WEIGHTS)]
        # perceptual: the two large scales one VGG pass each, the 32 x 32 scale and the four part crops share one (5 B rows)
        with branch(0):
            side_terms += vgg_pair(gf64, img_F64, mask64, "vgg64")
            (el, elt), (er, ert), (no, nog), (mo, mog) = self.parts
            fx = self.vgg(torch.cat((gf32 * mask32, el, er, mo, no), 0))
            with torch.no_grad():
                fy = self.vgg(torch.cat((img_F32 * mask32, elt, ert, mog, nog), 0))
            for k, w in zip(PRC_LAYERS, PRC_WEIGHTS):
                side_terms.append((fx[k], fy[k], None, [(0, 0, B, 1.5 * w, PRC), (B, B, B, 2 * w, FC), (2 * B, 2 * B, B, 2 * w, FC),
                                                        (3 * B, 3 * B, B, w, FC), (4 * B, 4 * B, B, w, FC)]))
        # identity: ground-truth features once; an output that appears twice (warm-up branch: gf128 IS fake128) runs once
        uniq, wsum = [], []
        for o, w in zip((self.fake128, gf128), (0.5, 1.0)):
            for i, u in enumerate(uniq):
                if u is o:
                    wsum[i] += w
                    break
            else:
                uniq.append(o)
                wsum.append(w)
        with branch(1):
            if "light" in pre:
                _, fc_g, pool_g = pre["light"]
            else:
                with torch.no_grad():
                    _, fc_g, pool_g = self.lightCNN(img_F.mean(1, keepdim=True))
            _, fc_o, pool_o = self.lightCNN(torch.cat([u.mean(1, keepdim=True) for u in uniq], 0))
            side_terms.append((fc_o, fc_g, None, [(i * B, 0, B, w, IDEN) for i, w in enumerate(wsum)]))
            side_terms.append((pool_o, pool_g, None, [(i * B, 0, B, w, IDEN) for i, w in enumerate(wsum)]))
        t128 = vgg_pair(gf128, img_F, mask_F, "vgg128")
        terms += t128
        # illumination (MSL1Loss): the three generated scales warped back with flowNetB (one multi-problem launch)
        warped = self.warp_many([self.fake128, self.fake64, self.fake32], list(self.flows_B))
        for w, flow, back in zip((1, 1, 1.5), self.flows_B, warped):
            size = flow.shape[2:]
            tgt = F.interpolate(b["img_S"], size, mode="bilinear", align_corners=True)
            m = F.interpolate(b["mask_S"], size, mode="nearest")
            terms.append((back, tgt, m, 15.0 * w, ILLU))
        if self.loss_streams is not None:
            for st in self.loss_streams:
                cur.wait_stream(st)
            for t in side_terms:                           # allocated on a side stream, read by the fused L1 launch on this one
                t[0].record_stream(cur)
                t[1].record_stream(cur)
            if "vgg128" in pre:
                for t in t128:
                    t[1].record_stream(cur)
        terms += side_terms
        v = l1_terms(terms, 5)
        self._join_D()
        loss_adv = self.lsgan(self.netD(self.img_GF128 * mask_F), True) * 0.1
        self.loss_G = v.sum() + loss_adv
        self.losses = {"G": self.loss_G, "l1": v[L1], "iden": v[IDEN], "illu": v[ILLU], "adv": loss_adv, "prc": v[PRC], "fc": v[FC]}
        self._backward_from_loss_G()

    def backward_G(self, b):
        if self.fused_l1 and self.parts[0][0].shape[2:] == (32, 32):
            return self._backward_G_fused_l1(b)
        img_F, mask_F = b["img_F"], b["mask_F"]
        img_F64 = F.interpolate(img_F, (64, 64), mode="bilinear")
        img_F32 = F.interpolate(img_F, (32, 32), mode="bilinear")
        mask64 = F.interpolate(mask_F, (64, 64), mode="nearest")
        mask32 = F.interpolate(mask_F, (32, 32), mode="nearest")
        if self.titers < 20000:        # warm-up branch (ffwm_model.py:97-101)
            gf128, gf64, gf32 = self.fake128, self.fake64, self.fake32
        else:
            gf128 = self.img_GF128
            gf64 = self.gf[64](self.fake64, img_F64)
            gf32 = self.gf[32](self.fake32, img_F32)
        pairs = ((gf128, img_F, mask_F, 1.0), (gf64, img_F64, mask64, 1.0), (gf32, img_F32, mask32, 1.5))
        (el, elt), (er, ert), (no, nog), (mo, mog) = self.parts
        if self.batched_losses and el.shape == gf32.shape:
            # result-preserving launch diet: the 32 x 32 scale and the four 32 x 32 part crops share
            # one VGG pass (five pairs per call instead of five calls)
            p32 = self.perceptual_many([(gf32 * mask32, img_F32 * mask32), (el, elt), (er, ert), (mo, mog), (no, nog)])
            loss_prc = self.perceptual(gf128 * mask_F, img_F * mask_F) + self.perceptual(gf64 * mask64, img_F64 * mask64) + \
                1.5 * p32[0]
            loss_fc = 2 * (p32[1] + p32[2]) + p32[3] + p32[4]
        else:
            loss_prc = sum(w * self.perceptual(x * m, y * m) for x, y, m, w in pairs)
            loss_fc = 2 * (self.perceptual(el, elt) + self.perceptual(er, ert)) + self.perceptual(mo, mog) + \
                self.perceptual(no, nog)
        loss_l1 = sum(w * F.l1_loss(x * m, y * m) for x, y, m, w in pairs) * 5
        loss_illu = self.illumination(self.flows_B, (self.fake128, self.fake64, self.fake32), b["img_S"],
                                      b["mask_S"]) * 15
        if self.batched_losses:
            loss_iden = self.identity_many((self.fake128, gf128), img_F, (0.5, 1.0))
        else:
            loss_iden = self.identity(self.fake128, img_F) * 0.5 + self.identity(gf128, img_F) * 1
        self._join_D()
        loss_adv = self.lsgan(self.netD(self.img_GF128 * mask_F), True) * 0.1
        self.loss_G = loss_iden + loss_l1 + loss_prc + loss_illu + loss_fc + loss_adv
        self.losses = {"G": self.loss_G, "l1": loss_l1, "iden": loss_iden, "illu": loss_illu, "adv": loss_adv,
                       "prc": loss_prc, "fc": loss_fc}
        self._backward_from_loss_G()

    # ------------------------------------------------------------------ backward of the generator loss, whole or in segments
    G_F, G_B, G_NET = 0, 1, 2          # bucket groups of red_G (dp.BucketedGradReducer groups=)

    def _backward_from_loss_G(self):
        """loss_G.backward() -- or, segmented, its first part: the loss networks back to the generated images and, beside it (its own
        stream), flowNetB: d(loss)/d(flows_B) comes straight from the illumination warp, so flowNetB's gradients are the first complete."""
        if not self.segmented:
            self.loss_G.backward()
            return
        fakes = [self.fake128, self.fake64, self.fake32]
        torch.autograd.backward([self.loss_G], inputs=fakes + self._params_B, retain_graph=True)

    def _bwd_netG(self):
        """second segment: netG, from the gradients of the three generated scales down to its weights and the flow fields."""
        fakes = [self.fake128, self.fake64, self.fake32]
        grads = [f.grad for f in fakes]
        for f in fakes:
            f.grad = None
        torch.autograd.backward(fakes, grads, inputs=[p for p in self.netG.parameters() if p.requires_grad] + self.flows_F,
                                retain_graph=True)

    def _bwd_flowF(self):
        """third segment: flowNetF from the gradients of its three flow fields."""
        grads = [f.grad for f in self.flows_F]
        for f in self.flows_F:
            f.grad = None
        torch.autograd.backward(self.flows_F, grads, inputs=self._params_F)

    def _finish_backward_G(self):
        """the segments after the first, eagerly: every finished network's buckets go out (asynchronous) before the next segment is
        issued.  (An unsegmented step has nothing left to do here.)"""
        if not self.segmented:
            return
        self.red_G.launch_group(self.G_B)
        self._bwd_netG()
        self.red_G.launch_group(self.G_NET)
        self._bwd_flowF()
        self.red_G.launch_group(self.G_F)

    # optimize_parameters (ffwm_model.py:151-160) in three segments, cut where data parallelism has its
    # exchange steps (gradient all-reduce of the D set, then of the G set)
    def _seg_forward_and_D(self, b):
        if self.device.type == "cuda" and _GRAD_ARENA_ON:
            conv.GRAD_ARENA.begin(self.device)          # ONE launch clears every weight-gradient buffer of the step (conv._GradArena)
        self.forward(b)
        for p in self.netD.parameters():
            p.requires_grad = True
        if self.d_stream is not None and self._d_side:
            # fork: the D step (forward, backward, gradient packing, Adam) beside the generator's loss passes; _join_D() in
            # front of the adversarial term (the first reader of the updated netD) is the join
            self.d_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.d_stream):
                self.red_D.zero_grad()
                self.backward_D(b)
                if self._d_side == "reduce_on_main":
                    # several ranks, collectives captured into the graph: only netD's forward / backward run on the side branch; the
                    # all-reduce of its gradients and its Adam are issued on the step's stream at the join (RCCL's stream forked from a
                    # side branch AND joined into it crashed hipStreamEndCapture on ROCm 7.0 -- profiles/r04_dp_capture_modes.txt)
                    self.red_D.pack_all()
                else:
                    self.red_D.finish()
                    self.opt_D.step()
            self._d_pending = True
            return
        self.red_D.zero_grad()
        self.backward_D(b)

    def _reduce_D(self):
        if not self._d_pending:
            self.red_D.finish()

    def _drop_autograd_graph(self):
        """Detach what the step keeps for logging / visuals.  A live graph keeps every parameter's AccumulateGrad node alive, and a
        node remembers the stream it was created on: the next step (or the capture after the warm-up steps, which runs on another
        stream) would accumulate -- and pack the gradient buckets -- on that stale stream, serialising the side streams' backward."""
        self.losses = {k: v.detach() for k, v in self.losses.items()}
        self.loss_G = self.loss_G.detach()
        for name in ("fake32", "fake64", "fake128", "img_GF128", "img_S_warp", "img_S_rec"):
            setattr(self, name, getattr(self, name).detach())
        self.flows_B = [f.detach() for f in self.flows_B]
        self.flows_F = [f.detach() for f in self.flows_F]
        self.parts = [(a.detach(), b.detach()) for a, b in self.parts]
        for net in (self.netG, self.netD):
            for m in net.modules():                  # spectral norm leaves the normalised weight (a graph output) on the module
                w = m.__dict__.get("weight")
                if torch.is_tensor(w) and w.grad_fn is not None:
                    m.weight = w.detach()

    def _join_D(self):
        if self._d_pending:
            torch.cuda.current_stream(self.device).wait_stream(self.d_stream)
            self._d_pending = False
            if self._d_side == "reduce_on_main":
                self.red_D.finish()
                self.opt_D.step()

    def _seg_stepD_and_G(self, b):
        if not self._d_pending:
            self.opt_D.step()
        for p in self.netD.parameters():
            p.requires_grad = False
        self.red_G.zero_grad()
        self.backward_G(b)
        if not self.segmented:
            self._drop_autograd_graph()

    def _seg_stepG(self):
        self.opt_G.step()
        self.opt_F.step()
        conv.GRAD_ARENA.end()

    def step(self, b, batch_increment=None):
        """optimize_parameters (ffwm_model.py:151-160): forward, D step, G step."""
        if self._graphs is not None:
            return self._step_graphed(b, batch_increment)
        try:
            self._seg_forward_and_D(b)
            self._reduce_D()
            self._seg_stepD_and_G(b)
            if self.segmented:
                self._finish_backward_G()
                self._drop_autograd_graph()
            self.red_G.finish()
            self._seg_stepG()
        finally:
            conv.GRAD_ARENA.end()             # also when the step raised: a later pass outside a step must not carve the arena
        self.titers += batch_increment if batch_increment is not None else b["img_S"].size(0)
        self.losses["D"] = self.loss_D
        return self.losses

    # ------------------------------------------------------------------ hipGraph replay
    def capture(self, b, warmup=3, mode=None):
        """Capture the train step into hipGraphs (HIP graphs through torch.cuda.CUDAGraph): the eager
        step issues ~2700 small launches and is launch-bound (SURVEY 7 'hard parts'); a replay submits
        them as pre-built graphs.  Single GPU: ONE graph for the whole step.  Data parallel, `mode`
        (default: FFWM_DP_CAPTURE, else "serial"; "ingraph" is opt-in until it has run on >= 2 real ranks):
          "ingraph"   ONE graph, as on one GPU, with the collectives inside it: capture runs the Python once, so red_G's
                      autograd hooks fire and launch each bucket's all-reduce the moment its last gradient is written; RCCL's
                      stream is forked from / joined to the step's streams by the events torch.distributed records, and the
                      replayed graph overlaps every all-reduce with the rest of backward.  netD's forward / backward keep their side
                      branch; its (4.5 MB) all-reduce and Adam are issued on the step's stream at the join in front of the
                      adversarial term.  Verified on the one-GPU box with a one-rank RCCL group (tests/test_gpu_dp.py).
          "serial"    round 3's three graphs with the two all-reduces between them, nothing overlapped: the fallback.
          "segments"  opt-in: backward_G cut where a network's gradients are complete (loss networks + flowNetB | netG | flowNetF),
                      one graph per segment, the finished network's buckets reduced asynchronously while the next segment replays.
                      The segmented backward is exact (tests/test_dp_gloo.py: bit-equal to the unsegmented step, eager).  Round 4 saw
                      non-finite weight gradients in this mode's replay; round 5 found the cause outside it -- hipMemsetAsync nodes whose
                      fill pattern the runtime corrupts when graphs with side branches are replayed back to back
                      (profiles/r05_wgrad_nan_root_cause.txt); the library zero-fills with its own kernel since, and the mode keeps the
                      side streams (FFWM_SEG_STREAMS=0: without them).
        The batch is copied into static device buffers before every replay; the `titers` branch
        (< 20000 / >= 20000) is frozen at capture time -- re-capture when it flips."""
        assert self.device.type == "cuda" and self._graphs is None
        if not all(g.get("capturable", False) for o in (self.opt_F, self.opt_G, self.opt_D)
                   for g in getattr(o, "param_groups", [{}])):
            raise RuntimeError("capture() needs FFWMTrainer(..., capturable=True)")
        if self.dp_active:
            # "serial" unless asked otherwise: "ingraph" has only ever run with ONE RCCL rank (no multi-GPU box in four rounds), and a
            # passing probe graph says nothing about the whole step's cross-stream forks and joins between real ranks
            mode = mode or os.environ.get("FFWM_DP_CAPTURE") or "serial"
            if mode not in ("ingraph", "segments", "serial"):
                raise ValueError("capture mode %r" % (mode,))
        else:
            mode = "single"
        self.capture_mode = mode
        self.segmented = mode == "segments"
        # FFWM_INGRAPH_DSIDE: "main" (the D step's all-reduce and Adam at the join, on the step's stream; default), "side" (all of the D
        # step on the side branch), "0" (no side branch for D)
        dside = os.environ.get("FFWM_INGRAPH_DSIDE", "main")
        self._d_side = True if mode == "single" else ({"main": "reduce_on_main", "side": True}.get(dside, False) if mode == "ingraph" else False)
        if mode == "segments" and os.environ.get("FFWM_SEG_STREAMS", "1") == "0":
            self._streams_saved = (self.flow_stream, self.loss_streams)          # (diagnosis: the five-graph step on one stream)
            self.flow_stream, self.loss_streams = None, None
        self._static = {k: v.clone() for k, v in b.items()}
        sb = self._static
        hooks_on = mode == "ingraph" and os.environ.get("FFWM_INGRAPH_HOOKS", "1") == "1"     # (0: all-reduces at the end of backward, still in-graph)
        self.red_D.set_overlap(hooks_on and self._d_side != "reduce_on_main")
        self.red_G.set_overlap(hooks_on)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # MIOpen solver selection, allocator warm-up, Adam state
                self._seg_forward_and_D(sb)
                self._reduce_D()
                self._seg_stepD_and_G(sb)
                if self.segmented:
                    self._finish_backward_G()
                    self._drop_autograd_graph()
                self.red_G.finish()
                self._seg_stepG()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        # the fused BatchNorm + LeakyReLU modules count their batches on the host (norm.py): a replay runs no Python, so the
        # calls of ONE captured step are recorded here and added per replay (num_batches_tracked stays what nn.BatchNorm2d's is)
        from .norm import BatchNormLeakyReLU2d, HostCountBatchNorm2d, reset_scratch
        reset_scratch()                      # the capture makes (and fills, on every replay) its own BatchNorm scratch buffers
        fused = [m for net in (self.flowNetF, self.flowNetB, self.netG, self.netD) for m in net.modules()
                 if isinstance(m, (BatchNormLeakyReLU2d, HostCountBatchNorm2d))]
        before = [m._pending_batches for m in fused]
        graphs = []
        if mode in ("single", "ingraph"):
            kw = {}
            if mode == "ingraph":
                # the process group's watchdog thread polls the events of collectives in flight: let everything issued so far retire
                # before the capture starts, and do not let another thread's event query invalidate it
                import time
                time.sleep(0.3)
                kw["capture_error_mode"] = os.environ.get("FFWM_CAPTURE_ERROR_MODE", "thread_local")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **kw):
                self._seg_forward_and_D(sb)
                self._reduce_D()                 # one GPU: packs the D gradients into the flat array (a captured multi-tensor copy)
                self._seg_stepD_and_G(sb)
                self.red_G.finish()              # "ingraph": joins RCCL's stream back (the hooks launched the buckets during backward)
                self._seg_stepG()
            graphs = [g]
            self._captured_launch_log = (list(self.red_D.launch_log), list(self.red_G.launch_log))
        elif mode == "serial":
            g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # three graphs with the all-reduces between them; the fresh gradients are packed into the flat buckets INSIDE the graph that
            # produced them (a replayed gradient is not a new tensor to Python), finish() outside only reduces
            with torch.cuda.graph(g1):
                self._seg_forward_and_D(sb)
                self.red_D.pack_all()
            self.red_D.finish()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g2, pool=g1.pool()):
                self._seg_stepD_and_G(sb)
                self.red_G.pack_all()
            self.red_G.finish()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g3, pool=g1.pool()):
                self._seg_stepG()
            graphs = [g1, g2, g3]
        else:
            gs = [torch.cuda.CUDAGraph() for _ in range(5)]
            # (between the captures the collectives run for real, on whatever the buffers hold -- every rank issues the same sequence --
            # and are WAITED for: a backend thread still copying / synchronising while the next capture is open would invalidate it)
            with torch.cuda.graph(gs[0]):
                self._seg_forward_and_D(sb)
                self.red_D.pack_all()
            self.red_D.finish()
            torch.cuda.synchronize(self.device)
            pool = gs[0].pool() if os.environ.get("FFWM_SEG_OWN_POOLS", "0") != "1" else None          # (1: every graph its own pool, a diagnosis switch)
            with torch.cuda.graph(gs[1], pool=pool):       # D's Adam, the loss passes, backward down to the generated images + flowNetB
                self._seg_stepD_and_G(sb)
                self.red_G.pack_all(self.G_B)
            self.red_G.launch_group(self.G_B)
            self.red_G.wait_launched(host=True)
            with torch.cuda.graph(gs[2], pool=pool):
                self._bwd_netG()
                self.red_G.pack_all(self.G_NET)
            self.red_G.launch_group(self.G_NET)
            self.red_G.wait_launched(host=True)
            with torch.cuda.graph(gs[3], pool=pool):
                self._bwd_flowF()
                self._drop_autograd_graph()
                self.red_G.pack_all(self.G_F)
            self.red_G.launch_group(self.G_F)
            self.red_G.finish()
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(gs[4], pool=pool):
                self._seg_stepG()
            graphs = gs
        self.losses["D"] = self.loss_D
        self._bn_calls_per_replay = [(m, m._pending_batches - n0) for m, n0 in zip(fused, before) if m._pending_batches != n0]
        for m, n in self._bn_calls_per_replay:
            m._pending_batches -= n          # the capture pass itself executed nothing
        self._graphs = graphs
        self._frozen_branch = self.titers < 20000
        return self

    def _step_graphed(self, b, batch_increment):
        if (self.titers < 20000) != self._frozen_branch:
            raise RuntimeError("the titers branch flipped: call release_graphs() and capture() again")
        for k, v in self._static.items():
            if b[k] is not v:
                v.copy_(b[k], non_blocking=True)
        for opt in (self.opt_F, self.opt_G, self.opt_D):          # a learning-rate schedule reaches the replay through the device state
            if hasattr(opt, "sync_lr"):
                opt.sync_lr()
        gs = self._graphs
        if len(gs) == 1:
            gs[0].replay()
        elif len(gs) == 3:
            self.red_D.begin_replay()
            self.red_G.begin_replay()
            gs[0].replay()
            self.red_D.finish()
            gs[1].replay()
            self.red_G.finish()
            gs[2].replay()
        else:
            self.red_D.begin_replay()
            self.red_G.begin_replay()
            gs[0].replay()
            self.red_D.finish()
            gs[1].replay()
            self.red_G.launch_group(self.G_B)        # flowNetB's buckets travel while netG's backward replays
            gs[2].replay()
            self.red_G.launch_group(self.G_NET)      # netG's while flowNetF's backward replays
            gs[3].replay()
            self.red_G.launch_group(self.G_F)
            self.red_G.finish()
            gs[4].replay()
        for m, n in self._bn_calls_per_replay:
            m._pending_batches += n
        self.titers += batch_increment if batch_increment is not None else b["img_S"].size(0)
        return self.losses

    def release_graphs(self):
        self._graphs = None
        self._static = None
        self.segmented = False
        # the eager step that follows must not take the paths that are valid only inside an in-graph capture (event joins, the
        # ground-truth prefetch on side streams), and scratch buffers whose zero-fill was only CAPTURED must not be found in the cache
        self.capture_mode = None
        conv.GRAD_ARENA.active = False
        from .norm import reset_scratch
        reset_scratch()
        self._d_side = not self.dp_active
        if getattr(self, "_streams_saved", None) is not None:
            self.flow_stream, self.loss_streams = self._streams_saved
            self._streams_saved = None
        self.red_D.set_overlap(True)
        self.red_G.set_overlap(True)
        self.red_D.set_gather(True)
        self.red_G.set_gather(True)

    def set_side_streams(self, on):
        """Switch the side streams of the EAGER step off / back on (a captured step keeps the layout it was captured with).  With
        them off every kernel of a step runs alone on the step's stream: what a per-kernel measurement with HIP events needs --
        beside a side stream's kernels a launch shares the chip and its duration says nothing about the kernel (bench.py)."""
        if self._graphs is not None:
            raise RuntimeError("set_side_streams: release_graphs() first")
        if not on and getattr(self, "_streams_parked", None) is None:
            self._streams_parked = (self.flow_stream, self.loss_streams, self.d_stream, self.flowf_stream)
            self.flow_stream = self.loss_streams = self.d_stream = self.flowf_stream = None
        elif on and getattr(self, "_streams_parked", None) is not None:
            self.flow_stream, self.loss_streams, self.d_stream, self.flowf_stream = self._streams_parked
            self._streams_parked = None

    def loss_values(self):
        return {k: float(v.detach()) for k, v in self.losses.items()}

    def rank_spread(self):
        """max over the ranks of |weights - rank 0's weights|, per network (0.0 everywhere = the ranks are in lock step).  Two
        collectives per network over its flat parameter vector -- a check for after the first replays of a captured data-parallel step
        (bench.py), not for the timed region."""
        import torch.distributed as dist
        out = {}
        for name in self.MODEL_NAMES:
            net = getattr(self, name)
            flat = torch.cat([p.detach().flatten().float() for p in net.parameters()])
            if self.world_size > 1 and dist.is_available() and dist.is_initialized():
                hi, lo = flat.clone(), flat.clone()
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                out[name] = float((hi - lo).abs().max())
            else:
                out[name] = 0.0
            if not bool(torch.isfinite(flat).all()):
                out[name] = float("nan")
        return out

    # ------------------------------------------------------------------ checkpoint interchange / evaluation forward
    MODEL_NAMES = ("netG", "netD", "flowNetF", "flowNetB")      # ffwm_model.py:20-24

    def save_networks(self, save_dir, epoch):
        """BaseModel.save_networks (models/base_model.py:172-191): one '<epoch>_net_<name>.pth' state dict per
        network, CPU tensors, the reference's key names (weight_orig / weight_u / weight_v of the spectral-norm
        convs included -- the fused spectral norm keeps them), so either code base can load the other's files.
        Like the reference, optimizer state is not saved."""
        import os
        os.makedirs(save_dir, exist_ok=True)
        for name in self.MODEL_NAMES:
            sd = {k: v.detach().cpu() for k, v in getattr(self, name).state_dict().items()}
            torch.save(sd, os.path.join(save_dir, "%s_net_%s.pth" % (epoch, name)))

    def load_networks(self, load_dir, epoch, names=None):
        """BaseModel.load_networks (models/base_model.py:207-229)."""
        import os
        for name in (names or self.MODEL_NAMES):
            sd = torch.load(os.path.join(load_dir, "%s_net_%s.pth" % (epoch, name)), map_location=str(self.device))
            if hasattr(sd, "_metadata"):
                del sd._metadata
            getattr(self, name).load_state_dict(sd)

    def _no_eager_while_captured(self, what):
        """Vendor convolutions issued eagerly between replays of a captured step made the replays read freed memory (measured in
        round 3; not root-caused: MIOpen's workspaces of the eager calls and the graph's private pool meet
        somewhere).  Until that is understood the eager entry points refuse to run beside live graphs instead of silently corrupting
        the training run: release_graphs() first (and capture() again afterwards)."""
        if self._graphs is not None:
            raise RuntimeError("%s: the step is captured in hipGraphs; eager network passes beside the live graphs are not safe "
                               "-- call release_graphs() first, capture() again afterwards" % what)

    @torch.no_grad()
    def test_forward(self, b):
        """FFWMModel.test_forward (models/ffwm_model.py:183-189): flowNetF -> warped profile, netG -> frontal view and
        attention map, guided-filtered output.  Returns (fake_F128, img_GF128, img_S_warp, att)."""
        self._no_eager_while_captured("test_forward")
        flow_F128, flow_F64, flow_F32 = self.flowNetF(b["img_S"])
        img_S_warp = self.warp(b["img_S"], flow_F128)
        _, _, fake_F128, att = self.netG(b["img_S"], flow=[flow_F32, flow_F64, flow_F128], return_att=True)
        att = torch.mean(att[:, :64, :, :], (1,), keepdim=True)
        img_GF128 = self.gf[128](fake_F128, b["img_F"])
        return fake_F128, img_GF128, img_S_warp, att

    @torch.no_grad()
    def identity_feature(self, fake_F128):
        """FFWMModel.test (ffwm_model.py:191-202, crop=False): the LightCNN feature used for rank-1 matching."""
        self._no_eager_while_captured("identity_feature")
        _, fea, _ = self.lightCNN(torch.mean(fake_F128, dim=(1,), keepdim=True))
        return fea


class FlowNetTrainer(object):
    """FlowNet pre-training, the other trainer of the reference (train_flow.py -> models/flownet_model.py:57-78):
        flows = flowNet(img_S); fake_F = WarpNet(img_S, flow128)
        loss  = 20 * PerceptualCorrectness(img_F, img_S, flows[::-1], [2, 1, 0], mask)
              + 0.01 * MultiAffineRegularizationLoss({1: 7, 2: 5, 3: 3})(flows[::-1])
              + MultiScaleLDLoss(flows, lm_S, lm_F, gate)
        Adam(lr 4e-4, betas (0.5, 0.999)) on flowNet
    This is the only training path of the reference that runs the custom ops.  Here the warp of
    the VGG features and of the image go through the HIP warp kernels and the regulariser is the fused
    affine-regularisation kernel.  VGG19 is seeded random (no pretrained weights offline), frozen."""

    def __init__(self, device, world_size=1, seed=0, ngf=64, warp=None, fused_regularization=None, bucket_bytes=64 << 20,
                 routed=None, capturable=False):
        from .losses import MultiAffineRegularizationLoss, MultiScaleLDLoss, PerceptualCorrectness
        self.device = torch.device(device)
        torch.manual_seed(seed)
        self.warp = warp if warp is not None else WarpNet()
        if warp is None and self.device.type == "cuda":
            # independent warps issued together go out as ONE multi-problem launch (csrc/warp.hip)
            from .external_function import warp_many as _hip_warp_many
            self.warp_many = lambda feats, flows: _hip_warp_many(feats, flows, False)
        else:
            self.warp_many = lambda feats, flows: [self.warp(f, fl) for f, fl in zip(feats, flows)]
        if self.device.type == "cuda":
            from . import miopen_tuning
            miopen_tuning.install()
        self.flowNet = nets.FlowNet(ngf).to(self.device)
        self.vgg = nets.VGG19("relu3_1").to(self.device).eval()
        broadcast_module_state([self.flowNet, self.vgg])
        if fused_regularization is None:
            fused_regularization = self.device.type == "cuda"
        self.Regularization = MultiAffineRegularizationLoss({1: 7, 2: 5, 3: 3}, fused=fused_regularization)
        self.Correctness = PerceptualCorrectness(self.vgg, self.warp)
        self.criterionLD = MultiScaleLDLoss()
        # the same routes FFWMTrainer gives its flow nets (round 3): Winograd forward / data gradient, conv_fwd.hip for the stride-2 /
        # transposed / small-plane layers, direct kernels for the two-channel layers, tiled weight gradients, fused BatchNorm + LeakyReLU
        if routed is None:
            routed = self.device.type == "cuda" and warp is None
        self.routed_layers = 0
        if routed:
            from .conv import route_conv_bwd, route_conv_fwd, route_conv_winograd, route_flow_heads
            from .norm import fuse_bn_lrelu
            self.routed_layers = (route_conv_winograd(self.flowNet) + route_conv_fwd(self.flowNet) + route_flow_heads(self.flowNet)
                                  + route_conv_bwd(self.flowNet) + fuse_bn_lrelu(self.flowNet))
        self.world_size = world_size
        cap = bool(capturable) and self.device.type == "cuda"
        params = [p for n, p in self.flowNet.named_parameters() if not n.startswith("inter_conv_occ")]
        # (a captured step packs the gradients inside the graph that produced them, as FFWMTrainer)
        self.reducer = BucketedGradReducer(params, bucket_bytes=bucket_bytes, gather=True)
        if self.device.type == "cuda":
            from .optim import FlatAdam
            self.optimizer = FlatAdam(params, self.reducer, lr=0.0004, betas=(0.5, 0.999), capturable=cap)
        else:
            self.optimizer = torch.optim.Adam(params, lr=0.0004, betas=(0.5, 0.999))
        self.losses = {}
        self._graphs = None
        self._static = None

    def _seg_backward(self, b):
        if self.device.type == "cuda" and _GRAD_ARENA_ON:
            conv.GRAD_ARENA.begin(self.device)
        gate = torch.cat((b["gate"], b["gate"]), 2)
        flow, flow64, flow32 = self.flowNet(b["img_S"])
        self.fake_F = self.warp(b["img_S"], flow)
        flows = [flow, flow64, flow32]
        loss_cor = self.Correctness(b["img_F"], b["img_S"], flows[::-1], [2, 1, 0], norm_mask=b["mask_F"]) * 20
        loss_reg = self.Regularization(flows[::-1]) * 0.01
        loss_lm = self.criterionLD(flows, b["lm_S"], b["lm_F"], gate)
        loss = loss_cor + loss_lm + loss_reg
        self.reducer.zero_grad()
        try:
            loss.backward()
        finally:
            conv.GRAD_ARENA.end()
        self.losses = {"loss": loss.detach(), "cor": loss_cor.detach(), "reg": loss_reg.detach(), "lm": loss_lm.detach()}
        self.fake_F = self.fake_F.detach()

    def capture(self, b, warmup=3):
        """The step as hipGraphs, like FFWMTrainer.capture: ONE graph on one GPU, two around the gradient all-reduce with several
        ranks.  The eager step issues ~1400 launches of a few microseconds and is host-bound (GPU busy ~50 %)."""
        assert self.device.type == "cuda" and self._graphs is None
        if not getattr(self.optimizer, "param_groups", [{}])[0].get("capturable", False):
            raise RuntimeError("capture() needs FlowNetTrainer(..., capturable=True)")
        self._static = {k: v.clone() for k, v in b.items()}
        sb = self._static
        self.reducer.set_overlap(False)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._seg_backward(sb)
                self.reducer.finish()
                self.optimizer.step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        from .norm import BatchNormLeakyReLU2d, HostCountBatchNorm2d, reset_scratch
        reset_scratch()
        fused = [m for m in self.flowNet.modules() if isinstance(m, (BatchNormLeakyReLU2d, HostCountBatchNorm2d))]
        before = [m._pending_batches for m in fused]
        if self.world_size == 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._seg_backward(sb)
                self.reducer.finish()
                self.optimizer.step()
            graphs = [g]
        else:
            g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g1):
                self._seg_backward(sb)
                self.reducer.pack_all()
            self.reducer.finish()
            with torch.cuda.graph(g2, pool=g1.pool()):
                self.optimizer.step()
            graphs = [g1, g2]
        self._bn_calls_per_replay = [(m, m._pending_batches - n0) for m, n0 in zip(fused, before) if m._pending_batches != n0]
        for m, n in self._bn_calls_per_replay:
            m._pending_batches -= n          # the capture pass itself executed nothing
        self._graphs = graphs
        return self

    def release_graphs(self):
        self._graphs = None
        self._static = None
        from .norm import reset_scratch
        reset_scratch()                      # (scratch buffers first made inside the capture: their zero-fill was only captured)
        self.reducer.set_overlap(True)
        self.reducer.set_gather(True)

    def step(self, b):
        """optimize_parameters (flownet_model.py:74-78)."""
        if self._graphs is not None:
            for k, v in self._static.items():
                if b[k] is not v:
                    v.copy_(b[k], non_blocking=True)
            if hasattr(self.optimizer, "sync_lr"):
                self.optimizer.sync_lr()
            self.reducer.begin_replay()
            self._graphs[0].replay()
            if len(self._graphs) > 1:
                self.reducer.finish()
                self._graphs[1].replay()
            for m, n in self._bn_calls_per_replay:
                m._pending_batches += n
            return self.losses
        self._seg_backward(b)
        self.reducer.finish()
        self.optimizer.step()
        return self.losses

    def loss_values(self):
        return {k: float(v.detach()) for k, v in self.losses.items()}
