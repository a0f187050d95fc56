"""One FFWM train step on CPU (batch 1, narrow FlowNets) with the torch stand-ins injected for the
warp ops: checks the step's structure -- D then G update, requires_grad toggling, frozen feature
nets, unused FlowNet branch excluded, both titers branches -- not its speed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_refs  # noqa: E402


def _snapshot(mods):
    return [p.detach().clone() for m in mods for p in m.parameters()]


def test_train_step_updates_the_right_parameters():
    from ffwm_amd import trainer
    torch.set_num_threads(8)
    t = trainer.FFWMTrainer("cpu", seed=0, titers=0, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat,
                            ngf=8)
    batch = trainer.synthetic_batch(1, "cpu", seed=1)
    # BatchNorm in train mode needs > 1 value per channel at FlowNet's 1x1 bottleneck: use batch 2 there
    batch = {k: torch.cat((v, v.flip(-1) if v.dtype.is_floating_point else v), 0) for k, v in batch.items()}
    before = {k: _snapshot([getattr(t, k)]) for k in ("flowNetF", "flowNetB", "netG", "netD", "lightCNN", "vgg")}
    losses = t.step(batch)
    vals = t.loss_values()
    assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), vals
    assert set(vals) == {"G", "D", "l1", "iden", "illu", "adv", "prc", "fc"}
    after = {k: _snapshot([getattr(t, k)]) for k in before}
    for k in ("flowNetF", "flowNetB", "netG", "netD"):
        changed = sum(int(not torch.equal(a, b)) for a, b in zip(before[k], after[k]))
        assert changed > 0, k
    for k in ("lightCNN", "vgg"):
        assert all(torch.equal(a, b) for a, b in zip(before[k], after[k])), k
    # the never-used occlusion branch is neither updated nor in an optimizer
    occ = [p for n, p in t.flowNetF.named_parameters() if n.startswith("inter_conv_occ")]
    held = {id(p) for g in t.opt_F.param_groups for p in g["params"]}
    assert occ and not any(id(p) in held for p in occ)
    assert all(not p.requires_grad for p in t.netD.parameters())      # left frozen after the G step
    assert t.titers == 2
    # second step on the guided-filter branch (titers >= 20000, ffwm_model.py:97-105)
    t.titers = 20000
    t.step(batch)
    assert all(torch.isfinite(torch.tensor(v)) for v in t.loss_values().values())


def test_batched_loss_passes_are_result_preserving():
    """One VGG pass over the five 32 x 32 pairs / one LightCNN pass over the distinct outputs must give
    the losses of the separate calls (no batch statistics in either net)."""
    from ffwm_amd import trainer
    torch.set_num_threads(8)
    vals = []
    for batched in (False, True):
        t = trainer.FFWMTrainer("cpu", seed=0, titers=0, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat,
                                ngf=8, batched_losses=batched)
        batch = trainer.synthetic_batch(1, "cpu", seed=1)
        batch = {k: torch.cat((v, v.flip(-1) if v.dtype.is_floating_point else v), 0) for k, v in batch.items()}
        t.step(batch)
        vals.append(t.loss_values())
        t.titers = 20000
        t.step(batch)
        vals.append(t.loss_values())
    for a, b in ((vals[0], vals[2]), (vals[1], vals[3])):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-4 * (1 + abs(a[k])), (k, a[k], b[k])


def test_flownet_pretraining_step_on_cpu():
    """FlowNetModel.optimize_parameters (flownet_model.py:57-78) on CPU with the torch stand-in warp and the
    op-composition regulariser replaced by its fused form being GPU-only: here fused_regularization=False is
    not available either (the composed ops are GPU kernels), so the regulariser is stubbed; what is checked is
    the structure: three flows, the three loss terms, one Adam step on the used FlowNet parameters only."""
    from ffwm_amd import trainer
    torch.set_num_threads(8)
    t = trainer.FlowNetTrainer("cpu", seed=0, ngf=8, warp=torch_refs.warp, fused_regularization=False)
    t.Regularization = lambda flows: sum((f[:, :, 1:] - f[:, :, :-1]).abs().mean() for f in flows)   # CPU stand-in
    batch = trainer.synthetic_batch(2, "cpu", seed=2)
    before = [p.detach().clone() for p in t.flowNet.parameters()]
    t.step(batch)
    vals = t.loss_values()
    assert set(vals) == {"loss", "cor", "reg", "lm"}
    assert all(torch.isfinite(torch.tensor(v)) for v in vals.values()), vals
    after = list(t.flowNet.parameters())
    changed = {n for (n, _), a, b in zip(t.flowNet.named_parameters(), before, after) if not torch.equal(a, b)}
    assert changed and not any(n.startswith("inter_conv_occ") for n in changed)


def test_checkpoint_interchange_roundtrip_and_reference_key_names(tmp_path):
    """save_networks / load_networks use the reference's file naming ('<epoch>_net_<name>.pth', base_model.py:172-229)
    and key names (the goldens hold the reference modules' own state-dict keys); test_forward runs the evaluation path
    (ffwm_model.py:183-202)."""
    import os
    from ffwm_amd import trainer
    torch.set_num_threads(8)
    gold = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_modules.pt"))
    a = trainer.FFWMTrainer("cpu", seed=1, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat)
    a.save_networks(str(tmp_path), "latest")
    assert sorted(os.listdir(str(tmp_path))) == ["latest_net_flowNetB.pth", "latest_net_flowNetF.pth", "latest_net_netD.pth",
                                                 "latest_net_netG.pth"]
    assert sorted(torch.load(os.path.join(str(tmp_path), "latest_net_netG.pth")).keys()) == gold["ffwm_keys"]
    assert sorted(torch.load(os.path.join(str(tmp_path), "latest_net_flowNetF.pth")).keys()) == gold["flownet64_keys"]
    assert sorted(torch.load(os.path.join(str(tmp_path), "latest_net_netD.pth")).keys()) == gold["netD_keys"]
    b = trainer.FFWMTrainer("cpu", seed=2, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat)
    b.load_networks(str(tmp_path), "latest")
    for name in a.MODEL_NAMES:
        for (k, v), (_, w) in zip(getattr(a, name).state_dict().items(), getattr(b, name).state_dict().items()):
            assert torch.equal(v, w), (name, k)
    batch = trainer.synthetic_batch(1, "cpu", seed=3)
    for t in (a, b):
        for n in t.MODEL_NAMES:
            getattr(t, n).eval()
    fa, fb = a.test_forward(batch), b.test_forward(batch)
    # (a randomly initialised netG overflows in eval mode -- running statistics of an untrained net: NaNs compare equal)
    assert all(torch.allclose(x, y, atol=1e-5, equal_nan=True) for x, y in zip(fa, fb))
    assert fa[0].shape == (1, 3, 128, 128) and fa[3].shape == (1, 1, 128, 128)
    assert a.identity_feature(fa[0]).shape[0] == 1


def _eval_golden():
    here = os.path.dirname(os.path.abspath(__file__))
    return torch.load(os.path.join(here, "golden", "reference_eval.pt"), weights_only=False), os.path.join(here, "golden", "ckpt")


def _packed_close(got, p, tol):
    got = got.detach().cpu()
    assert tuple(got.shape) == p["shape"], (tuple(got.shape), p["shape"])
    s = p["step"]
    scale = 1.0 + float(p["sample"].abs().max())
    d = float((got[..., ::s, ::s] - p["sample"]).abs().max())
    assert d <= tol * scale, "max abs diff %.3e > %.3e" % (d, tol * scale)
    n = got.numel()
    assert abs(float(got.double().sum()) - p["sum"]) / n <= tol * scale, "mean drift"
    assert abs(float(got.double().abs().sum()) - p["abs_sum"]) / n <= tol * scale, "mean |.| drift"
    return d / scale


def _gf_close(gf, gold):
    """The composed img_GF128 (round 6: generated image of std 0.14): within 2e-4 of the reference's fp32 output -- the distance of that
    output from the reference's own float64 evaluation, the cumsum arithmetic's fp32 floor -- and at least as close to float64 as 1.5 x
    the reference's fp32 result is (round 5: a bound of 0.13 on a near-constant image)."""
    assert gold["fake_F128_std"] >= 0.1 and gold["img_GF128_ref_fp32_vs_fp64"] <= 3e-4
    _packed_close(gf, gold["test_forward"]["img_GF128"], 2e-4)
    p = gold["img_GF128_fp64"]
    d64 = float((gf.detach().cpu().double()[..., ::p["step"], ::p["step"]] - p["sample"]).abs().max())
    assert d64 <= max(1.5 * gold["img_GF128_ref_fp32_vs_fp64"], 1e-4), (d64, gold["img_GF128_ref_fp32_vs_fp64"])
    return d64


def _prepare_reference_checkpoints(tmp_path, gold, ckpt_dir):
    """A checkpoint directory as the reference's BaseModel.save_networks leaves it: flowNetF = the file the REFERENCE's FlowNet(4) wrote
    (tests/golden/ckpt, committed); netG / netD (65 MB / 4.5 MB: they do not travel) re-derived with the closed form the reference's
    modules were filled with, under the reference's key names -- and held, key by key, to the shapes and float64 checksums of the
    reference's own state dicts."""
    import shutil
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import fill
    from ffwm_amd import nets
    ep = gold["epoch"]
    shutil.copy(os.path.join(ckpt_dir, "%s_net_flowNetF.pth" % ep), str(tmp_path))
    for name, mod in (("netG", nets.FFWM(sn=True)), ("netD", nets.MSDiscriminator(128, sigmoid=False))):
        fill.fill_module(mod)
        if name == "netG":
            fill.boost_output_gain(mod)                  # as make_eval_golden.py did to the reference's netG (image heads x 40)
        sd = mod.state_dict()
        ref = gold["state"][name]
        assert list(sd.keys()) == list(ref.keys()), name          # the reference's names in the reference's ORDER
        for k, v in sd.items():
            shape, ssum, sabs = ref[k]
            assert tuple(v.shape) == shape, (name, k)
            assert abs(float(v.double().sum()) - ssum) <= 1e-6 * (1 + sabs) and abs(float(v.double().abs().sum()) - sabs) <= 1e-6 * (1 + sabs), (name, k)
        torch.save({k: v.detach().cpu() for k, v in sd.items()}, os.path.join(str(tmp_path), "%s_net_%s.pth" % (ep, name)))
    return ep


def test_a_checkpoint_written_by_the_reference_loads_and_test_forward_matches_the_reference(tmp_path):
    """(1) '7_net_flowNetF.pth' was written by the REFERENCE's FlowNet(4) through the code of BaseModel.save_networks
    (tests/golden/make_eval_golden.py); load_networks must take it as it is.  (2) The composed FFWMModel.test_forward
    (models/ffwm_model.py:183-189: flowNetF -> WarpNet -> netG -> GuidedFilter(32)) of the reference's modules is a committed fixture;
    FFWMTrainer.test_forward on the loaded networks must reproduce it (CPU here, torch stand-ins for the warps; the GPU twin of this
    test runs the HIP kernels: tests/test_gpu_nets_golden.py)."""
    from ffwm_amd import trainer
    torch.set_num_threads(8)
    gold, ckpt_dir = _eval_golden()
    ep = _prepare_reference_checkpoints(tmp_path, gold, ckpt_dir)
    ref_sd = torch.load(os.path.join(ckpt_dir, "%s_net_flowNetF.pth" % ep))
    assert {k: (tuple(v.shape), float(v.double().sum())) for k, v in ref_sd.items()} == \
           {k: (s, x) for k, (s, x, _) in gold["state"]["flowNetF"].items()}          # the committed file IS the reference's state
    t = trainer.FFWMTrainer("cpu", seed=5, ngf=4, warp=torch_refs.warp, warp_flipcat=torch_refs.warp_flipcat)
    t.load_networks(str(tmp_path), ep, names=("flowNetF", "netG", "netD"))
    for k, v in t.flowNetF.state_dict().items():
        assert torch.equal(v.cpu(), ref_sd[k]), k
    for n in t.MODEL_NAMES:
        getattr(t, n).eval()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import fill
    b = {"img_S": fill.image(2, 3, 128, 128, "eval_img_S"), "img_F": fill.image(2, 3, 128, 128, "eval_img_F")}
    fake, gf, warped, att = t.test_forward(b)
    tf = gold["test_forward"]
    with torch.no_grad():
        flows = t.flowNetF(b["img_S"])
    for got, key in zip(flows, ("flow_F128", "flow_F64", "flow_F32")):
        _packed_close(got, tf[key], 1e-5)
    _packed_close(warped, tf["img_S_warp"], 1e-5)
    _packed_close(fake, tf["fake_F128"], 1e-4)
    _packed_close(att, tf["att"], 1e-4)
    # the guided filter: on the input pair, and as the end of the composed forward -- round 6: the fixture's generated image has a std of
    # 0.14 (fill.boost_output_gain), the filter is well conditioned and img_GF128 is held to the filter's fp32 floor (_gf_close)
    with torch.no_grad():
        _packed_close(t.gf[128](b["img_S"], b["img_F"]), gold["gf128_on_images"], 1e-4)
    _gf_close(gf, gold)
    with torch.no_grad():
        score = t.netD(fake)
    assert float((score - gold["netD_score_of_fake"]).abs().max()) <= 1e-4 * (1 + float(gold["netD_score_of_fake"].abs().max()))
