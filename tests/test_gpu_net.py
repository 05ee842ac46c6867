"""End-to-end GPU parity: the product ZSGNet / ZSGLoss / Evaluator / FusedAdam (HIP kernels through the C ABI) against
the CPU oracle on the same seeded inputs and against the reference goldens (g10).  fp32 tolerances: network outputs
abs 2e-3 (summation order through ~60 layers), gradients rel 1e-2 of their norm, loss rel 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import zsg_oracle as O  # noqa: E402

RATIOS, SCALES = O.default_ratios_scales()


@pytest.fixture(scope="module")
def Z():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim
    return config, evaluator, loss, mdl, optim


def build(Z, arch="resnet50", seed=7, **flags):
    config, evaluator, loss, mdl, optim = Z
    cfg = config.get_cfg(resnet_arch=arch, **flags)
    net = mdl.get_default_net(9, cfg)
    sd = O.seeded_state_dict(arch, seed)
    net.load_state_dict(sd)
    net.to("cuda")
    r, s = config.ratios_scales(cfg)
    return cfg, net, sd, loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)


def to_dev(bt):
    return {k: v.cuda() for k, v in bt.items()}


def rel_err(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tag,hw", [("e2e_128", 128), ("e2e_300", 300)])
def test_forward_backward_vs_reference_golden(Z, gold, tag, hw):
    g = gold("g10_" + tag)
    cfg, net, sd, lf, ev = build(Z, seed=int(g["seed"][0]))
    net.train()
    bt = O.synthetic_batch(2, hw, hw, seed=int(g["batch_seed"][0]))
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    out = net(inp)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist() and int(out["num_f_out"]) == len(g["feat_sizes"])
    att, bbx = out["att_out"].detach().cpu().numpy(), out["bbx_out"].detach().cpu().numpy()
    if hw == 128:
        np.testing.assert_allclose(att, g["att_out"], rtol=2e-3, atol=5e-3)
        np.testing.assert_allclose(bbx, g["bbx_out"], rtol=2e-3, atol=5e-3)
    else:
        np.testing.assert_allclose(att[:, ::7], g["att_out_s"], rtol=2e-3, atol=5e-3)
        np.testing.assert_allclose(bbx[:, ::7], g["bbx_out_s"], rtol=2e-3, atol=5e-3)
    ls = lf(out, inp)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(ls[k].item(), g[k], rtol=2e-4, err_msg=k)
    ls["loss"].backward()
    names = list(g["grad_names"])
    norms = dict(zip(names, g["grad_norms"]))
    bad = []
    for n, p in net.named_parameters():
        assert p.grad is not None, n
        gn = float(p.grad.double().norm())
        if abs(gn - norms[n]) > 3e-2 * norms[n] + 1e-7:      # B=2 train-mode BN: fp32 chaos (see test_forward_backward_vs_oracle)
            bad.append((n, gn, norms[n]))
    assert not bad, f"{len(bad)} gradient norms off: {bad[:8]}"
    for k in g.files:
        if k.startswith("grad__"):
            e = rel_err(dict(net.named_parameters())[k[6:]].grad.cpu(), torch.from_numpy(g[k]))
            assert e < 5e-2, f"{k}: relative error {e:.3g}"
    # running statistics after one train-mode forward
    st = net.state_dict()
    np.testing.assert_allclose(st["backbone.encoder.bn1.running_mean"].cpu().numpy(), g["rm_bn1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(st["backbone.encoder.layer4.2.bn3.running_var"].cpu().numpy(), g["rv_l4"], rtol=1e-3, atol=1e-6)
    assert int(st["backbone.encoder.bn1.num_batches_tracked"]) == 1
    em = ev(out, inp)
    assert em["Acc"].item() == g["Acc"] and em["MaxPos"].item() == g["MaxPos"]
    np.testing.assert_allclose(em["pred_scores"].cpu().numpy(), g["pred_scores"], rtol=1e-3)


def grad_tol(ec: float, g64: torch.Tensor) -> float:
    """Allowed |g_hip - g_fp64|: as good as the CPU-fp32 oracle (6x), or within 1.5 % of the gradient norm.  The second
    term covers ReLU-boundary flips: an activation within one fp32 ulp of 0 takes the other branch than in fp64, which
    zeroes one gradient element and perturbs its 9x9x256 receptive field through the head (observed: 2 flips in 3 M head
    activations move d(head conv0) by 4e-4 while the CPU oracle happened to have none; tests/debug_ssd.py)."""
    n = float(g64.norm())
    return max(6 * ec + 2e-4 * n, 1.5e-2 * n) + 1e-9


def fp64_twin(sd, bt, h0, c0, arch, anc, **kw):
    """The same oracle evaluated in float64: the ground truth both fp32 implementations are measured against."""
    sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    bt64 = {k: v.double() for k, v in bt.items()}
    ref = O.zsgnet_forward(sd64, bt64, h0.double(), c0.double(), arch=arch, rank=O.sort_rank(bt["qlens"]), **kw)
    ls = O.torch_loss(ref, bt["annot"], anc)
    ls["loss"].backward()
    return sd64, ref, ls


@pytest.mark.parametrize("arch,B,hw", [("resnet18", 2, 96), ("resnet50", 3, 160), ("resnet101", 1, 128)])
def test_forward_backward_vs_oracle(Z, arch, B, hw):
    """Other encoders / ragged sizes (odd pyramid shapes, query lengths 1..T with ties).  Train-mode BatchNorm on tiny
    batches amplifies fp32 rounding chaotically, so the yard-stick is the float64 oracle: the HIP result must be as
    close to it as the CPU fp32 oracle is (within 6x + 2e-4 of the gradient norm)."""
    cfg, net, sd, lf, ev = build(Z, arch=arch, seed=11)
    net.train()
    bt = O.synthetic_batch(B, hw, hw + 32, seed=5, tmax=13)
    bt["qlens"][-1] = bt["qlens"][0]
    gq = torch.Generator().manual_seed(2)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch=arch)
    assert out["feat_sizes"].tolist() == ref["feat_sizes"].tolist()
    fs = [tuple(r) for r in ref["feat_sizes"].tolist()]
    anc = torch.from_numpy(O.create_anchors(fs, RATIOS, SCALES).astype(np.float32))
    sd64, ref64, ls64 = fp64_twin(sd, bt, h0, c0, arch, anc)
    o_gpu = out["att_bbx_out"].detach().cpu().double()
    o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach().double()
    o_64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
    e_gpu, e_cpu = float((o_gpu - o_64).abs().max()), float((o_cpu - o_64).abs().max())
    assert e_gpu <= 6 * e_cpu + 1e-4, f"forward: HIP err {e_gpu:.3g} vs fp64, CPU fp32 err {e_cpu:.3g}"
    lr = O.torch_loss(ref, bt["annot"], anc)
    ls = lf(out, inp)
    np.testing.assert_allclose(ls["loss"].item(), ls64["loss"].item(), rtol=2e-4)
    lr["loss"].backward()
    ls["loss"].backward()
    torch.cuda.synchronize()
    worst = []
    for n, p in net.named_parameters():
        g64 = sd64[n].grad.flatten()
        eg = float((p.grad.cpu().double().flatten() - g64).norm())
        ec = float((sd[n].grad.double().flatten() - g64).norm())
        if eg > grad_tol(ec, g64):
            worst.append((n, eg / (float(g64.norm()) + 1e-30), ec / (float(g64.norm()) + 1e-30)))
    assert not worst, f"gradient error vs fp64 (HIP rel, CPU-fp32 rel): {worst[:8]}"


def test_600x600_four_level_pyramid(Z):
    """resize_img == [600, 600] (BASELINE configs[4]): the FPN returns P4..P7 only — P3 is computed and dropped, no pooled
    1x1 level (fpn_resnet.py:173-178) — 4 levels, A = 17 370 anchors.  ResNet-18, B=1, vs the oracle and its fp64 twin."""
    cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=21, resize_img=[600, 600])
    net.train()
    bt = O.synthetic_batch(1, 600, 600, seed=8, tmax=9)
    gq = torch.Generator().manual_seed(6)
    h0, c0 = torch.randn(2, 1, 128, generator=gq), torch.randn(2, 1, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    assert out["feat_sizes"].tolist() == [[38, 38], [19, 19], [10, 10], [5, 5]] and int(out["num_f_out"]) == 4
    assert out["att_out"].shape == (1, 17370, 1) and out["bbx_out"].shape == (1, 17370, 4)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet18", six_hundred=True)
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(600, 600, True), RATIOS, SCALES).astype(np.float32))
    sd64, ref64, ls64 = fp64_twin(sd, bt, h0, c0, "resnet18", anc, six_hundred=True)
    o_gpu = out["att_bbx_out"].detach().cpu().double()
    o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach().double()
    o_64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
    e_gpu, e_cpu = float((o_gpu - o_64).abs().max()), float((o_cpu - o_64).abs().max())
    assert e_gpu <= 6 * e_cpu + 1e-4, f"forward: HIP err {e_gpu:.3g} vs fp64, CPU fp32 err {e_cpu:.3g}"
    ls = lf(out, inp)
    np.testing.assert_allclose(ls["loss"].item(), ls64["loss"].item(), rtol=2e-4)
    O.torch_loss(ref, bt["annot"], anc)["loss"].backward()
    ls["loss"].backward()
    torch.cuda.synchronize()
    worst = []
    for n, p in net.named_parameters():
        g64 = sd64[n].grad
        if g64 is None:                          # P3_2 feeds nothing at 600x600: the reference leaves its gradient unset
            assert n.startswith("backbone.fpn.P3_") and float(p.grad.abs().max()) == 0.0, n
            continue
        g64 = g64.flatten()
        eg = float((p.grad.cpu().double().flatten() - g64).norm())
        ec = float((sd[n].grad.double().flatten() - g64).norm())
        if eg > grad_tol(ec, g64):
            worst.append((n, eg / (float(g64.norm()) + 1e-30), ec / (float(g64.norm()) + 1e-30)))
    assert not worst, f"gradient error vs fp64 (HIP rel, CPU-fp32 rel): {worst[:8]}"
    em = ev(out, inp)
    e = O.zsg_eval(o_cpu[..., 4].numpy().astype(np.float32), o_cpu[..., :4].numpy().astype(np.float32), bt["annot"].numpy(), bt["img_size"].numpy(),
                   anc.numpy())
    assert 0 <= int(em["MaxPos"].item()) <= 1 and em["pred_boxes"].shape == (1, 4) and np.isfinite(e["pred_scores"]).all()


def test_ssd_vgg_backbone_vs_golden_and_fp64(Z, gold):
    """BASELINE configs[3] model family (SSD-VGG16 backbone, ssd_vgg.py) at B=1: reference golden + fp64 yard-stick."""
    config, evaluator, loss, mdl, optim = Z
    g = gold("g11_ssd")
    cfg = config.get_cfg(mdl_to_use="ssd_vgg")
    net = mdl.get_default_net(9, cfg)
    sd = O.seeded_ssd_state_dict(int(g["seed"][0]))
    net.load_state_dict(sd)
    net.to("cuda").train()
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    bt = O.synthetic_batch(1, 300, 300, seed=int(g["batch_seed"][0]))
    h0, c0 = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    np.testing.assert_allclose(out["att_out"].detach().cpu().numpy()[:, ::7], g["att_out_s"], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out["bbx_out"].detach().cpu().numpy()[:, ::7], g["bbx_out_s"], rtol=2e-3, atol=2e-3)
    ls = lf(out, inp)
    np.testing.assert_allclose(ls["loss"].item(), g["loss"], rtol=2e-4)
    ls["loss"].backward()
    torch.cuda.synchronize()
    for k, v in sd.items():
        v.requires_grad_(True)
    anc = torch.from_numpy(O.create_anchors([tuple(x) for x in g["feat_sizes"].tolist()], RATIOS, SCALES).astype(np.float32))
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch="ssd_vgg")
    O.torch_loss(ref, bt["annot"], anc)["loss"].backward()
    sd64, _, _ = fp64_twin(sd, bt, h0, c0, "ssd_vgg", anc)
    unused = set(g["unused"])
    worst = []
    for n, p in net.named_parameters():
        if n in unused:
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        g64 = sd64[n].grad.flatten()
        eg = float((p.grad.cpu().double().flatten() - g64).norm())
        ec = float((sd[n].grad.double().flatten() - g64).norm())
        if eg > grad_tol(ec, g64):
            worst.append((n, eg / (float(g64.norm()) + 1e-30), ec / (float(g64.norm()) + 1e-30)))
    assert not worst, f"SSD gradient error vs fp64 (HIP rel, CPU-fp32 rel): {worst[:8]}"


def test_eval_mode_and_state_dict_roundtrip(Z):
    cfg, net, sd, lf, ev = build(Z, seed=3)
    bt = O.synthetic_batch(2, 128, 128, seed=9)
    gq = torch.Generator().manual_seed(4)
    h0, c0 = torch.randn(2, 2, 128, generator=gq), torch.randn(2, 2, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    for k in sd:                                   # non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape) * 0.05
        if k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape) + 0.5
    net.load_state_dict(sd)
    net.eval()
    with torch.no_grad():
        out = net(inp)
    ref = O.zsgnet_forward({k: v.clone() for k, v in sd.items()}, bt, h0, c0, training=False)
    scale = float(ref["bbx_out"].abs().max())           # random running statistics blow the activations up: scale-relative
    assert float((out["bbx_out"].cpu() - ref["bbx_out"]).abs().max()) < 2e-4 * scale
    assert float((out["att_out"].cpu() - ref["att_out"]).abs().max()) < 2e-4 * float(ref["att_out"].abs().max())
    back = {k: v.cpu() for k, v in net.state_dict().items()}
    assert set(back) == set(sd)
    for k in sd:
        assert torch.equal(back[k], sd[k]), k
    with pytest.raises(RuntimeError):
        net.load_state_dict({"bogus": torch.zeros(1)})
    ddp_style = {"module." + k: v for k, v in sd.items()}
    ddp_style["module.backbone.encoder.fc.weight"] = torch.zeros(1000, 2048)
    net.load_state_dict(ddp_style)


def test_train_steps_match_oracle_and_reduce_loss(Z):
    """3 full steps (zero_grad -> fwd -> loss -> bwd -> fused Adam -> eval, utils.py:407-414) against the CPU oracle
    stepping torch.optim.Adam on the same batch; then the loss must go down."""
    config, evaluator, loss, mdl, optim = Z
    cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=21)
    net.train()
    opt = optim.FusedAdam(net, lr=1e-3, betas=(0.9, 0.99))
    params = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    opt_ref = torch.optim.Adam(list(params.values()), lr=1e-3, betas=(0.9, 0.99))
    bt = O.synthetic_batch(2, 96, 96, seed=77)
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(96, 96), RATIOS, SCALES).astype(np.float32))
    inp = to_dev(bt)
    gq = torch.Generator().manual_seed(6)
    losses = []
    for it in range(3):
        h0, c0 = torch.randn(2, 2, 128, generator=gq), torch.randn(2, 2, 128, generator=gq)
        inp["h0"], inp["c0"] = h0, c0
        opt.zero_grad()
        out = net(inp)
        ls = lf(out, inp)
        ls["loss"].mean().backward()
        opt.step()
        em = ev(out, inp)
        lr, _ = O.cpu_train_step(params, buffers, opt_ref, bt, h0, c0, anc, arch="resnet18")
        losses.append((ls["loss"].item(), lr["loss"].item()))
        assert 0.0 <= em["Acc"].item() <= 1.0
    np.testing.assert_allclose(losses[0][0], losses[0][1], rtol=2e-4)      # before any update: same weights
    for a, b in losses[1:]:                  # Adam's first steps are ~sign(g): rounding-level gradient differences on
        np.testing.assert_allclose(a, b, rtol=3e-2)   # near-zero elements flip their update, so later losses only agree to ~1 %
    assert losses[-1][0] < losses[0][0]
    got = net.state_dict()
    for k in ("att_reg_box.5.bias", "backbone.fpn.P3_2.weight", "backbone.encoder.layer2.0.conv1.weight", "lstm.weight_hh_l0"):
        e = rel_err(got[k].cpu() - sd[k], params[k].detach() - sd[k])
        assert e < 1e-1, f"{k}: parameter update differs from the oracle by {e:.3g}"


ABLATIONS = {"lang_blind": dict(use_lang=False), "img_blind": dict(use_img=False),
             "both_blind": dict(use_lang=False, use_img=False), "do_norm": dict(do_norm=True),
             "two_heads": dict(use_same_atb=False)}


@pytest.mark.parametrize("tag", sorted(ABLATIONS))
def test_ablation_variants_vs_reference_golden(Z, gold, tag):
    """Blind heads (mdl.py:199-210, 363-375) and do_norm (mdl.py:118-130) against the imported reference (g12):
    outputs abs 5e-3, loss rel 2e-4, gradient norms within 3 % (B=2 train-mode BN), unused parameters get exactly 0."""
    config, evaluator, loss, mdl, optim = Z
    g = gold("g12_" + tag)
    cfg = config.get_cfg(**ABLATIONS[tag])
    net = mdl.get_default_net(9, cfg)
    net.load_state_dict(O.seeded_state_dict("resnet50", int(g["seed"][0]), head_in=int(g["head_in"][0]), same_atb=bool(cfg["use_same_atb"])))
    if "keys" in g.files:      # (the golden's stand-in encoder carries an unused FPN of its own)
        assert set(net.state_dict().keys()) == {str(k) for k in g["keys"] if not str(k).startswith("backbone.encoder.fpn.")}
    net.to("cuda").train()
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    inp = to_dev(O.synthetic_batch(2, 128, 128, seed=int(g["batch_seed"][0])))
    inp["h0"], inp["c0"] = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    out = net(inp)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    np.testing.assert_allclose(out["att_out"].detach().cpu().numpy(), g["att_out"], rtol=2e-3, atol=5e-3)
    np.testing.assert_allclose(out["bbx_out"].detach().cpu().numpy(), g["bbx_out"], rtol=2e-3, atol=5e-3)
    ls = lf(out, inp)
    np.testing.assert_allclose(ls["loss"].item(), g["loss"], rtol=2e-4)
    ls["loss"].backward()
    norms = dict(zip(g["grad_names"], g["grad_norms"]))
    unused = set(g["unused"])
    bad = []
    for n, p in net.named_parameters():
        gn = float(p.grad.double().norm())
        if n in norms:
            if abs(gn - norms[n]) > 3e-2 * norms[n] + 1e-7:
                bad.append((n, gn, norms[n]))
        else:
            assert n in unused and gn == 0.0, f"{n}: the reference leaves this gradient unset, got norm {gn}"
    assert not bad, f"{len(bad)} gradient norms off: {bad[:8]}"
    P = dict(net.named_parameters())
    for k in g.files:
        if k.startswith("grad__"):
            got = P[k[6:-2]].grad.cpu()[::8, ::5] if k.endswith("weight_s") else P[k[6:]].grad.cpu()
            e = rel_err(got, torch.from_numpy(g[k]))
            assert e < 5e-2, f"{k}: relative error {e:.3g}"
    np.testing.assert_allclose(net.state_dict()["backbone.encoder.bn1.running_mean"].cpu().numpy(), g["rm_bn1"], rtol=1e-4, atol=1e-6)


def test_query_length_buckets_share_one_plan(Z):
    """The collater cuts qvec to the batch's longest query (dat_loader.py:187-196): T varies per batch.  Plans are built
    per geometry with T bucketed (20 / 50); a shorter qvec is zero-padded into the same plan and gives the same output."""
    cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=5)
    net.eval()
    bt = O.synthetic_batch(3, 96, 96, seed=12, tmax=11)
    tmax = int(bt["qlens"].max())
    h0, c0 = torch.zeros(2, 3, 128), torch.zeros(2, 3, 128)
    outs = []
    with torch.no_grad():
        for T in (20, tmax, 15):
            inp = to_dev({**bt, "qvec": bt["qvec"][:, :T].contiguous()})
            inp["h0"], inp["c0"] = h0, c0
            outs.append(net(inp)["att_bbx_out"].clone())
    assert len(net._plans) == 1, "T = 20, 15 and the batch maximum must share one plan"
    tol = 1e-4 * float(outs[0].abs().max())          # split-K launches use fp32 atomics: equal up to summation order
    assert float((outs[0] - outs[1]).abs().max()) < tol and float((outs[0] - outs[2]).abs().max()) < tol
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet18", training=False)
    o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2)
    assert float((outs[1].cpu() - o_cpu).abs().max()) < 2e-3
    with torch.no_grad():
        long = to_dev({**bt, "qvec": torch.cat([bt["qvec"], torch.zeros(3, 17, 300)], 1)})          # T = 37 -> the 50-token bucket
        long["h0"], long["c0"] = h0, c0
        o37 = net(long)["att_bbx_out"]
    assert len(net._plans) == 2 and float((o37 - outs[0]).abs().max()) < tol


def test_caller_owns_the_output_tensor(Z, monkeypatch):
    """ZSGNet.forward returns a tensor of its own, as the reference's module does (mdl.py:338-403): the launches that produce the
    [B, A, 5] output write straight into a fresh tensor (no copy out of the plan's buffers), so a result kept across later forwards
    keeps its values; the copying path (ZSG_FRESH_OUT=0 -> _out_slots() is None) gives the same numbers."""
    cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=3)
    h0, c0 = torch.zeros(2, 2, 128), torch.zeros(2, 2, 128)
    for training in (False, True):
        net.train(training)
        outs, keep = [], []
        for seed in (21, 22, 21):
            inp = to_dev(O.synthetic_batch(2, 96, 96, seed=seed))
            inp["h0"], inp["c0"] = h0, c0
            with torch.set_grad_enabled(training):
                o = net(inp)["att_bbx_out"]
            outs.append(o)
            keep.append(o.detach().clone())
        plan = [p for k, p in net._plans.items() if k[-1] == training][0]
        assert plan._out_slots(), "the default configuration must take the direct-output path"
        assert len({o.data_ptr() for o in outs}) == 3, "every forward returns its own tensor"
        for o, k in zip(outs, keep):
            assert torch.equal(o.detach(), k), "a later forward must not touch an earlier result"
        assert not torch.equal(keep[0], keep[1])
        if not training:      # the same input gives the same output (up to the summation order of atomic split-K launches)
            tol = 1e-4 * float(keep[0].abs().max())
            assert float((keep[0] - keep[2]).abs().max()) < tol
            ref_eval = keep[2]
    # the copying path (ZSG_FRESH_OUT=0, read when a plan first runs): a second module with the same weights gives the same numbers
    monkeypatch.setenv("ZSG_FRESH_OUT", "0")
    cfg2, net2, _, _, _ = build(Z, arch="resnet18", seed=3)
    net2.eval()
    with torch.no_grad():
        o2 = net2(inp)["att_bbx_out"]
    assert next(iter(net2._plans.values()))._out_slots() is None
    assert float((o2 - ref_eval).abs().max()) < 1e-4 * float(ref_eval.abs().max())


def test_one_launch_input_staging_equals_torch_copies(Z, monkeypatch):
    """run_forward's one-launch input staging (zsg_stage_inputs: qvec into the zero-padded token bucket, qlens int64 / float, the
    host-drawn h0 | c0 from a pinned ring slot, num_batches_tracked += 1) stages exactly what the separate torch copies of rounds 1-4
    staged: staged buffers, BatchNorm counters and network outputs bit-identical, for T below the bucket size, across more forwards
    than the ring has slots."""
    config, evaluator, loss, mdl, optim = Z
    cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=5)
    net.train()
    bt = O.synthetic_batch(3, 96, 96, seed=12, tmax=13)
    T = int(bt["qlens"].max())
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(mdl, "STAGE_INPUTS", mode)
        net.load_state_dict(sd)
        outs = []
        torch.manual_seed(3)
        for it in range(11):
            inp = to_dev({**bt, "qvec": bt["qvec"][:, :T].contiguous()})
            if it % 2:
                inp["qlens"] = inp["qlens"].long()
            h0, c0 = net.lstm_init_hidden(3)          # the reference's two host draws (mdl.py:279-294)
            inp["h0"], inp["c0"] = h0, c0
            with torch.no_grad():
                o = net(inp)["att_bbx_out"].clone()
            plan = next(iter(net._plans.values()))
            outs.append((o, plan.in_qvec.clone(), plan.in_qlens.clone(), plan.in_hc.clone(), h0, c0))
        res[mode] = (outs, net._nbt.clone())
    assert int(res[True][1][0]) == int(res[False][1][0]) and torch.equal(res[True][1], res[False][1]), "num_batches_tracked"
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]), "same host draws in both modes"
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]), "staged qvec / qlens / h0|c0"
        assert torch.equal(a[3].cpu().view(2, -1), torch.stack([a[4], a[5]]).view(2, -1))
        assert float((a[0] - b[0]).abs().max()) <= 1e-5 * float(b[0].abs().max())
