"""Parity at the shapes the benchmark and BASELINE.json's configs actually run (not toy shapes):
  * configs[1]: ResNet-50 FPN, 300x300, B=16 — against the REFERENCE golden g10_e2e_300_b16 (outputs, loss, evaluator,
    every gradient norm, sampled gradients) and, parameter by parameter, against the fp32 CPU oracle (cosine / relative
    error).  At B=16 the BatchNorm statistics average over >= 1600 pixels per channel, so the bounds are tighter than
    the B=2 tests' (which have to allow for fp32 chaos through batch statistics of a handful of pixels).
  * configs[4] per-GPU shape: ResNet-101 FPN at 600x600 (four-level pyramid), B=4 against an oracle-made fixture (o1_*).
  * configs[3]: SSD-VGG16 at 300x300, B=2.
  * a 20-step training trajectory at the configs[1] shape and eval-mode arg-max agreement over 256 samples (the proxy for
    the north_star's Acc@IoU0.5 clause: no dataset is available offline).
Tolerances are asserted AND the measured values are printed (pytest -s / -rP shows them)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import zsg_oracle as O  # noqa: E402
from test_gpu_net import RATIOS, SCALES, Z, build, rel_err, to_dev  # noqa: E402,F401


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def test_configs1_b16_vs_reference_golden_and_oracle(Z, gold):
    g = gold("g10_e2e_300_b16")
    cfg, net, sd, lf, ev = build(Z, seed=int(g["seed"][0]))
    net.train()
    bt = O.synthetic_batch(16, 300, 300, seed=int(g["batch_seed"][0]))
    inp = to_dev(bt)
    h0, c0 = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    att, bbx = out["att_out"].detach().cpu().numpy(), out["bbx_out"].detach().cpu().numpy()
    e_att, e_bbx = float(np.abs(att[:, ::37] - g["att_out_s"]).max()), float(np.abs(bbx[:, ::37] - g["bbx_out_s"]).max())
    print(f"forward max abs err vs reference: att {e_att:.2e} bbx {e_bbx:.2e}")
    assert e_att <= 2e-3 and e_bbx <= 2e-3
    np.testing.assert_allclose(np.abs(att).astype(np.float64).sum(), g["att_abs_sum"], rtol=1e-5)
    ls = lf(out, inp)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(ls[k].item(), g[k], rtol=2e-4, err_msg=k)
    ls["loss"].backward()
    em = ev(out, inp)
    assert em["Acc"].item() == g["Acc"] and em["MaxPos"].item() == g["MaxPos"]
    # every gradient norm vs the reference; sampled gradients element-wise
    norms = dict(zip(list(g["grad_names"]), g["grad_norms"]))
    P = dict(net.named_parameters())
    worst = max((abs(float(P[n].grad.double().norm()) - v) / (v + 1e-12), n) for n, v in norms.items())
    print(f"worst gradient-norm deviation vs reference: {worst[0]:.2e} ({worst[1]})")
    assert worst[0] <= 1e-2, worst
    for k in g.files:
        if k.startswith("grad__"):
            got = P[k[6:]].grad.cpu().numpy()
            ref = g[k]
            if got.size > 20000:
                got = got.reshape(-1)[::max(1, got.size // 20000)]
            e = rel_err(torch.from_numpy(np.ascontiguousarray(got)), torch.from_numpy(ref.reshape(got.shape)))
            lim = 1e-3 if k[6:].startswith("att_reg_box") else 5e-2      # (deep-layer bound: see the fp64 yard-stick below)
            assert e <= lim, f"{k}: relative error {e:.3g} > {lim}"
    np.testing.assert_allclose(net.state_dict()["backbone.encoder.bn1.running_mean"].cpu().numpy(), g["rm_bn1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(net.state_dict()["backbone.encoder.layer4.2.bn3.running_var"].cpu().numpy(), g["rv_l4"], rtol=1e-3, atol=1e-6)
    # parameter by parameter against the fp32 CPU oracle and its fp64 twin on the same inputs.  The oracle side — one fp32 and one fp64
    # forward + backward at B=16 on the host, ~40 s — is a committed fixture since round 6 (tests/golden/o2_r50_300_b16.npz, made by
    # tests/golden/make_oracle_fixtures.py from the oracle the reference goldens pin; the loss it reproduces is the golden's, checked here).
    o2 = gold("o2_r50_300_b16")
    np.testing.assert_allclose(float(o2["loss32"][0]), g["loss"], rtol=2e-5, err_msg="the fixture's oracle is not the reference golden's run")
    # arg-max score anchor (evaluator.py:74).  On a randomly initialised network the best anchors are near-ties (the
    # golden records top-2 score gaps of 1e-6..3e-5), so the criterion is agreement UP TO TIES: the anchor the HIP path picks
    # must be, in the oracle's scores, within twice the measured score difference of the oracle's maximum — and identical
    # wherever the oracle's top-2 gap exceeds that bound.
    s_hip = torch.sigmoid(out["att_out"].detach().squeeze(-1)).cpu()
    s_ref = torch.sigmoid(torch.from_numpy(o2["att32"]))
    delta = (s_hip - s_ref).abs().max(1).values
    i_hip, top2 = s_hip.argmax(1), s_ref.topk(2, dim=1)
    rows = torch.arange(16)
    assert bool((s_ref[rows, i_hip] >= top2.values[:, 0] - 2 * delta).all()), "arg-max anchors differ by more than a tie"
    sure = (top2.values[:, 0] - top2.values[:, 1]) > 2 * delta
    assert bool((i_hip[sure] == top2.indices[sure, 0]).all())
    assert np.array_equal(top2.indices[:, 0].numpy()[g["top2_gap"] > 1e-5], g["top1_idx"][g["top2_gap"] > 1e-5])      # oracle == reference
    print(f"arg-max: {int(sure.sum())}/16 samples beyond the tie bound (max score diff {float(delta.max()):.1e}), all consistent")
    # gradients on the fixture's sampled entries (<= 512 per parameter, fixed stride): cosine / relative error against the fp32 oracle,
    # distance from fp64 next to the fp32 oracle's own (the yard-stick for how far two CORRECT fp32 implementations may be apart at
    # this depth: ReLU / max-pool decisions of activations within an ulp of a tie flip, and each flip perturbs everything upstream)
    stats, rows = [], []
    for i, n in enumerate(o2["names"]):
        n = str(n)
        gh = P[n].grad.detach().cpu().double().reshape(-1)
        k = gh.numel()
        idx = np.arange(0, k, max(1, k // 512))[:512]
        g64s, g32s = torch.from_numpy(o2["g64_s"][i][:len(idx)]), torch.from_numpy(o2["g32_s"][i][:len(idx)]).double()
        stats.append((cos(gh[idx], g32s), rel_err(gh[idx], g32s), n))
        sc = (k / len(idx)) ** 0.5                      # sampled -> full tensor (both errors alike)
        n64 = float(o2["norm64"][i]) + 1e-300
        rows.append((float((gh[idx] - g64s).norm()) * sc / n64, max(float(o2["err32"][i]), float((g32s - g64s).norm()) * sc) / n64, n))
    lo_cos, hi_rel = min(stats), max(stats, key=lambda t: t[1])
    head_rel = max(t[1] for t in stats if t[2].startswith("att_reg_box"))
    print(f"vs fp32 oracle (sampled entries): min cosine {lo_cos[0]:.7f} ({lo_cos[2]}), max rel err {hi_rel[1]:.2e} ({hi_rel[2]}), head max rel {head_rel:.2e}")
    # the same yard-stick for the FORWARD bound asserted above (2e-3 against the reference golden): how far is the CPU fp32 oracle
    # (= the reference's own arithmetic) from float64 at this shape, next to the HIP path?
    o64 = torch.from_numpy(o2["out64_s"])
    f_hip = float((out["att_bbx_out"].detach().cpu().double()[:, ::int(o2["out_stride"][0])] - o64).abs().max())
    f_cpu = float(o2["fwd_err_cpu"][0])
    print(f"forward max abs err vs fp64 at B=16: HIP {f_hip:.2e} (every 7th anchor), CPU fp32 oracle {f_cpu:.2e} (all anchors)")
    assert f_hip <= max(4 * f_cpu, 2e-3)
    w_hip, w_cpu = max(rows), max(rows, key=lambda t: t[1])
    print(f"vs fp64: worst HIP rel err {w_hip[0]:.2e} ({w_hip[2]}; CPU fp32 there {w_hip[1]:.2e}); worst CPU fp32 rel err {w_cpu[1]:.2e} ({w_cpu[2]})")
    print("vs fp64, deepest layers (hip, cpu):", [(n.split('encoder.')[-1], f"{a:.1e}", f"{b:.1e}") for a, b, n in rows if n in
          ("backbone.encoder.conv1.weight", "backbone.encoder.layer1.0.conv1.weight", "backbone.encoder.layer3.0.conv2.weight", "att_reg_box.0.0.weight")])
    assert lo_cos[0] >= 0.999, lo_cos
    assert head_rel <= 1e-3, head_rel
    for a, b, n in rows:
        assert a <= max(4 * b, 2e-3), f"{n}: HIP is {a:.2e} from fp64, the CPU fp32 oracle {b:.2e}"


def _fwd_bwd_vs_fp64(Z, arch, B, hw, six, kind="retina", seed=13):
    from test_gpu_net import fp64_twin, grad_tol
    config, evaluator, loss, mdl, optim = Z
    flags = dict(resize_img=[hw, hw])
    if kind == "ssd_vgg":
        flags["mdl_to_use"] = "ssd_vgg"
        cfg = config.get_cfg(**flags)
        net = mdl.get_default_net(9, cfg)
        sd = O.seeded_ssd_state_dict(seed=seed)
        net.load_state_dict(sd)
        net.to("cuda")
        r, s = config.ratios_scales(cfg)
        lf = loss.get_default_loss(r, s, cfg)
    else:
        cfg, net, sd, lf, ev = build(Z, arch=arch, seed=seed, **flags)
    net.train()
    bt = O.synthetic_batch(B, hw, hw, seed=8)
    gq = torch.Generator().manual_seed(4)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch=arch, six_hundred=six)
    assert out["feat_sizes"].tolist() == ref["feat_sizes"].tolist()
    anc = torch.from_numpy(O.create_anchors([tuple(r) for r in ref["feat_sizes"].tolist()], RATIOS, SCALES).astype(np.float32))
    o_gpu = out["att_bbx_out"].detach().cpu()
    o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach()
    sd64, ref64, ls64 = fp64_twin(sd, bt, h0, c0, arch, anc, six_hundred=six)
    o_64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
    err = float((o_gpu.double() - o_64).abs().max())
    err_cpu = float((o_cpu.double() - o_64).abs().max())
    print(f"{arch} {hw}x{hw} B={B}: forward max abs err vs fp64: HIP {err:.2e}, CPU fp32 oracle {err_cpu:.2e}")
    assert err <= max(4 * err_cpu, 2e-3)           # (train-mode BatchNorm on one image through ~100 layers amplifies rounding)
    ls = lf(out, inp)
    lr = O.torch_loss(ref, bt["annot"], anc)
    np.testing.assert_allclose(ls["loss"].item(), ls64["loss"].item(), rtol=max(5e-4, 4 * abs(lr["loss"].item() - ls64["loss"].item()) / abs(ls64["loss"].item())))
    ls["loss"].backward()
    lr["loss"].backward()
    bad = []
    for n, p in net.named_parameters():
        if sd64[n].grad is None:
            continue
        g64 = sd64[n].grad
        ec = float((sd[n].grad.double() - g64).norm())
        eg = float((p.grad.cpu().double() - g64).norm())
        if eg > grad_tol(ec, g64):
            bad.append((n, eg, ec, float(g64.norm())))
    assert not bad, f"{len(bad)} gradients further from fp64 than allowed: {bad[:6]}"


def test_configs4_resnet101_600_b4_vs_oracle_fixture(Z, gold):
    """flickr30k_c1's per-GPU shape: ResNet-101 + FPN at 600x600 (four levels, fpn_resnet.py:173-174).  The reference cannot construct
    ResNet-101 (mdl.py:411 hard-codes resnet50), so there is no reference golden for this depth; every block type it is made of is pinned
    by reference goldens (g8_bottleneck, g8_fpn600), the live HIP-vs-fp64 comparison of this depth runs at 128^2
    (test_gpu_net.py::test_forward_backward_vs_oracle[resnet101-1-128]) and the four-level pyramid at 600^2 in
    test_gpu_net.py::test_600x600_four_level_pyramid.  Here, B=4: outputs, loss and EVERY parameter gradient against the oracle's fp64 twin (VERDICT r05 item 5:
    the largest batch whose CPU oracle is affordable — B=32 is not).  The oracle side — one fp32 and one fp64 forward + backward on the
    host, minutes — is a committed fixture (tests/golden/o1_r101_600_b4.npz, made by tests/golden/make_oracle_fixtures.py from the
    oracle whose blocks the reference goldens pin); the criterion is _fwd_bwd_vs_fp64's: the HIP path must be as close to fp64 as the
    CPU fp32 oracle is (forward: max(4 x, 2e-3); per parameter: test_gpu_net.grad_tol), measured on the fixture's sampled gradient
    entries (<= 512 per parameter, fixed stride) and scaled to the full tensor by the CPU oracle's own sampled / full ratio."""
    from test_gpu_net import grad_tol
    g = gold("o1_r101_600_b4")
    B, hw = int(g["B"][0]), int(g["hw"][0])
    cfg, net, sd, lf, ev = build(Z, arch="resnet101", seed=int(g["seed"][0]), resize_img=[hw, hw])
    net.train()
    bt = O.synthetic_batch(B, hw, hw, seed=int(g["batch_seed"][0]))
    gq = torch.Generator().manual_seed(int(g["hc_seed"][0]))
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    out = net(inp)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    o_gpu = out["att_bbx_out"].detach().cpu().double()[:, ::int(g["out_stride"][0])]
    err, err_cpu = float((o_gpu - torch.from_numpy(g["out64_s"])).abs().max()), float(g["fwd_err_cpu"][0])
    print(f"resnet101 {hw}x{hw} B={B}: forward max abs err vs fp64 (sampled anchors): HIP {err:.2e}, CPU fp32 oracle (all anchors) {err_cpu:.2e}")
    assert err <= max(4 * err_cpu, 2e-3)
    ls = lf(out, inp)
    l64, l32 = float(g["loss64"][0]), float(g["loss32"][0])
    np.testing.assert_allclose(ls["loss"].item(), l64, rtol=max(5e-4, 4 * abs(l32 - l64) / abs(l64)))
    ls["loss"].backward()
    P = dict(net.named_parameters())
    bad, worst = [], (0.0, "")
    for i, n in enumerate(g["names"]):
        n = str(n)
        gh = P[n].grad.detach().cpu().double().reshape(-1)
        k = gh.numel()
        idx = np.arange(0, k, max(1, k // 512))[:512]
        g64s, g32s = torch.from_numpy(g["g64_s"][i][:len(idx)]), torch.from_numpy(g["g32_s"][i][:len(idx)]).double()
        e_hip_s, e_cpu_s = float((gh[idx] - g64s).norm()), float((g32s - g64s).norm())
        n64, e_cpu = float(g["norm64"][i]), float(g["err32"][i])
        # sampled -> full tensor: the sample covers len(idx) of k entries; both errors are scaled alike (sqrt(k / len(idx)))
        scale = (k / len(idx)) ** 0.5
        eg, ec = e_hip_s * scale, max(e_cpu, e_cpu_s * scale)
        tol = grad_tol(ec, torch.tensor([n64]))
        worst = max(worst, (eg / (n64 + 1e-300), n))
        if eg > tol:
            bad.append((n, eg, ec, n64))
    print(f"resnet101 {hw}x{hw} B={B}: worst HIP gradient distance from fp64 (relative, sampled): {worst[0]:.2e} ({worst[1]}); {len(g['names'])} parameters")
    assert not bad, f"{len(bad)} gradients further from fp64 than allowed: {bad[:6]}"


def test_configs3_ssd_vgg_b2(Z):
    """SSD-VGG16 trunk (ssd_vgg.py path) at 300x300, B=2"""
    _fwd_bwd_vs_fp64(Z, "ssd_vgg", 2, 300, False, kind="ssd_vgg", seed=5)


def _full_batch_properties(Z, arch, B, hw, kind, seed, oracle_grads=False):
    """Size-independent checks at a configuration's OWN per-GPU batch: finite outputs; loss and evaluator EXACT functions of the
    HIP outputs (the numpy oracle on what the network produced); the step repeatable to fp32 summation order (bit-identical runs in
    deterministic mode are covered by test_gpu_determinism.py); the first SUB images agree with the same images run as a batch of
    SUB through eval-mode (BatchNorm folded, batch-independent) plans.
    oracle_grads=True (the SSD-VGG16 configuration: no BatchNorm, so the full batch is a sum of per-sample terms and the CPU fp32
    oracle follows in about a minute): outputs, loss and EVERY parameter gradient of the B-sample step against the CPU oracle's.
    What stays UNCHECKED for ResNet-101 at 600^2, B=32: its gradients at that batch are compared with nothing but themselves (finite,
    repeatable) — train-mode batch statistics make a sub-batch step a different function, the eval plans of this package are
    forward-only, and the CPU oracle needs ~10 minutes for one such step in fp32 + fp64; the same network's gradients are checked
    against the fp64 twin at B=1 and B=2 (test_configs4_resnet101_600)."""
    config, evaluator, loss, mdl, optim = Z
    flags = dict(resize_img=[hw, hw], bs=B)
    if kind == "ssd_vgg":
        cfg = config.get_cfg(mdl_to_use="ssd_vgg", **flags)
        net = mdl.get_default_net(9, cfg)
        sd = O.seeded_ssd_state_dict(seed=seed)
        net.load_state_dict(sd)
        net.to("cuda")
        r, s = config.ratios_scales(cfg)
        lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
    else:
        cfg, net, sd, lf, ev = build(Z, arch=arch, seed=seed, **flags)
    bt = O.synthetic_batch(B, hw, hw, seed=21)
    gq = torch.Generator().manual_seed(6)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    inp = to_dev(bt)
    inp["h0"], inp["c0"] = h0, c0
    net.train()
    out = net(inp)
    ab = out["att_bbx_out"].detach()
    assert bool(torch.isfinite(ab).all()), "non-finite network output"
    ls = lf(out, inp)
    em = ev(out, inp)
    ls["loss"].backward()
    gflat = net.store.grad.clone()
    assert bool(torch.isfinite(gflat).all()) and float(gflat.abs().max()) > 0, "non-finite / empty gradient"
    fs = [tuple(r) for r in out["feat_sizes"].tolist()]
    anc = O.create_anchors(fs, RATIOS, SCALES).astype(np.float32)
    att, bbx = out["att_out"].detach().squeeze(-1).cpu().numpy(), out["bbx_out"].detach().cpu().numpy()
    lo = O.zsg_loss(att, bbx, bt["annot"].numpy(), anc)
    np.testing.assert_allclose(ls["loss"].item(), lo["loss"], rtol=2e-5, err_msg="loss of the HIP outputs vs the numpy oracle on the same outputs")
    eo = O.zsg_eval(att, bbx, bt["annot"].numpy(), bt["img_size"].numpy(), anc)
    # (MaxPos rests on the exact IoU arg-max; Acc on the arg-max SCORE, where a random-init network has near-ties: one sample of slack)
    assert float(em["MaxPos"]) == float(eo["MaxPos"]) and abs(float(em["Acc"]) - float(eo["Acc"])) <= 1.0 / B + 1e-9, "evaluator on the HIP outputs vs the numpy oracle"
    # repeatability: the same step again (fresh gradient buffer) — split-K atomics are the only non-deterministic sums
    for p_ in net.parameters():
        p_.grad = None
    out2 = net(inp)
    lf(out2, inp)["loss"].backward()
    d_out = float((out2["att_bbx_out"].detach() - ab).abs().max())
    d_g = float((net.store.grad - gflat).norm() / gflat.norm())
    print(f"{kind}/{arch} {hw}x{hw} B={B}: loss {ls['loss'].item():.5f} (oracle on the same outputs {lo['loss']:.5f}), Acc {float(em['Acc']):.3f}; "
          f"re-run: outputs differ by {d_out:.1e}, gradient by {d_g:.1e} (relative)")
    assert d_out <= 1e-4 and d_g <= 1e-4
    if oracle_grads:
        # the whole B-sample step against the CPU fp32 oracle (reference ssd_vgg.py:54-102 + mdl.py:338-403 + loss.py:43-143 + autograd)
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_()
        ref = O.zsgnet_forward(sd, bt, h0, c0, arch=arch)
        lr = O.torch_loss(ref, bt["annot"], torch.from_numpy(anc))
        lr["loss"].backward()
        o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach()
        e_out = float((ab.cpu() - o_cpu).abs().max())
        np.testing.assert_allclose(ls["loss"].item(), lr["loss"].item(), rtol=2e-4)
        worst_n, worst_r, n_par = 0.0, 0.0, 0
        for n, _p in net.named_parameters():
            gr = sd[n].grad
            if gr is None:                               # (a parameter the SSD path does not use)
                continue
            gh = net.store.view(n, gflat).cpu()          # (the FIRST step's gradients: the repeat above overwrote the live buffer)
            nr = float(gr.norm()) + 1e-30
            worst_n = max(worst_n, abs(float(gh.norm()) - nr) / nr)
            worst_r = max(worst_r, rel_err(gh, gr))
            n_par += 1
        print(f"{kind} B={B} vs the CPU fp32 oracle: forward max abs diff {e_out:.2e}; {n_par} gradients: worst norm deviation {worst_n:.2e}, worst relative error {worst_r:.2e}")
        assert e_out <= 2e-3 and n_par > 0
        assert worst_n <= 1e-2 and worst_r <= 5e-3
    # eval mode is batch-independent: the first SUB images alone must reproduce their rows of the full batch
    # (zero LSTM start states: the reference hands h0 / c0 to the queries by their position in the batch's length ORDER
    #  (mdl.py:296-330), so a non-zero state ties a sample's output to the rest of the batch)
    SUB = 2
    net.eval()
    with torch.no_grad():
        inp["h0"], inp["c0"] = torch.zeros(2, B, 128), torch.zeros(2, B, 128)
        full = net(inp)["att_bbx_out"].clone()
        sub = {k: (v[:SUB] if torch.is_tensor(v) and v.shape[:1] == (B,) else v) for k, v in inp.items()}
        sub["h0"], sub["c0"] = torch.zeros(2, SUB, 128), torch.zeros(2, SUB, 128)
        part = net(sub)["att_bbx_out"]
    mag = float(full[:SUB].abs().max())
    e_sub = float((full[:SUB] - part).abs().max()) / max(mag, 1e-30)
    # (relative: with the running statistics of two training steps the eval-mode activations of a random-init network are not O(1);
    #  the two plans use different tiles / split-K factors = other fp32 summation orders)
    print(f"eval: rows 0..{SUB - 1} of the B={B} batch vs the same images as a batch of {SUB}: max diff {e_sub:.1e} of the output range {mag:.3g}")
    assert e_sub <= 1e-3
    return net


def test_configs4_resnet101_600_b32_properties(Z):
    """flickr30k_c1's per-GPU shape at ITS batch: ResNet-101 + FPN, 600x600, B=32 (what the builder benches)"""
    _full_batch_properties(Z, "resnet101", 32, 600, "retina", seed=13)


def test_configs3_ssd_vgg_b32_properties(Z):
    """configs[3] at its batch: SSD-VGG16 backbone, 300x300, B=32 (ssd_vgg.py:54-102)"""
    _full_batch_properties(Z, "ssd_vgg", 32, 300, "ssd_vgg", seed=5, oracle_grads=True)


def test_training_trajectory_and_eval_argmax_agreement(Z, gold):
    """Acc@IoU0.5 proxy (no dataset is available offline).  (1) 12 optimisation steps at the configs[1] shape (ResNet-50 FPN,
    300x300, B=16; a fresh synthetic batch and fresh LSTM states every step, Adam lr 1e-4 as main_dist.py:50) against the
    REFERENCE's own 12 steps from the same start on the same stream of batches (golden g16_trajectory: mdl.py / loss.py +
    torch.optim.Adam on the CPU, tests/golden/make_golden.py gen_trajectory; until round 5 the CPU oracle was stepped beside the
    HIP model inside this test — 100 s of the GPU suite).  Adam's first steps are ~lr*sign(g), so rounding-level
    gradient differences on near-zero elements flip their update and the two fp32 trajectories drift apart chaotically
    (any two fp32 implementations do; measured: up to ~10 % on single steps while the loss falls 8x): asserted are the
    same starting loss (2e-4), every step within 15 %, the mean of the last six steps within 5 %, and the same overall
    decrease; BatchNorm running statistics within 1 %.  (2) the trained weights, loaded into the CPU oracle, in eval mode on 128
    fresh samples: the arg-max-score anchor (evaluator.py:74) must equal the oracle's wherever the oracle's
    top-2 score gap exceeds 1e-3, and the Acc@IoU0.5 hit counts must agree to within the samples below that gap."""
    config, evaluator, loss, mdl, optim = Z
    g = gold("g16_trajectory")
    cfg, net, sd, lf, ev = build(Z, seed=int(g["seed"][0]))
    net.train()
    opt = optim.FusedAdam(net, lr=float(g["lr"][0]), betas=(0.9, 0.99))
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(300, 300), RATIOS, SCALES).astype(np.float32))
    gq = torch.Generator().manual_seed(int(g["hc_seed"][0]))
    ref_losses = [float(v) for v in g["losses"]]
    curve = []
    for it in range(len(ref_losses)):
        bt = O.synthetic_batch(16, 300, 300, seed=int(g["batch_seed0"][0]) + it)
        h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
        inp = to_dev(bt)
        inp["h0"], inp["c0"] = h0, c0
        opt.zero_grad()
        out = net(inp)
        ls = lf(out, inp)
        ls["loss"].mean().backward()
        opt.step()
        curve.append((ls["loss"].item(), ref_losses[it]))
    dev_ = max(abs(a - b) / abs(b) for a, b in curve)
    tail_h, tail_o = np.mean([a for a, _ in curve[-6:]]), np.mean([b for _, b in curve[-6:]])
    print("loss curve (hip, reference):", [(round(a, 3), round(b, 3)) for a, b in curve], f"max rel deviation {dev_:.2e}; last-6 mean {tail_h:.3f} vs {tail_o:.3f}")
    np.testing.assert_allclose(curve[0][0], curve[0][1], rtol=2e-4)
    assert dev_ <= 0.15, curve
    assert abs(tail_h - tail_o) <= 0.05 * tail_o
    assert curve[-1][0] < 0.5 * curve[0][0] and curve[-1][1] < 0.5 * curve[0][1]
    got = net.state_dict()
    for k, gk in (("backbone.encoder.bn1.running_mean", "rm_bn1"), ("backbone.encoder.layer2.3.bn3.running_var", "rv_l2"),
                  ("backbone.encoder.layer4.2.bn3.running_mean", "rm_l4")):
        e = rel_err(got[k].cpu(), torch.from_numpy(g[gk]))
        print(f"{k}: rel diff after 12 steps {e:.2e}")
        assert e <= 1e-2, f"{k}: running statistic differs by {e:.3g} after 12 steps"
    # (2) eval-mode arg-max agreement on 128 fresh samples at the TRAINED weights, loaded into the CPU oracle (the eval path — BatchNorm
    # folded into the convolutions, evaluator kernel — is what is compared here, at weights whose scores are no longer near-ties)
    sd_ref = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}      # the trained weights, for the CPU oracle
    net.eval()
    n_sure = n_agree = 0
    acc_hip = acc_ref = 0.0
    with torch.no_grad():
        for bi in range(8):
            bt = O.synthetic_batch(16, 300, 300, seed=900 + bi)
            h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
            inp = to_dev(bt)
            inp["h0"], inp["c0"] = h0, c0
            out = net(inp)
            em = ev(out, inp)
            ref = O.zsgnet_forward(sd_ref, bt, h0, c0, arch="resnet50", training=False)
            sc = torch.sigmoid(ref["att_out"].squeeze(-1))
            top2 = sc.topk(2, dim=1)
            sure = (top2.values[:, 0] - top2.values[:, 1]) > 1e-3
            idx_hip = torch.sigmoid(out["att_out"].squeeze(-1)).argmax(1).cpu()
            n_sure += int(sure.sum())
            n_agree += int((idx_hip[sure] == top2.indices[sure, 0]).sum())
            rv = O.zsg_eval(ref["att_out"].squeeze(-1).numpy(), ref["bbx_out"].numpy(), bt["annot"].numpy(), bt["img_size"].numpy(), anc.numpy())
            acc_hip += float(em["Acc"]) * 16
            acc_ref += float(rv["Acc"]) * 16
    print(f"eval: {n_agree}/{n_sure} arg-max anchors agree (of 128 samples, {128 - n_sure} below the 1e-3 score gap); "
          f"Acc@0.5 hits hip {acc_hip:.0f} vs oracle {acc_ref:.0f}")
    assert n_sure >= 64
    assert n_agree == n_sure, (n_agree, n_sure)
    assert abs(acc_hip - acc_ref) <= (128 - n_sure)


def test_learnable_task_reaches_the_same_accuracy(Z, gold):
    """Acc@IoU0.5 proxy, part 2 (reference: evaluator.py:48-117 scoring the training of utils.py:353-414).
    A task the network can actually learn — O.learnable_batch: the box is a bright rectangle in the image.  Golden g15 holds what
    the REFERENCE made of it (mdl.py / loss.py / evaluator.py on the CPU, torch.optim.Adam as main_dist.py:50; tests/golden/
    make_golden.py gen_learnable): ResNet-50 + FPN, 128x128, batch 16, 320 steps at lr 5e-4 (the last 60 at 5e-5) from a seeded
    start, a fresh batch and fresh LSTM states every step — every step's loss, then Acc@IoU0.5 on 256 held-out samples.  The HIP
    model trains on the same stream.  Two fp32 trajectories drift apart step by step (see the trajectory test above); what must
    agree is what the two LEARN: the same first loss, both smoothed losses below 8 % of it at the end and within 25 % of each
    other, and Acc@IoU0.5 >= 0.98 on both sides.
    The recipe (round 6): until round 5 the run was 240 steps at lr 1e-3.  At that rate Adam's trajectory on this task is unstable
    in fp32 — of 48 HIP runs in fresh processes (each tunes its tiles anew: another summation order) about one in six shows a loss
    spike of 5-70x between steps 45 and 140 and ends at 0.36-1.1 / 4-251 hits instead of 0.21-0.27 / 254-256 (profiles/
    r06_learnable_stability.txt).  It is the recipe, not a kernel: in deterministic mode with a shared tuning cache ten runs of 150
    steps are bit-identical, every tuner candidate of every launch shape of this network agrees with its siblings (ZSG_TUNE_VERIFY:
    2 629 comparisons), 440 000 stream-K launches under a noisy neighbour reproduce their first result (tools/sk_stress.py), and at
    lr 5e-4 twenty of twenty runs converge (end loss 0.17-0.24, 246-256 hits at 240 steps, 254-256 at 320).  (Rounds 2-3 trained
    the CPU oracle beside the HIP model inside the test: 8 minutes of the suite; measured then: HIP 253 / 256, oracle 254 / 256.)"""
    config, evaluator, loss, mdl, optim = Z
    g = gold("g15_learnable")
    S, B, steps, lr_ = int(g["S"][0]), int(g["B"][0]), int(g["steps"][0]), float(g["lr"][0])
    ref_losses, ref_hits = g["losses"], float(g["hits"].sum())
    cfg = config.get_cfg(resnet_arch="resnet50", resize_img=[S, S])
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    net = mdl.get_default_net(9, cfg)
    net.load_state_dict(sd)
    net.to("cuda").train()
    lf, ev = loss.get_default_loss(RATIOS, SCALES, cfg), evaluator.get_default_eval(RATIOS, SCALES, cfg)
    opt = optim.FusedAdam(net, lr=lr_, betas=(0.9, 0.99))
    gq = torch.Generator().manual_seed(8)
    hip_losses = []
    for it in range(steps):
        if it == int(g["decay_at"][0]):
            for grp in opt.param_groups:
                grp["lr"] = lr_ * 0.1
        bt = O.learnable_batch(B, S, seed=100 + it)
        h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
        inp = to_dev(bt)
        inp["h0"], inp["c0"] = h0, c0
        opt.zero_grad()
        out = net(inp)
        if it == 0:
            assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
        ls = lf(out, inp)
        ls["loss"].mean().backward()
        opt.step()
        hip_losses.append(float(ls["loss"].detach()))
    first = (hip_losses[0], float(ref_losses[0]))
    last = None
    for cur in zip(hip_losses, ref_losses):
        last = cur if last is None else (0.8 * last[0] + 0.2 * cur[0], 0.8 * last[1] + 0.2 * cur[1])
    print(f"loss: first step hip {first[0]:.3f} / reference {first[1]:.3f}; smoothed end hip {last[0]:.3f} / reference {last[1]:.3f}")
    np.testing.assert_allclose(first[0], first[1], rtol=5e-4)
    assert last[0] < 0.08 * first[0] and last[1] < 0.08 * first[1], (first, last)
    net.eval()
    hits_h = 0.0
    with torch.no_grad():
        for bi in range(16):
            bt = O.learnable_batch(16, S, seed=9000 + bi)
            h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
            inp = to_dev(bt)
            inp["h0"], inp["c0"] = h0, c0
            hits_h += float(ev(net(inp), inp)["Acc"]) * 16
    print(f"eval Acc@IoU0.5 on 256 held-out samples: hip {hits_h:.0f}/256, reference {ref_hits:.0f}/256")
    assert abs(last[0] - last[1]) <= 0.25 * last[1], last
    assert hits_h >= 0.98 * 256 and ref_hits >= 0.98 * 256, (hits_h, ref_hits)


def test_configs0_resnet18_fpn_300_b2(Z):
    """BASELINE configs[0]'s own workload on the HIP path (ResNet-18 FPN, bs=2, 20-token queries) at the 300x300 the config implies —
    rounds 1-4 ran ResNet-18 at 96^2 / 600^2 only: forward, loss and every gradient against the CPU oracle and its
    fp64 twin (fpn_resnet.py BasicBlock path, mdl.py:338-403)."""
    _fwd_bwd_vs_fp64(Z, "resnet18", 2, 300, False, seed=3)
