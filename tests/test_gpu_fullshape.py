"""Parity at the shapes the benchmark and BASELINE.json's configs actually run (not toy shapes):
  * configs[1]: ResNet-50 FPN, 300x300, B=16 — against the REFERENCE golden g10_e2e_300_b16 (outputs, loss, evaluator,
    every gradient norm, sampled gradients) and, parameter by parameter, against the fp32 CPU oracle (cosine / relative
    error).  At B=16 the BatchNorm statistics average over >= 1600 pixels per channel, so the bounds are tighter than
    the B=2 tests' (which have to allow for fp32 chaos through batch statistics of a handful of pixels).
  * configs[4] per-GPU shape: ResNet-101 FPN at 600x600 (four-level pyramid), B=1.
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
    # arg-max score anchor: exact wherever the reference's top-2 score gap exceeds 1e-5 (evaluator.py:74)
    sc = torch.sigmoid(out["att_out"].detach().squeeze(-1))
    sure = g["top2_gap"] > 1e-5
    assert sure.sum() >= 14
    assert np.array_equal(sc.argmax(1).cpu().numpy()[sure], g["top1_idx"][sure])
    np.testing.assert_allclose(em["pred_boxes"].cpu().numpy()[sure], g["pred_boxes"][sure], rtol=1e-4, atol=2e-2)
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
            lim = 1e-3 if k[6:].startswith("att_reg_box") else 1e-2
            assert e <= lim, f"{k}: relative error {e:.3g} > {lim}"
    np.testing.assert_allclose(net.state_dict()["backbone.encoder.bn1.running_mean"].cpu().numpy(), g["rm_bn1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(net.state_dict()["backbone.encoder.layer4.2.bn3.running_var"].cpu().numpy(), g["rv_l4"], rtol=1e-3, atol=1e-6)
    # parameter by parameter against the fp32 CPU oracle on the same inputs
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet50")
    anc = torch.from_numpy(O.create_anchors([tuple(r) for r in ref["feat_sizes"].tolist()], RATIOS, SCALES).astype(np.float32))
    O.torch_loss(ref, bt["annot"], anc)["loss"].backward()
    stats = []
    for n, p in net.named_parameters():
        r = sd[n].grad
        stats.append((cos(p.grad.cpu(), r), rel_err(p.grad.cpu(), r), n))
    lo_cos, hi_rel = min(stats), max(stats, key=lambda t: t[1])
    head_rel = max(t[1] for t in stats if t[2].startswith("att_reg_box"))
    print(f"vs fp32 oracle: min cosine {lo_cos[0]:.7f} ({lo_cos[2]}), max rel err {hi_rel[1]:.2e} ({hi_rel[2]}), head max rel {head_rel:.2e}")
    assert lo_cos[0] >= 0.9999, lo_cos
    assert hi_rel[1] <= 1e-2, hi_rel
    assert head_rel <= 1e-3, head_rel


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
    err = float((o_gpu - o_cpu).abs().max())
    print(f"{arch} {hw}x{hw} B={B}: forward max abs err vs fp32 oracle {err:.2e}")
    assert err <= 5e-3
    ls = lf(out, inp)
    lr = O.torch_loss(ref, bt["annot"], anc)
    np.testing.assert_allclose(ls["loss"].item(), lr["loss"].item(), rtol=5e-4)
    ls["loss"].backward()
    lr["loss"].backward()
    sd64, ref64, ls64 = fp64_twin(sd, bt, h0, c0, arch, anc, six_hundred=six)
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


def test_configs4_resnet101_600(Z):
    """flickr30k_c1's per-GPU shape: ResNet-101 + FPN at 600x600 (four levels, fpn_resnet.py:173-174), B=1"""
    _fwd_bwd_vs_fp64(Z, "resnet101", 1, 600, True)


def test_configs3_ssd_vgg_b2(Z):
    """SSD-VGG16 trunk (ssd_vgg.py path) at 300x300, B=2"""
    _fwd_bwd_vs_fp64(Z, "ssd_vgg", 2, 300, False, kind="ssd_vgg", seed=5)


def test_training_trajectory_and_eval_argmax_agreement(Z):
    """Acc@IoU0.5 proxy.  (1) 20 optimisation steps at the configs[1] shape (ResNet-50 FPN, 300x300, B=16; a fresh
    synthetic batch and fresh LSTM states every step, Adam lr 1e-4 as main_dist.py:50) next to the CPU oracle stepping
    torch.optim.Adam from the same start: the loss curves must stay within 1 %, the BatchNorm running statistics within
    1e-3.  (2) the trained weights in eval mode on 256 fresh samples: the arg-max-score anchor (evaluator.py:74) must equal
    the oracle's wherever the oracle's top-2 score gap exceeds 1e-4, and the Acc@IoU0.5 counts must agree to within the
    samples below that gap."""
    config, evaluator, loss, mdl, optim = Z
    cfg, net, sd, lf, ev = build(Z, seed=17)
    net.train()
    opt = optim.FusedAdam(net, lr=1e-4, betas=(0.9, 0.99))
    params = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    opt_ref = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.99))
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(300, 300), RATIOS, SCALES).astype(np.float32))
    gq = torch.Generator().manual_seed(8)
    curve = []
    for it in range(20):
        bt = O.synthetic_batch(16, 300, 300, seed=500 + it)
        h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
        inp = to_dev(bt)
        inp["h0"], inp["c0"] = h0, c0
        opt.zero_grad()
        out = net(inp)
        ls = lf(out, inp)
        ls["loss"].mean().backward()
        opt.step()
        lr, _ = O.cpu_train_step(params, buffers, opt_ref, bt, h0, c0, anc, arch="resnet50")
        curve.append((ls["loss"].item(), lr["loss"].item()))
    dev_ = max(abs(a - b) / abs(b) for a, b in curve)
    print("loss curve (hip, oracle):", [(round(a, 4), round(b, 4)) for a, b in curve[::4]], f"max rel deviation {dev_:.2e}")
    assert dev_ <= 1e-2, curve
    got = net.state_dict()
    for k in ("backbone.encoder.bn1.running_mean", "backbone.encoder.layer2.3.bn3.running_var", "backbone.encoder.layer4.2.bn3.running_mean"):
        e = rel_err(got[k].cpu(), buffers[k])
        assert e <= 1e-3, f"{k}: running statistic differs by {e:.3g} after 20 steps"
    # (2) eval-mode arg-max agreement on 256 fresh samples, both models carrying THEIR OWN trained weights
    net.eval()
    sd_ref = {k: v.detach() for k, v in params.items()}
    sd_ref.update(buffers)
    n_sure = n_agree = 0
    acc_hip = acc_ref = 0.0
    with torch.no_grad():
        for bi in range(16):
            bt = O.synthetic_batch(16, 300, 300, seed=900 + bi)
            h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
            inp = to_dev(bt)
            inp["h0"], inp["c0"] = h0, c0
            out = net(inp)
            em = ev(out, inp)
            ref = O.zsgnet_forward(sd_ref, bt, h0, c0, arch="resnet50", training=False)
            sc = torch.sigmoid(ref["att_out"].squeeze(-1))
            top2 = sc.topk(2, dim=1)
            sure = (top2.values[:, 0] - top2.values[:, 1]) > 1e-4
            idx_hip = torch.sigmoid(out["att_out"].squeeze(-1)).argmax(1).cpu()
            n_sure += int(sure.sum())
            n_agree += int((idx_hip[sure] == top2.indices[sure, 0]).sum())
            rv = O.zsg_eval(ref["att_out"].squeeze(-1).numpy(), ref["bbx_out"].numpy(), bt["annot"].numpy(), bt["img_size"].numpy(), anc.numpy())
            acc_hip += float(em["Acc"]) * 16
            acc_ref += float(rv["Acc"]) * 16
    print(f"eval: {n_agree}/{n_sure} arg-max anchors agree (of 256 samples, {256 - n_sure} below the 1e-4 score gap); "
          f"Acc@0.5 hits hip {acc_hip:.0f} vs oracle {acc_ref:.0f}")
    assert n_sure >= 200
    assert n_agree >= n_sure - 2, (n_agree, n_sure)     # weights differ by 20 steps of fp32 rounding: allow 2 near-ties beyond the gap filter
    assert abs(acc_hip - acc_ref) <= (256 - n_sure) + 2
