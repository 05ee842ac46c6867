"""The oracle (oracle/zsg_oracle.py) pinned against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import zsg_oracle as O

RATIOS, SCALES = O.default_ratios_scales()
FS300 = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]
FS600 = [(38, 38), (19, 19), (10, 10), (5, 5)]


def test_grid_bit_exact(gold):
    g = gold("g1_grid")
    for k in g.files:
        _, h, w = k.split("_")
        assert np.array_equal(O.create_grid(int(h), int(w)), g[k]), k


def test_grid_known_answers():
    g = O.create_grid(4, 1)
    assert g[0, 0] == np.float32(-0.75) and g[-1, 0] == np.float32(0.75) and np.all(g[:, 1] == 0)


def test_anchors_bit_exact(gold):
    g = gold("g2_anchors")
    a300 = O.create_anchors(FS300, RATIOS, SCALES)
    a600 = O.create_anchors(FS600, RATIOS, SCALES)
    assert a300.shape == (17460, 4) and a600.shape == (17370, 4)
    assert hashlib.sha256(a300.tobytes()).digest() == g["a300_sha256"].tobytes()
    assert hashlib.sha256(a600.tobytes()).digest() == g["a600_sha256"].tobytes()
    assert np.array_equal(a300.astype(np.float32), g["a300_f32"])
    assert np.array_equal(a300[g["sample_ids"]], g["a300_f64_rows"])
    # anchors reach far outside [-1,1] on the 1x1 level
    assert abs(a300[-1]).max() > 8.9


def test_feat_sizes():
    assert O.feat_sizes_for(300, 300) == FS300
    assert O.feat_sizes_for(600, 600, True) == [(38, 38), (19, 19), (10, 10), (5, 5)]


def test_iou_argmax_mask_exact(gold):
    g = gold("g3_iou")
    anc = gold("g2_anchors")["a300_f32"]
    iou = O.iou_values(g["boxes"], anc)
    assert np.array_equal(iou.max(1), g["maxval"])
    mask, best = O.match_mask(iou, 0.6)
    assert np.array_equal(best.astype(np.int32), g["argmax"])
    rows, cols = np.nonzero(iou > np.float32(0.6))
    assert np.array_equal(rows.astype(np.int32), g["pos_rows"]) and np.array_equal(cols.astype(np.int32), g["pos_cols"])
    assert np.array_equal(iou[g["sample_rows"]][:, ::7], g["sample_iou"])


def test_iou_known_answers():
    a = np.array([[-0.5, -0.5, 0.5, 0.5]], np.float32)
    assert O.iou_values(a, a)[0, 0] == np.float32(1.0) / (np.float32(1.0) + np.float32(1e-8))
    assert O.iou_values(a, np.array([[0.6, 0.6, 0.9, 0.9]], np.float32))[0, 0] == 0
    inner = np.array([[-0.25, -0.25, 0.25, 0.25]], np.float32)
    assert abs(O.iou_values(inner, a)[0, 0] - 0.25) < 1e-6


def test_codec(gold):
    g = gold("g4_codec")
    enc = O.bbox_to_reg_params(g["anchors"], g["boxes"])
    fin = np.isfinite(g["enc"])
    assert np.array_equal(np.isfinite(enc), fin)
    np.testing.assert_allclose(enc[fin], g["enc"][fin], rtol=2e-6, atol=1e-6)
    dec = O.reg_params_to_bbox(g["anchors"], g["regs"])
    np.testing.assert_allclose(dec, g["dec"], rtol=2e-6, atol=1e-6)
    z = O.bbox_to_reg_params(g["anchors"][:5], g["anchors"][:5])
    assert np.allclose(np.stack([z[i, i] for i in range(5)]), 0, atol=1e-6)


@pytest.mark.parametrize("tag,flags", [("b1", {}), ("b2", {}), ("b16", {}), ("nomulti", dict(use_multi=False)),
                                       ("nofocal", dict(use_focal=False)),
                                       ("softmax", dict(use_multi=False, use_softmax=True)), ("nan", {})])
def test_loss_and_eval_small(gold, tag, flags):
    g = gold("g5_loss_eval_small")
    anc = g["anchors"]
    att, bbx, annot = g[f"{tag}_att"][..., 0], g[f"{tag}_bbx"], g[f"{tag}_annot"]
    r = O.zsg_loss(att, bbx, annot, anc, **flags)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(r[k], g[f"{tag}_{k}"], rtol=1e-5, err_msg=k)
    if tag == "nan":
        assert r["nan"] and r["loss"] == pytest.approx(1.01)
        assert f"{tag}_g_att" not in g.files       # reference returns fresh leaves: no grad reaches the logits
    else:
        np.testing.assert_allclose(r["g_att"], g[f"{tag}_g_att"][..., 0], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(r["g_reg"], g[f"{tag}_g_bbx"], rtol=2e-5, atol=1e-9)
    e = O.zsg_eval(att, bbx, annot, g[f"{tag}_img_size"], anc)
    assert np.array_equal(e["pred_ids"], g[f"{tag}_pred_ids"])
    assert e["Acc"] == g[f"{tag}_Acc"] and e["MaxPos"] == g[f"{tag}_MaxPos"]
    np.testing.assert_allclose(e["pred_boxes"], g[f"{tag}_pred_boxes"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(e["pred_scores"], g[f"{tag}_pred_scores"], rtol=1e-6)


def test_loss_and_eval_full(gold):
    g = gold("g5_loss_eval_full")
    anc = gold("g2_anchors")["a300_f32"]
    gen = torch.Generator().manual_seed(int(g["gen_seed"][0]))
    att = (torch.randn(2, 17460, 1, generator=gen) * 1.5 - 3.0).numpy()[..., 0]
    bbx = (torch.randn(2, 17460, 4, generator=gen) * 0.6).numpy()
    r = O.zsg_loss(att, bbx, g["annot"], anc)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(r[k], g[k], rtol=1e-5)
    np.testing.assert_allclose(r["g_att"][:, ::13], g["g_att_s"][..., 0], rtol=2e-5, atol=1e-10)
    np.testing.assert_allclose(r["g_reg"][:, ::13], g["g_bbx_s"], rtol=2e-5, atol=1e-10)
    np.testing.assert_allclose(np.abs(r["g_att"]).astype(np.float64).sum(), g["g_att_abs_sum"], rtol=1e-5)
    e = O.zsg_eval(att, bbx, g["annot"], g["img_size"], anc)
    assert e["Acc"] == g["Acc"] and e["MaxPos"] == g["MaxPos"]
    np.testing.assert_allclose(e["pred_boxes"], g["pred_boxes"], rtol=1e-5, atol=1e-3)


def test_lstm(gold):
    g = gold("g7_lstm")
    sd = {k: v.clone().requires_grad_(k.startswith("lstm.")) for k, v in O.seeded_state_dict("resnet50", int(g["seed"][0])).items()
          if k.startswith("lstm.")}
    qlens = torch.from_numpy(g["qlens"])
    rank = O.sort_rank(qlens)
    perm = torch.from_numpy(g["perm"])
    assert torch.equal(rank[perm], torch.arange(len(perm)))          # stable order == reference sort order
    we = O.query_encoder(sd, torch.from_numpy(g["qvec"]), qlens, torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"]))
    np.testing.assert_allclose(we.detach().numpy(), g["we"], rtol=1e-5, atol=1e-6)
    (we * torch.from_numpy(g["gw"])).sum().backward()
    for k, p in sd.items():
        ref = g["grad_" + k.replace("lstm.", "").replace(".", "_")]
        got = p.grad.numpy() if p.grad.dim() == 1 else p.grad.numpy()[::4, ::3]
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-6, err_msg=k)


def test_fpn(gold):
    g = gold("g8_fpn")
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    gf = torch.Generator().manual_seed(int(g["in_seed"][0]))
    c3 = torch.randn(1, 512, 38, 38, generator=gf)
    c4 = torch.randn(1, 1024, 19, 19, generator=gf)
    c5 = torch.randn(1, 2048, 10, 10, generator=gf)
    outs = O.fpn_forward(sd, c3, c4, c5)
    assert [tuple(o.shape[2:]) for o in outs] == O.feat_sizes_for(300, 300)
    for i, o in enumerate(outs):
        o = o.numpy()
        ref = g[f"p{i}"]
        if o.shape != ref.shape:
            o = o[:, ::8, ::3, ::3]
        np.testing.assert_allclose(o, ref, rtol=1e-4, atol=1e-4)


def test_bottleneck_train_bn(gold):
    g = gold("g8_bottleneck")
    full = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    pre = "backbone.encoder.layer2.0."
    sd = {}
    for k, v in full.items():
        if k.startswith(pre):
            sd[k] = v.clone().requires_grad_(v.is_floating_point() and "running" not in k)
    x = torch.from_numpy(g["x"]).requires_grad_()
    bn = O.BNState(sd, True)
    o = torch.relu(bn(torch.nn.functional.conv2d(x, sd[pre + "conv1.weight"]), pre + "bn1"))
    o = torch.relu(bn(torch.nn.functional.conv2d(o, sd[pre + "conv2.weight"], None, 2, 1), pre + "bn2"))
    o = bn(torch.nn.functional.conv2d(o, sd[pre + "conv3.weight"]), pre + "bn3")
    idt = bn(torch.nn.functional.conv2d(x, sd[pre + "downsample.0.weight"], None, 2), pre + "downsample.1")
    y = torch.relu(o + idt)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.from_numpy(g["gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "conv2.weight"].grad.numpy()[::2, ::2], g["g_conv2"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "bn3.weight"].grad.numpy(), g["g_bn3_w"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "running_mean".join(["bn2.", ""])].numpy(), g["rm_bn2"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd[pre + "bn2.running_var"].numpy(), g["rv_bn2"], rtol=1e-5, atol=1e-7)


def test_head_order(gold):
    g = gold("g9_head")
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    y = O.head_forward(sd, torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=1e-4, atol=1e-4)
    assert torch.equal(sd["att_reg_box.5.bias"], torch.tensor([0, 0, 0, 0, -4.0] * 9))


@pytest.mark.parametrize("tag,hw", [("e2e_128", 128), ("e2e_300", 300)])
def test_end_to_end(gold, tag, hw):
    g = gold("g10_" + tag)
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    bt = O.synthetic_batch(2, hw, hw, seed=int(g["batch_seed"][0]))
    out = O.zsgnet_forward(sd, bt, torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"]))
    fs = [tuple(r) for r in out["feat_sizes"].tolist()]
    assert fs == [tuple(r) for r in g["feat_sizes"].tolist()] == O.feat_sizes_for(hw, hw)
    att, bbx = out["att_out"].detach().numpy(), out["bbx_out"].detach().numpy()
    if hw == 128:
        np.testing.assert_allclose(att, g["att_out"], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(bbx, g["bbx_out"], rtol=1e-3, atol=1e-3)
    else:
        np.testing.assert_allclose(att[:, ::7], g["att_out_s"], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(bbx[:, ::7], g["bbx_out_s"], rtol=1e-3, atol=1e-3)
    anc = torch.from_numpy(O.create_anchors(fs, RATIOS, SCALES).astype(np.float32))
    ls = O.torch_loss(out, bt["annot"], anc)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(ls[k].item(), g[k], rtol=1e-4)
    r = O.zsg_loss(att[..., 0], bbx, bt["annot"].numpy(), anc.numpy())
    np.testing.assert_allclose(r["loss"], g["loss"], rtol=1e-4)
    ls["loss"].backward()
    names = list(g["grad_names"])
    norms = dict(zip(names, g["grad_norms"]))
    for k, v in sd.items():
        if v.requires_grad:
            assert k in norms, k
            np.testing.assert_allclose(v.grad.double().norm().item(), norms[k], rtol=2e-3, atol=1e-6, err_msg=k)
    for k in g.files:
        if k.startswith("grad__"):
            ref = g[k]
            np.testing.assert_allclose(sd[k[6:]].grad.numpy(), ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    np.testing.assert_allclose(sd["backbone.encoder.bn1.running_mean"].numpy(), g["rm_bn1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd["backbone.encoder.layer4.2.bn3.running_var"].numpy(), g["rv_l4"], rtol=1e-4, atol=1e-7)
    e = O.zsg_eval(att[..., 0], bbx, bt["annot"].numpy(), bt["img_size"].numpy(), anc.numpy())
    assert e["Acc"] == g["Acc"] and e["MaxPos"] == g["MaxPos"]
    np.testing.assert_allclose(e["pred_scores"], g["pred_scores"], rtol=1e-4)


def test_ssd_vgg_backbone(gold):
    """config 4 backbone (ssd_vgg.py): oracle vs the reference's SSD.forward + ZSGNet head/loss on seeded weights"""
    g = gold("g11_ssd")
    sd = O.seeded_ssd_state_dict(int(g["seed"][0]))
    assert sorted(sd.keys()) == list(g["keys"])
    for k, v in sd.items():
        v.requires_grad_(True)
    bt = O.synthetic_batch(1, 300, 300, seed=int(g["batch_seed"][0]))
    feats = O.ssd_forward(sd, bt["img"])
    for i, f in enumerate(feats):
        f = f.detach().numpy()
        ref = g[f"feat{i}_s"]
        np.testing.assert_allclose(f[:, ::8, ::3, ::3] if f.shape[2] > 5 else f, ref, rtol=1e-4, atol=1e-5)
    out = O.zsgnet_forward(sd, bt, torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"]), arch="ssd_vgg")
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    np.testing.assert_allclose(out["att_out"].detach().numpy()[:, ::7], g["att_out_s"], rtol=1e-3, atol=1e-4)
    anc = torch.from_numpy(O.create_anchors([tuple(r) for r in g["feat_sizes"].tolist()], RATIOS, SCALES).astype(np.float32))
    ls = O.torch_loss(out, bt["annot"], anc)
    np.testing.assert_allclose(ls["loss"].item(), g["loss"], rtol=1e-4)
    ls["loss"].backward()
    norms = dict(zip(g["grad_names"], g["grad_norms"]))
    for k, v in sd.items():
        if k in norms:
            np.testing.assert_allclose(v.grad.double().norm().item(), norms[k], rtol=2e-3, atol=1e-7, err_msg=k)
        else:
            assert k in set(g["unused"]) and (v.grad is None or float(v.grad.abs().max()) == 0), k
    for k in g.files:
        if k.startswith("grad__"):
            np.testing.assert_allclose(sd[k[6:]].grad.numpy(), g[k], rtol=2e-3, atol=1e-6)


ABLATIONS = {"lang_blind": dict(use_lang=False), "img_blind": dict(use_img=False),
             "both_blind": dict(use_lang=False, use_img=False), "do_norm": dict(do_norm=True),
             "two_heads": dict(use_same_atb=False)}


@pytest.mark.parametrize("tag", sorted(ABLATIONS))
def test_ablation_variants(gold, tag):
    """mdl.py:118-130 (do_norm) and :199-210 / :363-375 (blind heads): outputs, loss, which parameters get a gradient."""
    g = gold("g12_" + tag)
    kw = dict(ABLATIONS[tag])
    same = kw.pop("use_same_atb", True)
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]), head_in=int(g["head_in"][0]), same_atb=same)
    if not same:      # (the golden's stand-in encoder carries an unused FPN of its own, tests/golden/make_golden.py)
        assert set(sd.keys()) == {str(k) for k in g["keys"] if not str(k).startswith("backbone.encoder.fpn.")}
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    bt = O.synthetic_batch(2, 128, 128, seed=int(g["batch_seed"][0]))
    out = O.zsgnet_forward(sd, bt, torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"]), **kw)
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    np.testing.assert_allclose(out["att_out"].detach().numpy(), g["att_out"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(out["bbx_out"].detach().numpy(), g["bbx_out"], rtol=1e-3, atol=1e-3)
    anc = torch.from_numpy(O.create_anchors([tuple(r) for r in g["feat_sizes"].tolist()], RATIOS, SCALES).astype(np.float32))
    ls = O.torch_loss(out, bt["annot"], anc)
    np.testing.assert_allclose(ls["loss"].item(), g["loss"], rtol=1e-4)
    ls["loss"].backward()
    norms = dict(zip(g["grad_names"], g["grad_norms"]))
    unused = set(g["unused"])
    for k, v in sd.items():
        if not v.requires_grad:
            continue
        if k in norms:
            np.testing.assert_allclose(v.grad.double().norm().item(), norms[k], rtol=3e-3, atol=1e-6, err_msg=k)
        else:
            assert k in unused and (v.grad is None or float(v.grad.abs().max()) == 0), k
    for k in g.files:
        if k.startswith("grad__"):
            ref, got = g[k], sd[k[6:].replace("_s", "") if k.endswith("weight_s") else k[6:]].grad.numpy()
            got = got[::8, ::5] if k.endswith("weight_s") else got
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-5 * max(1.0, np.abs(ref).max()), err_msg=k)
    # the encoder runs (and updates its BatchNorm running statistics) even when the head never sees the image
    np.testing.assert_allclose(sd["backbone.encoder.bn1.running_mean"].numpy(), g["rm_bn1"], rtol=1e-5, atol=1e-7)


def test_fpn_600_branch(gold):
    """fpn_resnet.py:173-174: resize_img == [600, 600] drops p3 and the global-pool level (pins the oracle's 600 path)"""
    g = gold("g8_fpn600")
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    gf = torch.Generator().manual_seed(int(g["in_seed"][0]))
    c3 = torch.randn(1, 512, 75, 75, generator=gf)
    c4 = torch.randn(1, 1024, 38, 38, generator=gf)
    c5 = torch.randn(1, 2048, 19, 19, generator=gf)
    outs = O.fpn_forward(sd, c3, c4, c5, six_hundred=True)
    assert [list(o.shape[2:]) for o in outs] == g["sizes"].tolist() == [list(s) for s in O.feat_sizes_for(600, 600, six_hundred=True)]
    for i, o in enumerate(outs):
        o = o.numpy()
        ref = g[f"p{i}"]
        if o.shape != ref.shape:
            o = o[:, ::8, ::3, ::3]
        np.testing.assert_allclose(o, ref, rtol=1e-4, atol=1e-4)


def test_basicblock_stage_and_trunk(gold):
    """fpn_resnet.py:26-58 BasicBlock (ResNet(1, BasicBlock, [2,2,2,2])): one stride-2 block fwd/bwd in train-mode BN and the
    whole trunk's FPN taps (pins the oracle's resnet18/34 path)"""
    g = gold("g8_basicblock")
    full = O.seeded_state_dict("resnet18", int(g["seed"][0]))
    pre = "backbone.encoder.layer2.0."
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in full.items() if k.startswith(pre)}
    x = torch.from_numpy(g["x"]).requires_grad_()
    bn = O.BNState(sd, True)
    Fn = torch.nn.functional
    o = torch.relu(bn(Fn.conv2d(x, sd[pre + "conv1.weight"], None, 2, 1), pre + "bn1"))
    o = bn(Fn.conv2d(o, sd[pre + "conv2.weight"], None, 1, 1), pre + "bn2")
    idt = bn(Fn.conv2d(x, sd[pre + "downsample.0.weight"], None, 2), pre + "downsample.1")
    y = torch.relu(o + idt)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=1e-4, atol=1e-5)
    (y * torch.from_numpy(g["gy"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["gx"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "conv1.weight"].grad.numpy()[::2, ::2], g["g_conv1"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "conv2.weight"].grad.numpy()[::2, ::2], g["g_conv2"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "bn2.weight"].grad.numpy(), g["g_bn2_w"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "downsample.0.weight"].grad.numpy(), g["g_ds"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(sd[pre + "bn1.running_mean"].numpy(), g["rm_bn1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(sd[pre + "bn1.running_var"].numpy(), g["rv_bn1"], rtol=1e-5, atol=1e-7)
    sd2 = {k: v.clone() for k, v in full.items()}
    c3, c4, c5 = O.encoder_forward(sd2, torch.from_numpy(g["img"]), "resnet18", O.BNState(sd2, True))
    np.testing.assert_allclose(c3.numpy()[:, ::4], g["c3"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(c4.numpy()[:, ::8], g["c4"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(c5.numpy()[:, ::16], g["c5"], rtol=5e-4, atol=5e-5)


def test_learnable_task_first_step_vs_reference(gold):
    """G15 (the reference trained on O.learnable_batch, tests/golden/make_golden.py gen_learnable): the oracle's loss on the first
    batch from the same seeded start equals the reference's — pins the proxy task's inputs (batch generator, start weights, LSTM start
    states) that tests/test_gpu_fullshape.py::test_learnable_task_reaches_the_same_accuracy trains the HIP model on."""
    import torch
    g = gold("g15_learnable")
    S, B = int(g["S"][0]), int(g["B"][0])
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    bt = O.learnable_batch(B, S, seed=100)
    gq = torch.Generator().manual_seed(8)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    with torch.no_grad():
        out = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet50")
    assert out["feat_sizes"].tolist() == g["feat_sizes"].tolist()
    r, s = O.default_ratios_scales()
    anc = torch.from_numpy(O.create_anchors([tuple(v) for v in out["feat_sizes"].tolist()], r, s).astype(np.float32))
    ls = O.torch_loss(out, bt["annot"], anc)
    np.testing.assert_allclose(float(ls["loss"]), g["losses"][0], rtol=2e-5)
    assert g["losses"][-10:].mean() < 0.01 * g["losses"][0] and g["hits"].sum() >= 0.95 * 256      # (the reference did learn the task)


def test_g16_oracle_reproduces_the_reference_trajectory_start(gold):
    """G16 (the reference's own 12-step training run at the configs[1] shape, tests/golden/make_golden.py gen_trajectory): the oracle's
    first step from the same seeded start on the same batch must give the reference's first loss (rel 2e-5) — the later steps are what
    the GPU suite steps the HIP model through (test_gpu_fullshape.py)."""
    g = gold("g16_trajectory")
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    bt = O.synthetic_batch(16, 300, 300, seed=int(g["batch_seed0"][0]))
    gq = torch.Generator().manual_seed(int(g["hc_seed"][0]))
    h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
    with torch.no_grad():
        ref = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet50")
        r, s = O.default_ratios_scales()
        anc = torch.from_numpy(O.create_anchors([tuple(x) for x in ref["feat_sizes"].tolist()], r, s).astype(np.float32))
        ls = O.torch_loss(ref, bt["annot"], anc)
    np.testing.assert_allclose(float(ls["loss"]), float(g["losses"][0]), rtol=2e-5)
    assert len(g["losses"]) == 12 and g["losses"][-1] < 0.5 * g["losses"][0]
