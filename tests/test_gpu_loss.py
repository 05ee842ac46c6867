"""GPU parity of the fused matching/loss/evaluator kernels (csrc/loss.hip) through the product modules
(zsgnet_pytorch_amd.loss / .evaluator / .anchors): indices and masks bit-exact, loss scalars rel 1e-5, gradients rel 2e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import zsg_oracle as O  # noqa: E402

RATIOS, SCALES = O.default_ratios_scales()


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zsgnet_pytorch_amd import anchors, config, evaluator, loss
    return anchors, config, evaluator, loss


def test_anchor_tables_match_golden(M, gold):
    anchors, *_ = M
    g = gold("g2_anchors")
    a = anchors.create_anchors(O.feat_sizes_for(300, 300), RATIOS, SCALES, device="cuda")
    assert a.dtype == torch.float32 and np.array_equal(a.cpu().numpy(), g["a300_f32"])
    g1 = gold("g1_grid")
    for k in g1.files:
        _, h, w = k.split("_")
        assert np.array_equal(anchors.create_grid((int(h), int(w))).numpy(), g1[k])


def test_iou_bit_exact(M, gold):
    anchors, *_ = M
    g = gold("g3_iou")
    anc = torch.from_numpy(gold("g2_anchors")["a300_f32"]).cuda()
    boxes = torch.from_numpy(g["boxes"]).cuda()
    iou = anchors.IoU_values(boxes, anc).cpu().numpy()
    ref = O.iou_values(g["boxes"], anc.cpu().numpy())
    assert np.array_equal(iou, ref), f"{(iou != ref).sum()} IoU values differ"
    assert np.array_equal(iou.argmax(1).astype(np.int32), g["argmax"])
    rows, cols = np.nonzero(iou > np.float32(0.6))
    assert np.array_equal(rows.astype(np.int32), g["pos_rows"]) and np.array_equal(cols.astype(np.int32), g["pos_cols"])


def _mods(M, flags=None):
    anchors, config, evaluator, loss = M
    cfg = config.get_cfg(**(flags or {}))
    r, s = config.ratios_scales(cfg)
    return loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)


@pytest.mark.parametrize("tag,flags", [("b1", {}), ("b2", {}), ("b16", {}), ("nomulti", dict(use_multi=False)),
                                       ("nofocal", dict(use_focal=False)), ("softmax", dict(use_multi=False, use_softmax=True)),
                                       ("nan", {})])
def test_loss_eval_vs_golden_small(M, gold, tag, flags):
    g = gold("g5_loss_eval_small")
    lf, ev = _mods(M, flags)
    anc = torch.from_numpy(g["anchors"]).cuda()
    lf.anchs = anc
    ev.anchs = anc
    att, bbx = torch.from_numpy(g[f"{tag}_att"]), torch.from_numpy(g[f"{tag}_bbx"])
    out5 = torch.cat([bbx, att], dim=2).cuda().requires_grad_()
    out = dict(att_bbx_out=out5, att_out=out5[..., 4:5], bbx_out=out5[..., :4], feat_sizes=None, num_f_out=torch.tensor([1]))
    inp = dict(annot=torch.from_numpy(g[f"{tag}_annot"]).cuda(), img_size=torch.from_numpy(g[f"{tag}_img_size"]).cuda(),
               idxs=torch.arange(att.shape[0]).float().cuda())
    ls = lf(out, inp)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(ls[k].item(), g[f"{tag}_{k}"], rtol=1e-5, err_msg=k)
    ls["loss"].backward()
    gr = out5.grad.cpu().numpy()
    if tag == "nan":
        assert np.all(gr == 0)
    else:
        np.testing.assert_allclose(gr[..., 4:5], g[f"{tag}_g_att"], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(gr[..., :4], g[f"{tag}_g_bbx"], rtol=2e-5, atol=1e-9)
    # matching indices: exact against the oracle
    r = O.zsg_loss(g[f"{tag}_att"][..., 0], g[f"{tag}_bbx"], g[f"{tag}_annot"], g["anchors"], **flags)
    assert np.array_equal(lf.match_idx.cpu().numpy(), r["best"].astype(np.int32))
    assert np.array_equal(lf.npos.cpu().numpy(), r["mask"].sum(1).astype(np.int32))
    em = ev(out, inp)
    assert np.array_equal(ev.pred_idx.cpu().numpy(), g[f"{tag}_pred_ids"].astype(np.int32))
    assert em["Acc"].item() == g[f"{tag}_Acc"] and em["MaxPos"].item() == g[f"{tag}_MaxPos"]
    np.testing.assert_allclose(em["pred_boxes"].cpu().numpy(), g[f"{tag}_pred_boxes"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(em["pred_scores"].cpu().numpy(), g[f"{tag}_pred_scores"], rtol=1e-6)


def test_loss_eval_full_size(M, gold):
    """A = 17460 (BASELINE config 2 anchors): golden loss values + oracle mask / arg-max (exact) + gradients."""
    g = gold("g5_loss_eval_full")
    lf, ev = _mods(M)
    gen = torch.Generator().manual_seed(int(g["gen_seed"][0]))
    att = torch.randn(2, 17460, 1, generator=gen) * 1.5 - 3.0
    bbx = torch.randn(2, 17460, 4, generator=gen) * 0.6
    out5 = torch.cat([bbx, att], dim=2).cuda().requires_grad_()
    fs = torch.tensor(O.feat_sizes_for(300, 300))
    out = dict(att_bbx_out=out5, feat_sizes=fs, num_f_out=torch.tensor([6]))
    inp = dict(annot=torch.from_numpy(g["annot"]).cuda(), img_size=torch.from_numpy(g["img_size"]).cuda(), idxs=torch.arange(2.0).cuda())
    ls = lf(out, inp)
    for k in ("loss", "cls_ls", "box_ls"):
        np.testing.assert_allclose(ls[k].item(), g[k], rtol=1e-5)
    ls["loss"].backward()
    gr = out5.grad.cpu().numpy()
    np.testing.assert_allclose(gr[:, ::13, 4:5], g["g_att_s"], rtol=2e-5, atol=1e-10)
    np.testing.assert_allclose(gr[:, ::13, :4], g["g_bbx_s"], rtol=2e-5, atol=1e-10)
    np.testing.assert_allclose(np.abs(gr[..., 4]).astype(np.float64).sum(), g["g_att_abs_sum"], rtol=1e-5)
    r = O.zsg_loss(att.numpy()[..., 0], bbx.numpy(), g["annot"], lf.anchs.cpu().numpy())
    assert np.array_equal(lf.match_idx.cpu().numpy(), r["best"].astype(np.int32))
    em = ev(out, inp)
    assert em["Acc"].item() == g["Acc"] and em["MaxPos"].item() == g["MaxPos"]
    np.testing.assert_allclose(em["pred_boxes"].cpu().numpy(), g["pred_boxes"], rtol=1e-5, atol=1e-3)


def test_matching_properties_large_batch(M):
    """size-independent properties at B=64: every sample has >= 1 positive, the arg-max anchor is the oracle's
    (ties -> lowest index), cls gradient sums, and the loss is invariant to a permutation of the batch."""
    lf, ev = _mods(M)
    B = 64
    bt = O.synthetic_batch(B, 8, 8, seed=321)
    gen = torch.Generator().manual_seed(8)
    out5 = torch.cat([torch.randn(B, 17460, 4, generator=gen) * 0.5, torch.randn(B, 17460, 1, generator=gen) - 4], dim=2).cuda()
    fs = torch.tensor(O.feat_sizes_for(300, 300))
    inp = dict(annot=bt["annot"].cuda(), img_size=bt["img_size"].cuda(), idxs=bt["idxs"].cuda())
    ls = lf(dict(att_bbx_out=out5, feat_sizes=fs, num_f_out=torch.tensor([6])), inp)
    iou = O.iou_values(bt["annot"].numpy(), lf.anchs.cpu().numpy())
    mask, best = O.match_mask(iou, 0.6)
    assert np.array_equal(lf.match_idx.cpu().numpy(), best.astype(np.int32))
    assert np.array_equal(lf.npos.cpu().numpy(), mask.sum(1).astype(np.int32)) and lf.npos.min().item() >= 1
    perm = torch.randperm(B, generator=gen)
    ls2 = lf(dict(att_bbx_out=out5[perm.cuda()].contiguous(), feat_sizes=fs, num_f_out=torch.tensor([6])),
             dict(annot=bt["annot"][perm].cuda(), img_size=bt["img_size"].cuda(), idxs=bt["idxs"].cuda()))
    np.testing.assert_allclose(ls["loss"].item(), ls2["loss"].item(), rtol=1e-5)
