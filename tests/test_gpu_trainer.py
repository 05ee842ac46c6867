"""The Learner / CLI plumbing on the GPU with synthetic batches: train -> validate -> checkpoint on
improvement -> prediction pickles in the reference's format -> offline eval_script accuracy == in-loop accuracy."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_validate_predictions_and_eval_script(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zsgnet_pytorch_amd import eval_script
    from zsgnet_pytorch_amd.main_dist import main_dist
    kw = dict(resnet_arch="resnet18", bs=2, bsv=2, resize_img=[96, 96], steps_per_epoch=10, epochs=1, tmp_path=str(tmp_path), synthetic=True)
    learn = main_dist("t0", **{k: str(v) for k, v in kw.items()})
    assert learn.num_it == 10 and learn.num_epoch == 1
    ck = torch.load(learn.model_file, map_location="cpu") if learn.model_file.exists() else None
    # predictions of the validation pass: written together with the checkpoint when the metric improved (utils.py:606-611)
    vp = learn.predictions_dir / "val_preds_t0.pkl"
    assert (ck is not None) == vp.exists()
    # test pass: always writes '<name>_preds.pkl' (utils.py:646-665)
    res = learn.testing(learn.data.test_dl)
    pf = learn.predictions_dir / "synthetic_test_preds.pkl"
    preds = pickle.load(open(pf, "rb"))
    dl = learn.data.test_dl["synthetic_test"]
    assert isinstance(preds, list) and len(preds) == len(dl) * 2 and set(preds[0]) == {"id", "pred_boxes", "pred_scores"}
    assert [int(p["id"]) for p in preds] == list(range(len(preds))) and len(preds[0]["pred_boxes"]) == 4
    # ground-truth CSV in pixels, x1y1x2y2 (what the reference CSVs hold): the evaluator's own conversion of `annot`
    dl.epoch -= 1                                            # replay the same batches
    rows = []
    for bt in dl:
        a, sz = bt["annot"].double(), bt["img_size"].double()
        px = (a + 1) / 2 * torch.cat([sz, sz], 1)            # y1 x1 y2 x2 in pixels
        rows += px[:, [1, 0, 3, 2]].tolist()
    with open(tmp_path / "gt.csv", "w") as f:
        f.write("img_id,bbox,query\n")
        for i, b in enumerate(rows):
            f.write(f'{i}.jpg,"{b}",q\n')
    acc, corr, tot = eval_script.evaluate(pf, tmp_path / "gt.csv")
    assert tot == len(preds)
    assert abs(acc - res["synthetic_test"]["Acc"]) < 1e-6, "offline accuracy must equal the in-loop metric (no box sits exactly at IoU 0.5)"
    # resume from the checkpoint + validation only
    if ck is not None:
        assert set(ck) >= {"model_state_dict", "optimizer_state_dict", "num_it", "num_epoch", "best_met"}
        again = main_dist("t0", **{k: str(v) for k, v in dict(kw, resume=True, only_val=True).items()})
        assert again.num_it == 10


def test_real_data_loader_uint8_ingest_and_training(tmp_path, gold):
    """dat_loader path end to end: PIL decode -> uint8 HWC pinned batch -> side-stream copy -> /255 + NHWC4 on the GPU
    (bit-identical network output to the reference's float NCHW batch) -> a few training steps through the CLI."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import PIL.Image
    from zsgnet_pytorch_amd import dat_loader as D
    from zsgnet_pytorch_amd import config, mdl
    from zsgnet_pytorch_amd.main_dist import main_dist
    g = gold("g13_dataset")
    for k in "abc":
        PIL.Image.fromarray(g["png_" + k]).save(tmp_path / f"{k}.png")
    with open(tmp_path / "d.csv", "w") as f:
        f.write("img_id,bbox,query\n")
        for i, b, q in zip(g["csv_img"], g["csv_bbox"], g["csv_query"]):
            f.write(f'{i},"{[float(v) for v in b]}","{q}"\n')
    np.savez(tmp_path / "vec.npz", words=g["words"], vectors=g["table"])
    kw = {"resize_img": [96, 64], "word_vectors": str(tmp_path / "vec.npz"), "ds_to_use": "refclef", "bs": 2, "bsv": 2, "nw": 0, "nwv": 0,
          "resnet_arch": "resnet18", "synthetic": False, "epochs": 2, "tmp_path": str(tmp_path / "run"),
          "ds_info.refclef.img_dir": str(tmp_path), "ds_info.refclef.trn_csv_file": str(tmp_path / "d.csv"),
          "ds_info.refclef.val_csv_file": str(tmp_path / "d.csv"), "ds_info.refclef.test_csv_file": str(tmp_path / "d.csv")}
    cfg = config.get_cfg(**kw)
    ds_f = D.ImgQuDataset(cfg, tmp_path / "d.csv", "refclef", gpu_normalise=False)
    ds_u = D.ImgQuDataset(cfg, tmp_path / "d.csv", "refclef", gpu_normalise=True)
    bf, bu = D.collater([ds_f[0], ds_f[1]]), D.collater([ds_u[0], ds_u[1]])
    assert bu["img"].dtype == torch.uint8 and tuple(bu["img"].shape) == (2, 64, 96, 3) and tuple(bf["img"].shape) == (2, 3, 64, 96)
    net = mdl.get_default_net(9, cfg).to("cuda").eval()
    h0, c0 = torch.zeros(2, 2, 128), torch.zeros(2, 2, 128)
    with torch.no_grad():
        of = net({**{k: v.cuda() for k, v in bf.items()}, "h0": h0, "c0": c0})["att_bbx_out"].clone()
        ou = net({**{k: v.cuda() for k, v in bu.items()}, "h0": h0, "c0": c0})["att_bbx_out"].clone()
    # the kernel itself: bit-identical to pil2tensor(...).float().div_(255), channel 3 zero
    from zsgnet_pytorch_amd import _lib as L
    u8 = bu["img"].cuda()
    x4 = torch.full((2, 64, 96, 4), float("nan"), device="cuda")
    L.check(L.lib.zsg_u8hwc_to_nhwc4(u8.data_ptr(), 2 * 64 * 96, x4.data_ptr(), L.stream_ptr()), "u8hwc_to_nhwc4")
    assert torch.equal(x4[..., :3].cpu(), bf["img"].permute(0, 2, 3, 1)) and float(x4[..., 3].abs().max()) == 0.0
    # the network on either ingest path (split-K launches use fp32 atomics: equal up to summation order)
    assert float((of - ou).abs().max()) < 1e-5
    # gpu_resize: the workers only decode; the prefetcher resizes the raw images (different sizes: a list) on the GPU — the batch it
    # hands over is byte-identical to the one whose images PIL resized in the worker
    ds_r = D.ImgQuDataset(cfg, tmp_path / "d.csv", "refclef", gpu_resize=True)
    br = D.collater([ds_r[0], ds_r[1]])
    # (round 5: ONE flat uint8 tensor + the sizes — one upload, two launches for the whole batch)
    assert br["img"].dtype == torch.uint8 and br["img"].dim() == 1 and br["img"].numel() == (30 * 40 + 47 * 33) * 3
    assert br["img_hw"].tolist() == [[30, 40], [47, 33]]
    got = next(iter(D.DevicePrefetcher([br], "cuda", resize_hw=(64, 96))))
    torch.cuda.synchronize()
    assert got["img"].dtype == torch.uint8 and torch.equal(got["img"].cpu(), bu["img"]) and "img_hw" not in got
    assert all(torch.equal(got[k].cpu(), bu[k]) for k in bu if k != "img")
    # a caller that still hands over a LIST of raw images gets the same batch; several batches in a row reuse the job-table slots
    raw_list = {k: v for k, v in br.items() if k not in ("img", "img_hw")}
    raw_list["img"] = [ds_r[0]["img"], ds_r[1]["img"]]
    for g2 in D.DevicePrefetcher([raw_list, br, raw_list, br, br, br], "cuda", resize_hw=(64, 96)):
        torch.cuda.synchronize()
        assert torch.equal(g2["img"].cpu(), bu["img"])
    learn = main_dist("real0", **{k: str(v) for k, v in kw.items()})
    assert isinstance(learn.data.train_dl, D.DevicePrefetcher) and learn.num_it == 4        # 5 rows, bs 2, drop_last, 2 epochs
    res = learn.testing(learn.data.test_dl)
    preds = pickle.load(open(learn.predictions_dir / "test0_preds.pkl", "rb"))
    assert sorted(int(p["id"]) for p in preds) == [0, 1, 2, 3, 4] and 0.0 <= res["test0"]["Acc"] <= 1.0


def test_gpu_resize_bit_identical_to_the_reference_loader(gold):
    """N2: zsg_resize_u8 (through GpuResizer and through the raw C ABI) against what the reference's ImgQuDataset produced (golden
    g14: seven image sizes -> 100x100; g13: three up-scaling cases -> 48x40) — byte for byte, then the fused /255 + NHWC4 step."""
    import ctypes as C
    from zsgnet_pytorch_amd import dat_loader as D
    from zsgnet_pytorch_amd._lib import lib, stream_ptr
    g = gold("g14_resize")
    ow, oh = (int(v) for v in g["resize_img"])
    rz = D.GpuResizer((oh, ow))
    raws = [torch.from_numpy(np.ascontiguousarray(g["raw_" + str(nm)[0]])).cuda() for nm in g["names"]]
    out = rz(raws)
    torch.cuda.synchronize()
    for i, nm in enumerate(g["names"]):
        assert np.array_equal(out[i].cpu().numpy(), g["out_" + str(nm)[0]]), f"{nm} {tuple(raws[i].shape)}: the GPU resize differs from the reference loader"
    # the flat form (what the collater produces: one upload) gives the same bytes, and so does the per-image C entry point
    flat, hw = D.flatten_raw([r.cpu() for r in raws])
    out_f = rz.resize_flat(flat.cuda(), hw)
    torch.cuda.synchronize()
    assert torch.equal(out_f, out)
    # an axis that keeps its length (Pillow skips that pass; the batched launches run it with the identity table)
    import PIL.Image
    a = np.random.default_rng(0).integers(0, 256, (oh, 57, 3), dtype=np.uint8)
    ref = np.asarray(PIL.Image.fromarray(a).resize((ow, oh)))
    got1 = rz([torch.from_numpy(a).cuda()])
    torch.cuda.synchronize()
    assert np.array_equal(got1[0].cpu().numpy(), ref)
    g3 = gold("g13_dataset")
    ow3, oh3 = (int(v) for v in g3["resize_img"])
    rz3 = D.GpuResizer((oh3, ow3))
    names = [str(n) for n in g3["csv_img"]]
    out3 = rz3([torch.from_numpy(np.ascontiguousarray(g3["png_" + n[0]])).cuda() for n in names])
    nhwc4 = torch.empty(len(names), oh3, ow3, 4, device="cuda")
    assert lib.zsg_u8hwc_to_nhwc4(out3.data_ptr(), len(names) * oh3 * ow3, nhwc4.data_ptr(), stream_ptr()) == 0
    torch.cuda.synchronize()
    for i in range(len(names)):
        ref = torch.from_numpy(g3[f"item{i}_img"])                                   # [3, H, W] float, the reference's item
        assert torch.equal(nhwc4[i, :, :, :3].permute(2, 0, 1).cpu(), ref), f"{names[i]}: resize + /255 differs from the reference item"
    # raw C ABI: a size change without its tap table is an argument error (no silent copy), a same-size "resize" is a copy
    x = raws[0]
    o = torch.empty(oh, ow, 3, dtype=torch.uint8, device="cuda")
    assert lib.zsg_resize_u8(x.data_ptr(), x.shape[0], x.shape[1], 3, None, None, 0, None, None, 0, oh, ow, None, o.data_ptr(), stream_ptr()) == -1
    assert b"tap table" in lib.zsg_last_error()
    same = torch.empty_like(x)
    assert lib.zsg_resize_u8(x.data_ptr(), x.shape[0], x.shape[1], 3, None, None, 0, None, None, 0, x.shape[0], x.shape[1], None, same.data_ptr(), stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(same, x)
