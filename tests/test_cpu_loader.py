"""The batch producer (zsgnet_pytorch_amd/dat_loader.py) against the reference's ImgQuDataset / collater run on the same
tiny on-disk dataset (tests/golden/g13_dataset.npz: images, CSV rows, word table, the reference's items and batch)."""
import os

import numpy as np
import pytest
import torch


@pytest.fixture()
def tiny(tmp_path, gold):
    import PIL.Image
    g = gold("g13_dataset")
    for k in "abc":
        PIL.Image.fromarray(g["png_" + k]).save(tmp_path / f"{k}.png")
    with open(tmp_path / "d.csv", "w") as f:
        f.write("img_id,bbox,query\n")
        for i, b, q in zip(g["csv_img"], g["csv_bbox"], g["csv_query"]):
            f.write(f'{i},"{[float(v) for v in b]}","{q}"\n')
    np.savez(tmp_path / "vec.npz", words=g["words"], vectors=g["table"])
    from zsgnet_pytorch_amd.config import get_cfg
    cfg = get_cfg(resize_img=[int(v) for v in g["resize_img"]], word_vectors=str(tmp_path / "vec.npz"), ds_to_use="refclef", bs=3, bsv=2, nw=0, nwv=0,
                  **{"ds_info.refclef.img_dir": str(tmp_path), "ds_info.refclef.trn_csv_file": str(tmp_path / "d.csv"),
                     "ds_info.refclef.val_csv_file": str(tmp_path / "d.csv"), "ds_info.refclef.test_csv_file": str(tmp_path / "d.csv")})
    return g, cfg, tmp_path


def test_items_and_collater_match_reference(tiny):
    g, cfg, root = tiny
    from zsgnet_pytorch_amd import dat_loader as D
    ds = D.ImgQuDataset(cfg, root / "d.csv", "refclef")
    assert len(ds) == 5
    items = [ds[i] for i in range(5)]
    for i, it in enumerate(items):
        assert set(it) == {"img", "idxs", "qvec", "qlens", "annot", "orig_annot", "img_size"}
        for k, v in it.items():
            ref = g[f"item{i}_{k}"]
            assert tuple(v.shape) == ref.shape, (i, k)
            assert np.array_equal(v.numpy(), ref), f"item {i} field {k} differs from the reference"          # bit-exact, incl. PIL resize
    b = D.collater(items[:3])
    for k, v in b.items():
        assert str(v.dtype) == str(g["batchdtype_" + k]) == "torch.float32", k          # collater casts EVERY field to float
        assert np.array_equal(v.numpy(), g["batch_" + k]), k
    assert b["qvec"].shape[1] == int(b["qlens"].max())                                   # cut to the longest query


def test_uint8_path_and_loaders(tiny):
    g, cfg, root = tiny
    from zsgnet_pytorch_amd import dat_loader as D
    ds8 = D.ImgQuDataset(cfg, root / "d.csv", "refclef", gpu_normalise=True)
    it = ds8[1]
    assert it["img"].dtype == torch.uint8 and it["img"].shape == (40, 48, 3)             # HWC, resize_img = [W, H]
    host = it["img"].permute(2, 0, 1).float().div(255)
    assert np.array_equal(host.numpy(), g["item1_img"])                                  # what the GPU kernel must reproduce
    b = D.collater([ds8[0], ds8[1]])
    assert b["img"].dtype == torch.uint8 and b["idxs"].dtype == torch.float32
    data = D.get_data(cfg, prefetch=False)
    tr = list(data.train_dl)
    assert len(tr) == 1 and tr[0]["img"].shape[0] == 3                                   # drop_last on the training loader
    va = list(data.valid_dl)
    assert [int(i) for bt in va for i in bt["idxs"]] == [0, 1, 2, 3, 4]                  # sequential, keeps the tail
    assert set(data.test_dl) == {"test0"}
    # names the reference's _read_annotations does not know still load; flickr ids get the '.jpg' suffix
    cfg2 = cfg.clone()
    cfg2["ds_info"]["flickr30k_c0"]["img_dir"] = str(root)
    assert D.ImgQuDataset(cfg2, root / "d.csv", "flickr30k_c0").files[0] == "a.png.jpg"
    with pytest.raises(NotImplementedError):
        D.embed_query(ds8.embedder, "   ")


def test_distributed_sampler_rule():
    """dat_loader.py:36-65: per-epoch permutation, padded with its own head, contiguous rank slices."""
    from zsgnet_pytorch_amd.dat_loader import NewDistributedSampler
    ds = list(range(10))
    parts = []
    for r in range(4):
        s = NewDistributedSampler(ds, num_replicas=4, rank=r, shuffle=True)
        s.set_epoch(3)
        parts.append(list(s))
    perm = torch.randperm(10, generator=torch.Generator().manual_seed(3)).tolist()
    perm += perm[:2]
    assert [i for p in parts for i in p] == perm and all(len(p) == 3 for p in parts)
    s = NewDistributedSampler(ds, num_replicas=4, rank=3, shuffle=False)
    assert list(s) == [9, 0, 1]


def _two_pass(D, img, out_w, out_h):
    """the product's tap tables (dat_loader.resize_tables) applied with plain integer arithmetic — what zsg_resize_u8 computes"""
    def one(src, n_out):
        n_in = src.shape[0]
        if n_in == n_out:
            return src
        b, c, ks = D.resize_tables(n_in, n_out)
        out = np.empty((n_out,) + src.shape[1:], np.uint8)
        s = src.astype(np.int64)
        for i in range(n_out):
            x0, n = int(b[i, 0]), int(b[i, 1])
            acc = (1 << 21) + np.tensordot(c[i, :n].astype(np.int64), s[x0:x0 + n], axes=(0, 0))
            out[i] = np.clip(acc >> 22, 0, 255)
        return out
    return one(one(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2), out_h)


def test_pil_resize_restatement_and_tap_tables_vs_reference_loader(gold):
    """N2 (GPU-side resize): the reference loader's `img.resize(...)` (dat_loader.py:121, PIL's default filter) — golden g14 holds raw
    images of seven sizes and what the reference's ImgQuDataset made of them; g13 adds three up-scaling cases.  Both the oracle's
    restatement of Pillow's resampler and the product's tap tables must reproduce them byte for byte."""
    from oracle import zsg_oracle as O
    from zsgnet_pytorch_amd import dat_loader as D
    g = gold("g14_resize")
    ow, oh = (int(v) for v in g["resize_img"])
    for nm in g["names"]:
        raw, ref = g["raw_" + str(nm)[0]], g["out_" + str(nm)[0]]
        assert np.array_equal(O.pil_resize_u8(raw, ow, oh), ref), f"oracle restatement differs from the reference loader on {nm} {raw.shape}"
        assert np.array_equal(_two_pass(D, raw, ow, oh), ref), f"tap tables differ from the reference loader on {nm} {raw.shape}"
    g3 = gold("g13_dataset")
    ow, oh = (int(v) for v in g3["resize_img"])
    for i, nm in enumerate(g3["csv_img"]):
        raw = g3["png_" + str(nm)[0]]
        ref = np.rint(g3[f"item{i}_img"].astype(np.float64) * 255).astype(np.uint8).transpose(1, 2, 0)
        assert np.array_equal(O.pil_resize_u8(raw, ow, oh), ref) and np.array_equal(_two_pass(D, raw, ow, oh), ref), (nm, raw.shape)


def test_pil_resize_restatement_vs_installed_pillow():
    """the same two implementations against the Pillow that is installed, on photo-sized geometries (500x375 -> 300x300 ...)"""
    PIL = pytest.importorskip("PIL.Image")
    from oracle import zsg_oracle as O
    from zsgnet_pytorch_amd import dat_loader as D
    rng = np.random.default_rng(5)
    for (h, w, oh, ow) in [(375, 500, 300, 300), (480, 640, 300, 300), (200, 150, 300, 300), (300, 300, 300, 300), (301, 299, 300, 300),
                           (683, 1024, 600, 600), (2, 3, 300, 300)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(PIL.fromarray(a).resize((ow, oh)))
        assert np.array_equal(O.pil_resize_u8(a, ow, oh), ref), (h, w)
        assert np.array_equal(_two_pass(D, a, ow, oh), ref), (h, w)
