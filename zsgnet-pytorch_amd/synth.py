"""Synthetic batches with the reference's batch contract (dat_loader.py:136-144, collater :187-196): every field a
float tensor — img [B,3,H,W] in [0,1), qvec [B,T,300], qlens [B], annot [B,4] y1x1y2x2 in [-1,1], idxs [B],
img_size [B,2] (h, w).  Distributions: SURVEY.md §8(d).  The CSV/PIL/spaCy loader itself is out of scope (§8 'next' N2)."""
from typing import Dict, Iterator

import torch


def synthetic_batch(B: int, H: int = 300, W: int = 300, T: int = 20, seed: int = 1234, tmax: int = 20, emb: int = 300) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    qvec = torch.randn(B, T, emb, generator=g) * 0.35
    qlens = torch.randint(1, tmax + 1, (B,), generator=g).float()
    qlens[0] = float(tmax)
    c = torch.rand(B, 2, generator=g) * 1.2 - 0.6
    s = torch.rand(B, 2, generator=g) * 0.8 + 0.1
    annot = torch.cat([c - s / 2, c + s / 2], dim=1).clamp(-1, 1)
    return dict(img=img, qvec=qvec, qlens=qlens, annot=annot, idxs=torch.arange(B).float(),
                img_size=torch.tensor([[360.0, 480.0]]).repeat(B, 1))


class SyntheticLoader:
    """Iterable with len(): `steps` batches per epoch, deterministic per (seed, rank, step); a pool of `pool` distinct
    batches is generated once on the host and kept on the device (generating 16 x 3 x 300 x 300 uniform numbers on the CPU
    every step costs as much as the GPU training step itself), then cycled with fresh row ids."""

    def __init__(self, cfg, bs: int, steps: int, seed: int = 1234, rank: int = 0, device="cuda", pool: int = 8):
        self.cfg, self.bs, self.steps, self.seed, self.rank, self.device = cfg, bs, steps, seed, rank, device
        self.epoch = 0
        self.pool, self._cache = pool, {}

    def __len__(self):
        return self.steps

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        H, W = self.cfg["resize_img"]
        dev = torch.device(self.device if torch.cuda.is_available() else "cpu")
        for i in range(self.steps):
            k = i % self.pool
            if k not in self._cache:
                bt = synthetic_batch(self.bs, H, W, seed=self.seed + 1000003 * self.rank + k, emb=self.cfg["emb_dim"])
                self._cache[k] = {n: v.to(dev) for n, v in bt.items()}
            bt = dict(self._cache[k])
            bt["idxs"] = bt["idxs"] + float((self.rank * self.steps + i) * self.bs)      # dataset row ids (dat_loader.py:140), unique per sample
            yield bt
        self.epoch += 1


class DataWrap:
    """utils.py:116-120"""

    def __init__(self, train_dl, valid_dl, test_dl=None, path="./tmp"):
        self.train_dl, self.valid_dl, self.test_dl, self.path = train_dl, valid_dl, test_dl, path


def get_data(cfg, rank: int = 0) -> DataWrap:
    """dat_loader.get_data counterpart for synthetic runs (per-rank batch = cfg.bs, dat_loader.py:212-215)."""
    steps = int(cfg["steps_per_epoch"])
    return DataWrap(SyntheticLoader(cfg, cfg["bs"], steps, 1234, rank), SyntheticLoader(cfg, cfg["bsv"], max(1, steps // 5), 4321, rank),
                    {"synthetic_test": SyntheticLoader(cfg, cfg["bsv"], max(1, steps // 5), 9999, rank)}, cfg["tmp_path"])
