"""Batch producer for real datasets — counterpart of the reference's `code/dat_loader.py`.

Contract kept (dat_loader.py:98-146, 187-196): a CSV with columns `img_id, bbox, query` (bbox "[x1, y1, x2, y2]" in
pixels, query a string or a list literal of strings); one sample = the image resized to cfg.resize_img with PIL, the
query as `phrase_len` = 50 word vectors (the text is padded with ' PD' tokens; `qlens` = number of real tokens), the box
as y1x1y2x2 normalised to [-1, 1]; the collater stacks every field as float and cuts `qvec` to the longest query of the
batch.

What is different, and why:
  * word vectors come from a pluggable embedder: spaCy (`en_core_web_md`, as the reference) when it is installed, else a
    word-vector table file (cfg.word_vectors: .npz with `words` [V] and `vectors` [V, emb_dim]); spaCy is not available
    offline, so the table path is what the tests exercise;
  * with `gpu_normalise` the image travels as uint8 HWC (4x fewer PCIe bytes, pinned memory, copied on a side stream by
    `DevicePrefetcher`) and `/255` + the NHWC4 layout happen in one HIP kernel (`zsg_u8hwc_to_nhwc4`) — bit-identical to
    `pil2tensor(...).float().div_(255)` (dat_loader.py:26-33, 134);
  * dataset names other than the two the reference's `_read_annotations` knows (dat_loader.py:176-184 leaves `trn_df`
    undefined for flickr30k_c0/c1, vg_split_*) work: flickr30k* ids get the '.jpg' suffix, everything else is a path.
"""
import ast
import functools
import math
import re
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

PHRASE_LEN = 50            # dat_loader.py:88
PAD_TOKEN = "PD"           # dat_loader.py:110


# ---------------------------------------------------------------------------------------------------------------------
# word vectors
# ---------------------------------------------------------------------------------------------------------------------
class SpacyEmbedder:
    """The reference's tokeniser + vectors (dat_loader.py:23, 105-115)."""

    def __init__(self, model: str = "en_core_web_md"):
        import spacy                    # raises ImportError when absent: the caller then needs cfg.word_vectors
        self.nlp = spacy.load(model)

    def tokens(self, text: str) -> List[str]:
        return [t.text for t in self.nlp(text)]

    def vectors(self, text: str) -> np.ndarray:
        return np.array([t.vector for t in self.nlp(text)], dtype=np.float32)


class TableEmbedder:
    """Word-vector table: tokens are runs of word characters or single punctuation marks; unknown words (and the pad
    token, unless the table has it) map to the zero vector, as spaCy does for out-of-vocabulary tokens."""
    _tok = re.compile(r"\w+|[^\w\s]", re.UNICODE)

    def __init__(self, path: str, emb_dim: int = 300):
        z = np.load(path, allow_pickle=False)
        self.index = {str(w): i for i, w in enumerate(z["words"])}
        self.table = np.asarray(z["vectors"], dtype=np.float32)
        assert self.table.ndim == 2 and self.table.shape[1] == emb_dim, f"{path}: vectors must be [V, {emb_dim}]"
        self.emb_dim = emb_dim

    def tokens(self, text: str) -> List[str]:
        return self._tok.findall(text)

    def vectors(self, text: str) -> np.ndarray:
        toks = self.tokens(text)
        out = np.zeros((len(toks), self.emb_dim), np.float32)
        for i, t in enumerate(toks):
            j = self.index.get(t, self.index.get(t.lower(), -1))
            if j >= 0:
                out[i] = self.table[j]
        return out


def get_embedder(cfg):
    path = cfg["word_vectors"] if "word_vectors" in cfg else ""
    if path:
        return TableEmbedder(path, int(cfg["emb_dim"]))
    try:
        return SpacyEmbedder()
    except ImportError as e:
        raise RuntimeError("no word vectors: spaCy is not installed and cfg.word_vectors (a .npz with `words`, `vectors`) is empty") from e


def embed_query(embedder, query: str, phrase_len: int = PHRASE_LEN) -> Tuple[np.ndarray, int]:
    """dat_loader.py:104-115: -> ([phrase_len, emb] float32, number of real tokens)."""
    query = query.strip()
    qlen = len(embedder.tokens(query))
    if qlen == 0:
        raise NotImplementedError("empty query")               # as the reference (dat_loader.py:106-108)
    vecs = embedder.vectors(query + (" " + PAD_TOKEN) * (phrase_len - qlen))[:phrase_len]
    assert vecs.shape[0] == phrase_len, "tokenisation of the padded query is not stable"
    return vecs, qlen


# ---------------------------------------------------------------------------------------------------------------------
# GPU-side resize: PIL.Image.resize's default filter reproduced bit for bit on uint8
# ---------------------------------------------------------------------------------------------------------------------
_RESIZE_PRECISION_BITS = 32 - 8 - 2          # Pillow's Resample.c: 8-bit channels are filtered in 22-bit fixed point


def _bicubic(x: float) -> float:
    """Pillow's bicubic kernel (a = -0.5), evaluated in double precision in ITS operation order"""
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


@functools.lru_cache(maxsize=4096)
def resize_tables(in_size: int, out_size: int):
    """The per-axis tap tables `img.resize((W, H))` — the reference's only resampling call, dat_loader.py:121, PIL's default filter
    (bicubic for RGB) — uses for one axis: Pillow's precompute_coeffs + normalize_coeffs_8bpc restated (double-precision weights
    normalised to sum 1, then rounded to 22-bit fixed point).  Returns (bounds int32 [out, 2] = first tap, tap count; coefficients
    int32 [out, ksize]; ksize).  The HIP kernel zsg_resize_u8 accumulates these integers exactly as Pillow does, so its uint8 output
    is bit-identical to Pillow's (tests/test_cpu_loader.py pins the tables against Pillow itself, tests/test_gpu_trainer.py the kernel)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, xmax)
        for x, w in enumerate(k):
            v = w * (1 << _RESIZE_PRECISION_BITS)
            coef[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
    return bounds, coef, ksize


def flatten_raw(imgs) -> Tuple[torch.Tensor, torch.Tensor]:
    """Raw uint8 [h, w, 3] images of different sizes -> (one flat uint8 tensor, int32 [B, 2] heights / widths): what the collater hands on
    in gpu_resize mode — the DataLoader's pin thread pins ONE tensor and the trainer uploads it with ONE copy (a list of B tensors cost
    the trainer thread a pin_memory() + a copy per image)."""
    hw = torch.tensor([[int(im.shape[0]), int(im.shape[1])] for im in imgs], dtype=torch.int32)
    return torch.cat([im.reshape(-1) for im in imgs]), hw


class GpuResizer:
    """Resizes raw uint8 HWC images of ANY size to one [B, H, W, 3] uint8 batch on the GPU (Pillow's two-pass fixed-point bicubic, bit-
    identical to PIL.Image.resize's default filter, dat_loader.py:121).  The tap tables of an axis length are computed once on the host
    and kept on the device.  Round 5: the whole batch is resized by TWO launches (zsg_resize_u8_batched) from one job table — the
    per-image form (one call, two launches and one upload per image) capped the consumer at 3 680 img/s with sixteen workers."""

    def __init__(self, out_hw, device="cuda"):
        self.Ho, self.Wo, self.dev = int(out_hw[0]), int(out_hw[1]), device
        self._tab = {}
        self._tmp = None
        self._jobs_host = None
        import PIL
        if tuple(int(v) for v in PIL.__version__.split(".")[:2]) < (7, 0):      # (Pillow < 7 resizes with NEAREST by default: the two paths would differ)
            raise RuntimeError("GpuResizer reproduces PIL.Image.resize's default filter of Pillow >= 7 (bicubic); found Pillow " + PIL.__version__)

    def _tables(self, n_in, n_out, identity_ok=False):
        key = (n_in, n_out)
        if key not in self._tab:
            if n_in == n_out and not identity_ok:
                self._tab[key] = None                    # Pillow skips a pass whose size does not change
            elif n_in == n_out:
                # the batched launches run both passes for every image: an unchanged axis gets the identity table (one tap, 2^22)
                b = np.stack([np.arange(n_out, dtype=np.int32), np.ones(n_out, np.int32)], 1)
                c = np.full((n_out, 1), 1 << _RESIZE_PRECISION_BITS, np.int32)
                self._tab[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(c).to(self.dev), 1)
            else:
                b, c, ks = resize_tables(n_in, n_out)
                self._tab[key] = (torch.from_numpy(b).to(self.dev), torch.from_numpy(c).to(self.dev), ks)
        return self._tab[key]

    def resize_flat(self, flat: torch.Tensor, hw, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """flat: uint8 DEVICE tensor holding the raw images back to back; hw: [B, 2] (host) heights / widths.  Returns uint8 [B, Ho, Wo, 3]."""
        return self._run([(flat.data_ptr() + off, h, w) for off, h, w in self._offsets(hw)], out, keep=flat)

    @staticmethod
    def _offsets(hw):
        off, res = 0, []
        for h, w in (hw.tolist() if torch.is_tensor(hw) else hw):
            res.append((off, int(h), int(w)))
            off += int(h) * int(w) * 3
        return res

    def _run(self, items, out, keep=None):
        import struct
        from ._lib import check, lib, stream_ptr
        B = len(items)
        if out is None:
            out = torch.empty(B, self.Ho, self.Wo, 3, dtype=torch.uint8, device=self.dev)
        need = sum(h for _, h, _ in items) * self.Wo * 3
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.uint8, device=self.dev)
        blob, tmp_off, bx, by = b"", 0, 0, 0
        per_out = self.Ho * self.Wo * 3
        for i, (ptr, h, w) in enumerate(items):
            tx, ty = self._tables(w, self.Wo, True), self._tables(h, self.Ho, True)
            blob += struct.pack("<qqqqqqqiiiiiiii", ptr, self._tmp.data_ptr() + tmp_off, out.data_ptr() + i * per_out,
                                tx[0].data_ptr(), tx[1].data_ptr(), ty[0].data_ptr(), ty[1].data_ptr(), h, w, tx[2], ty[2], bx, by, 0, 0)
            tmp_off += h * self.Wo * 3
            bx += (h * self.Wo * 3 + 1023) // 1024
            by += (per_out + 1023) // 1024
        # job table: pinned host slot -> device, asynchronously; four slots in rotation, each guarded by the event of its last upload
        # (a slot is rewritten only when that copy has completed: normally four batches ago)
        if self._jobs_host is None or self._jobs_host[0].numel() < len(blob):
            n = max(len(blob), 88 * 256)
            self._jobs_host = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
            self._jobs_dev = [torch.empty(n, dtype=torch.uint8, device=self.dev) for _ in range(4)]
            self._jobs_ev = [None] * 4
            self._slot = 0
        k = self._slot = (self._slot + 1) % 4
        if self._jobs_ev[k] is not None:
            self._jobs_ev[k].synchronize()
        self._jobs_host[k][:len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
        self._jobs_dev[k][:len(blob)].copy_(self._jobs_host[k][:len(blob)], non_blocking=True)
        self._jobs_ev[k] = torch.cuda.Event()
        self._jobs_ev[k].record()
        check(lib.zsg_resize_u8_batched(self._jobs_dev[k].data_ptr(), B, 3, self.Ho, self.Wo, bx, by, stream_ptr()), "zsg_resize_u8_batched")
        return out

    def __call__(self, imgs, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """imgs: list of uint8 [h, w, 3] DEVICE tensors; returns uint8 [B, Ho, Wo, 3]"""
        for im in imgs:
            assert im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3 and im.is_contiguous()
        return self._run([(im.data_ptr(), int(im.shape[0]), int(im.shape[1])) for im in imgs], out, keep=imgs)


# ---------------------------------------------------------------------------------------------------------------------
# dataset + collater
# ---------------------------------------------------------------------------------------------------------------------
class ImgQuDataset(Dataset):
    """Any grounding dataset given as a CSV of (img_id, bbox, query) rows; the same image may appear on many rows."""

    def __init__(self, cfg, csv_file, ds_name: str, split_type: str = "train", embedder=None, gpu_normalise: bool = False,
                 gpu_resize: bool = False):
        """gpu_normalise: the item's image stays uint8 HWC (converted on the GPU); gpu_resize (implies it): the image is NOT resized by
        the worker either — it travels at its decoded size and DevicePrefetcher resizes it on the GPU (GpuResizer: Pillow's filter
        bit for bit), which leaves a worker only the JPEG decode."""
        import pandas as pd
        gpu_normalise = gpu_normalise or gpu_resize
        self.cfg, self.ds_name, self.split_type, self.gpu_normalise, self.gpu_resize = cfg, ds_name, split_type, gpu_normalise, gpu_resize
        self.embedder = embedder if embedder is not None else get_embedder(cfg)
        self.img_dir = Path(cfg["ds_info"][ds_name]["img_dir"])
        self.phrase_len = PHRASE_LEN
        df = pd.read_csv(csv_file)
        self.boxes = [ast.literal_eval(b) if isinstance(b, str) else list(b) for b in df["bbox"]]
        first = str(df["query"].iloc[0])
        self.queries = [ast.literal_eval(q) for q in df["query"]] if first[:1] == "[" else [str(q) for q in df["query"]]
        ids = [str(i) for i in df["img_id"]]
        self.files = [f"{i}.jpg" for i in ids] if ds_name.startswith("flickr30k") else ids        # dat_loader.py:176-184

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        import PIL.Image
        q = self.queries[idx]
        if isinstance(q, list):
            q = str(np.random.choice(q))                          # dat_loader.py:153-154
        q = q.replace("_", " ")
        img = PIL.Image.open(self.img_dir / self.files[idx]).convert("RGB")
        h, w = img.height, img.width
        qvec, qlen = embed_query(self.embedder, q, self.phrase_len)
        x1, y1, x2, y2 = self.boxes[idx]
        rs = self.cfg["resize_img"]
        if not self.gpu_resize:
            img = img.resize((rs[0], rs[1]))                      # PIL's default filter, as the reference (dat_loader.py:121)
        target = 2 * np.array([y1 / h, x1 / w, y2 / h, x2 / w]) - 1          # y1x1y2x2 in [-1, 1] (anchors are row, column)
        a = np.asarray(img)                                       # [H, W, 3] uint8
        if self.gpu_normalise:
            img_t = torch.from_numpy(a.copy())                    # normalised on the GPU (zsg_u8hwc_to_nhwc4)
        else:
            img_t = torch.from_numpy(a.transpose(2, 0, 1).astype(np.float64)).float().div_(255)     # pil2tensor(...).float().div_(255)
        return {"img": img_t, "idxs": torch.tensor(idx).long(), "qvec": torch.from_numpy(qvec),
                "qlens": torch.tensor(min(qlen, self.phrase_len)), "annot": torch.from_numpy(target).float(),
                "orig_annot": torch.tensor([x1, y1, x2, y2]).float(), "img_size": torch.tensor([h, w])}


def collater(batch: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """dat_loader.py:187-196: every field stacked as float (uint8 images stay uint8: they become float on the GPU);
    qvec cut to the longest query of the batch."""
    max_qlen = int(max(int(b["qlens"]) for b in batch))
    out = {}
    for k in batch[0]:
        if k == "img" and batch[0][k].dtype == torch.uint8 and len({tuple(b[k].shape) for b in batch}) > 1:
            # raw images of different sizes (gpu_resize): ONE flat tensor + the sizes; DevicePrefetcher uploads it with one copy and
            # resizes the batch on the GPU with two launches
            out["img"], out["img_hw"] = flatten_raw([b[k] for b in batch])
            continue
        t = torch.stack([b[k] for b in batch])
        out[k] = t if (k == "img" and t.dtype == torch.uint8) else t.float()
    out["qvec"] = out["qvec"][:, :max_qlen]
    if "img_hw" in out:
        out["img_hw"] = out["img_hw"].int()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# samplers / loaders
# ---------------------------------------------------------------------------------------------------------------------
class NewDistributedSampler(DistributedSampler):
    """DistributedSampler with a shuffle switch, so validation can be sharded too (dat_loader.py:36-65): deterministic
    per-epoch permutation, padded with the head of the list to a multiple of the world size, contiguous rank slices."""

    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        super().__init__(dataset, num_replicas=num_replicas, rank=rank)
        self.shuffle = shuffle

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.epoch)
            indices = torch.randperm(n, generator=g).tolist()
        else:
            indices = list(range(n))
        indices += indices[: (self.total_size - len(indices))]
        off = self.num_samples * self.rank
        return iter(indices[off: off + self.num_samples])


def get_dataloader(cfg, dataset: Dataset, is_train: bool) -> DataLoader:
    """dat_loader.py:208-230 (one process per GPU: per-rank batch = cfg.bs; validation is sharded and shuffled under DDP)."""
    dist_on = bool(cfg["do_dist"])
    if dist_on:
        sampler = NewDistributedSampler(dataset, shuffle=True)
    elif is_train:
        sampler = torch.utils.data.RandomSampler(dataset)
    else:
        sampler = torch.utils.data.SequentialSampler(dataset)
    bs = cfg["bs"] if is_train else (cfg["bsv"] if "bsv" in cfg else cfg["bs"])
    nw = cfg["nw"] if is_train else (cfg["nwv"] if "nwv" in cfg else cfg["nw"])
    return DataLoader(dataset, batch_size=bs, sampler=sampler, drop_last=is_train, num_workers=nw, collate_fn=collater,
                      pin_memory=torch.cuda.is_available(), persistent_workers=nw > 0)


class DevicePrefetcher:
    """Wraps a loader of pinned host batches: the next batch is copied to the GPU on a side stream while the current one
    is being consumed (HIP copy engine overlaps the training step), and handed over with an event wait."""

    def __init__(self, loader, device="cuda", resize_hw=None):
        """resize_hw = (H, W): batches whose "img" is raw uint8 HWC (a list of differently sized images, or a stack at another size)
        are resized on the GPU, on the upload stream (GpuResizer)."""
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.resize_hw = tuple(resize_hw) if resize_hw is not None else None
        self._resizer = None

    def __len__(self):
        return len(self.loader)

    def _upload(self, batch):
        if self.stream is None:
            return batch, None
        with torch.cuda.stream(self.stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in batch.items() if torch.is_tensor(v) and k != "img_hw"}
            img = batch.get("img")
            hw = batch.get("img_hw")
            if isinstance(img, list):              # (callers that still hand over a list of raw images: flattened here, on this thread)
                img, hw = flatten_raw(img)
                dev["img"] = img.pin_memory().to(self.device, non_blocking=True)
            elif hw is None and self.resize_hw and torch.is_tensor(img) and img.dtype == torch.uint8 and img.dim() == 4 \
                    and tuple(img.shape[1:3]) != self.resize_hw:
                hw = torch.tensor([[img.shape[1], img.shape[2]]] * img.shape[0], dtype=torch.int32)      # a stack at another size
                dev["img"] = dev["img"].reshape(-1)
            if hw is not None:
                if self.resize_hw is None:
                    raise RuntimeError("DevicePrefetcher: raw images of mixed sizes need resize_hw")
                if self._resizer is None:
                    self._resizer = GpuResizer(self.resize_hw, self.device)
                dev["img"] = self._resizer.resize_flat(dev["img"], hw)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return dev, ev

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            self._epoch = getattr(self, "_epoch", -1) + 1
            sampler.set_epoch(self._epoch)
        nxt = None
        for batch in self.loader:
            cur, nxt = nxt, self._upload(batch)
            if cur is not None:
                yield self._hand_over(cur)
        if nxt is not None:
            yield self._hand_over(nxt)

    def _hand_over(self, item):
        dev, ev = item
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            for v in dev.values():
                v.record_stream(torch.cuda.current_stream())
        return dev


def get_data(cfg, embedder=None, prefetch: Optional[bool] = None):
    """dat_loader.py:233-257: train / valid / test loaders of cfg.ds_to_use (paths from cfg.ds_info)."""
    from .synth import DataWrap
    ds_name = cfg["ds_to_use"]
    info = cfg["ds_info"][ds_name]
    emb = embedder if embedder is not None else get_embedder(cfg)
    gpu = torch.cuda.is_available() and (cfg["gpu_img_normalise"] if "gpu_img_normalise" in cfg else True)
    prefetch = gpu if prefetch is None else prefetch
    gpu_rs = bool(gpu and prefetch and (cfg["gpu_img_resize"] if "gpu_img_resize" in cfg else True))      # resize on the GPU too (N2)
    rs = cfg["resize_img"]

    def make(csv_key, split, is_train):
        ds = ImgQuDataset(cfg, info[csv_key], ds_name, split, emb, gpu_normalise=gpu, gpu_resize=gpu_rs)
        dl = get_dataloader(cfg, ds, is_train)
        return DevicePrefetcher(dl, cfg["device"], resize_hw=(rs[1], rs[0]) if gpu_rs else None) if prefetch else dl
    return DataWrap(make("trn_csv_file", "train", True), make("val_csv_file", "valid", False),
                    {"test0": make("test_csv_file", "valid", False)}, cfg["tmp_path"])
