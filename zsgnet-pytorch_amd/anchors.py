"""Anchor boxes for the MI355X ZSGNet path (reference `code/anchors.py`).

Anchor generation is init-time host work: the grid / anchor tables are computed once in numpy with the
reference's float64 formula, rounded once to float32 and uploaded; every per-step box operation (IoU, matching, box
encode/decode) lives in the fused HIP kernels of csrc/loss.hip.
"""
from typing import Sequence, Tuple

import numpy as np
import torch

from ._lib import lib, check, stream_ptr


def _linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """torch.linspace's float32 rule (anchors.py:54-55): step in fp32, symmetric halves, fused multiply-add."""
    start, end = np.float32(start), np.float32(end)
    if steps == 1:
        return np.array([start], dtype=np.float32)
    step = np.float32((end - start) / np.float32(steps - 1))
    i = np.arange(steps)
    lo = (np.float64(start) + np.float64(step) * i).astype(np.float32)
    hi = (np.float64(end) - np.float64(step) * (steps - 1 - i)).astype(np.float32)
    return np.where(i < steps // 2, lo, hi).astype(np.float32)


def create_grid_np(h: int, w: int) -> np.ndarray:
    """[h*w, 2] float32 cell centres (y, x) in (-1, 1); a 1-long axis sits at 0 (anchors.py:47-63)."""
    xs = _linspace_f32(-1 + 1 / w, 1 - 1 / w, w) if w > 1 else np.zeros(1, np.float32)
    ys = _linspace_f32(-1 + 1 / h, 1 - 1 / h, h) if h > 1 else np.zeros(1, np.float32)
    g = np.empty((h, w, 2), dtype=np.float32)
    g[:, :, 0] = ys[:, None]
    g[:, :, 1] = xs[None, :]
    return g.reshape(-1, 2)


def create_grid(size, flatten=True):
    "Create a grid of a given `size` (reference anchors.py:47)."
    h, w = size if isinstance(size, tuple) else (size, size)
    g = torch.from_numpy(create_grid_np(int(h), int(w)))
    return g if flatten else g.view(int(h), int(w), 2)


def create_anchors_np(sizes: Sequence[Tuple[int, int]], ratios, scales) -> np.ndarray:
    """float64 [A,4] (y1,x1,y2,x2): anchors.py:66-87.  Index of an anchor inside a level is (y*w + x)*n + a with
    a = ratio_idx*len(scales) + scale_idx."""
    aspects = np.array([[s * np.sqrt(r), s * np.sqrt(1 / r)] for r in ratios for s in scales], dtype=np.float64).reshape(-1, 2)
    out = []
    for h, w in sizes:
        h, w = int(h), int(w)
        lvl = np.array([2 / h, 2 / w], dtype=np.float32).astype(np.float64)      # a float32 tensor in the reference
        sized = aspects * lvl
        ctr = create_grid_np(h, w).astype(np.float64)
        n, a = ctr.shape[0], sized.shape[0]
        c = np.broadcast_to(ctr[:, None, :], (n, a, 2))
        s = np.broadcast_to(sized[None, :, :], (n, a, 2))
        out.append(np.concatenate([c - s / 2, c + s / 2], axis=2).reshape(-1, 4))
    return np.concatenate(out, axis=0)


def create_anchors(sizes, ratios, scales, flatten=True, device=torch.device("cuda")):
    "Create anchor of `sizes`, `ratios` and `scales` (reference anchors.py:66); float32 [A,4] tlbr on `device`."
    if isinstance(sizes, torch.Tensor):
        sizes = [tuple(int(v) for v in r) for r in sizes.tolist()]
    assert flatten, "only the flattened form is used on the hot path"
    a = create_anchors_np(sizes, ratios, scales).astype(np.float32)
    return torch.from_numpy(a).to(device)


def IoU_values(anchors: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """IoU [len(anchors), len(targets)] of tlbr boxes with the reference's fp32 operation order (anchors.py:106-116),
    evaluated by the HIP kernel zsg_iou."""
    a = anchors.contiguous().float()
    t = targets.contiguous().float()
    out = torch.empty(a.shape[0], t.shape[0], device=a.device, dtype=torch.float32)
    check(lib.zsg_iou(a.data_ptr(), t.data_ptr(), a.shape[0], t.shape[0], out.data_ptr(), stream_ptr()), "zsg_iou")
    return out
