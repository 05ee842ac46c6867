"""Flat parameter / gradient / buffer storage with reference-compatible names.

All learnable tensors of the network live in ONE fp32 buffer (and their gradients in one more): the fused Adam step
and the bucketed RCCL all-reduce operate on contiguous slices, and the HIP kernels get raw pointers into it.
Convolution weights are STORED in the layout the MFMA kernels consume — OHWI with the input-channel count padded to a
multiple of 4 — and EXPOSED as `nn.Parameter`s of the reference's OIHW shape through a strided (permuted, narrowed)
view, so `state_dict()` / `load_state_dict()` / `torch.optim` see exactly the reference's tensors and key names
(`backbone.encoder.layer1.0.conv1.weight`, ...) with zero copies.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn


def pad4(n: int) -> int:
    return (n + 3) // 4 * 4


@dataclass
class Entry:
    name: str
    kind: str                      # 'conv' | 'vec' | 'mat'
    shape: Tuple[int, ...]         # logical (reference) shape
    offset: int                    # element offset in the flat buffer
    size: int                      # elements reserved (incl. padding)
    cpad: int = 0                  # conv: padded input channels


class ParamStore:
    def __init__(self):
        self.entries: Dict[str, Entry] = {}
        self.order: List[str] = []
        self.total = 0
        self.flat: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None

    def _add(self, name, kind, shape, size, cpad=0) -> Entry:
        assert name not in self.entries, name
        e = Entry(name, kind, tuple(shape), self.total, size, cpad)
        self.entries[name] = e
        self.order.append(name)
        self.total += pad4(size)
        return e

    def add_conv(self, name, co, ci, k) -> Entry:
        cp = pad4(ci)
        return self._add(name, "conv", (co, ci, k, k), co * k * k * cp, cp)

    def add_vec(self, name, n) -> Entry:
        return self._add(name, "vec", (n,), n)

    def add_mat(self, name, rows, cols) -> Entry:
        assert cols % 4 == 0, f"{name}: inner dimension {cols} must be a multiple of 4"
        return self._add(name, "mat", (rows, cols), rows * cols)

    def allocate(self, device):
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)

    def view(self, name: str, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference-shaped (strided) view of one parameter inside `flat` (default: the value buffer)."""
        e = self.entries[name]
        flat = self.flat if flat is None else flat
        raw = flat[e.offset:e.offset + e.size]
        if e.kind == "conv":
            co, ci, k, _ = e.shape
            return raw.view(co, k, k, e.cpad)[..., :ci].permute(0, 3, 1, 2)
        return raw.view(*e.shape)

    def raw(self, name: str, flat: Optional[torch.Tensor] = None) -> torch.Tensor:
        e = self.entries[name]
        flat = self.flat if flat is None else flat
        return flat[e.offset:e.offset + e.size]


class NamedTree(nn.Module):
    """Plain container used to reproduce dotted reference names (e.g. 'backbone.encoder.layer1.0.bn1')."""

    def get_or_create(self, path: List[str]) -> "NamedTree":
        m = self
        for p in path:
            if p not in m._modules:
                m.add_module(p, NamedTree())
            m = m._modules[p]
        return m


def register_named(root: nn.Module, dotted: str, tensor: torch.Tensor, buffer: bool = False):
    *path, leaf = dotted.split(".")
    m = root
    for p in path:
        if p not in m._modules:
            m.add_module(p, NamedTree())
        m = m._modules[p]
    if buffer:
        m.register_buffer(leaf, tensor)
    else:
        m.register_parameter(leaf, nn.Parameter(tensor, requires_grad=True))
    return m, leaf
