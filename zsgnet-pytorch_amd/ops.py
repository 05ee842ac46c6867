"""Host-side op layer: descriptor builders for the C ABI and a static launch "program".

A `Program` is a flat list of (C function, pre-marshalled ctypes arguments): buffers are allocated once per input
geometry, so one training step is just a loop of foreign calls on torch's current stream — no Python tensor ops, no
host<->device synchronisation — and the whole list can be captured into a hipGraph.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from ._lib import ConvDesc, Seg, Taps, lib, check, ZSG_MAX_SEG, ZsgError


@dataclass
class Level:
    off: int          # element offset of image 0 inside the buffer
    H: int
    W: int
    bstride: int      # elements between consecutive images


@dataclass
class TView:
    """An NHWC activation (or a pyramid of them sharing one buffer): element (l,b,y,x,c) at
    buf + levels[l].off + b*levels[l].bstride + (y*W + x)*ld + c."""
    buf: torch.Tensor
    B: int
    C: int
    ld: int
    levels: List[Level]

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    def level(self, i: int) -> "TView":
        return TView(self.buf, self.B, self.C, self.ld, [self.levels[i]])

    def rows(self) -> int:
        return sum(self.B * l.H * l.W for l in self.levels)

    def tensor(self, i: int = 0) -> torch.Tensor:
        """[B,H,W,C] torch view of level i (debug / tests)."""
        l = self.levels[i]
        flat = self.buf.view(-1)
        return torch.as_strided(flat, (self.B, l.H, l.W, self.C), (l.bstride, l.W * self.ld, self.ld, 1), l.off)


def conv_out(n: int, k: int, s: int, p: int, d: int = 1) -> int:
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def _taps(n, w0, wstep, d0, dstep) -> Taps:
    return Taps(int(n), int(w0), int(wstep), int(d0), int(dstep))


def fwd_desc(src: TView, out: TView, C_red: int, N: int, k: int, stride: int, pad: int, dil: int, wC: int,
             wt_ld: Optional[int] = None, wc0: int = 0, relu: bool = False, merge_x: bool = False,
             tile_hint: int = 0) -> ConvDesc:
    """Forward convolution (also the descriptor zsg_conv_wgrad takes, with `out` = the dY view)."""
    d = ConvDesc()
    d.B, d.C, d.N = src.B, C_red, N
    d.src_ld, d.out_ld = src.ld, out.ld
    d.wR = d.wS = k
    d.wC, d.wc0 = wC, wc0
    d.wt_ld = wt_ld if wt_ld is not None else k * k * wC
    d.relu, d.merge_x, d.tile_hint = int(relu), int(merge_x), tile_hint
    assert len(src.levels) == len(out.levels) <= ZSG_MAX_SEG
    d.nseg = len(src.levels)
    for i, (ls, lo) in enumerate(zip(src.levels, out.levels)):
        assert lo.H == conv_out(ls.H, k, stride, pad, dil) and lo.W == conv_out(ls.W, k, stride, pad, dil), \
            f"conv geometry mismatch {ls.H}x{ls.W} -> {lo.H}x{lo.W} (k={k},s={stride},p={pad},d={dil})"
        s = d.seg[i]
        s.rows_y, s.rows_x = lo.H, lo.W
        s.src_H, s.src_W = ls.H, ls.W
        s.sy = s.sx = stride
        s.out_W = lo.W
        s.osy = s.osx = 1
        s.opy = s.opx = 0
        s.src_off, s.src_bstride = ls.off, ls.bstride
        s.out_off, s.out_bstride = lo.off, lo.bstride
        s.ty = _taps(k, 0, 1, -pad, dil)
        s.tx = _taps(k, 0, 1, -pad, dil)
    return d


def dgrad_desc(dy: TView, dx: TView, C_red: int, N: int, k: int, stride: int, pad: int, dil: int,
               tile_hint: int = 0) -> ConvDesc:
    """Data gradient as an implicit GEMM over the transposed weight image WT[cin][k][k][C_red]:
    rows = dx pixels, reduction = dy channels (C_red = dy.ld incl. zero padding), N = forward input channels.
    Strided convolutions are split into stride^2 parity classes (one segment each) so that every K tile carries only
    taps that really contribute (no multiply-by-zero work)."""
    d = ConvDesc()
    d.B, d.C, d.N = dy.B, C_red, N
    d.src_ld, d.out_ld = dy.ld, dx.ld
    d.wR = d.wS = k
    d.wC, d.wc0 = C_red, 0
    d.wt_ld = k * k * C_red
    d.relu, d.merge_x, d.tile_hint = 0, 0, tile_hint
    assert len(dy.levels) == len(dx.levels)
    segs = []
    for ly, lx in zip(dy.levels, dx.levels):
        assert ly.H == conv_out(lx.H, k, stride, pad, dil) and ly.W == conv_out(lx.W, k, stride, pad, dil)
        if stride == 1:
            segs.append((lx.H, lx.W, ly, lx, 1, 0, 0, _taps(k, 0, 1, pad, -dil), _taps(k, 0, 1, pad, -dil)))
            continue
        assert dil == 1, "strided + dilated dgrad is not needed by any supported model"
        for py in range(stride):
            ry = (lx.H - py + stride - 1) // stride
            if ry <= 0:
                continue
            r0 = (py + pad) % stride
            ny = len(range(r0, k, stride))
            ty = _taps(ny, r0, stride, (py + pad - r0) // stride, -1)
            for px in range(stride):
                rx = (lx.W - px + stride - 1) // stride
                if rx <= 0:
                    continue
                c0 = (px + pad) % stride
                nx = len(range(c0, k, stride))
                tx = _taps(nx, c0, stride, (px + pad - c0) // stride, -1)
                segs.append((ry, rx, ly, lx, stride, py, px, ty, tx))
    assert len(segs) <= ZSG_MAX_SEG, "too many dgrad segments"
    d.nseg = len(segs)
    for i, (ry, rx, ly, lx, os_, py, px, ty, tx) in enumerate(segs):
        s = d.seg[i]
        s.rows_y, s.rows_x = ry, rx
        s.src_H, s.src_W = ly.H, ly.W
        s.sy = s.sx = 1
        s.out_W = lx.W
        s.osy = s.osx = os_
        s.opy, s.opx = py, px
        s.src_off, s.src_bstride = ly.off, ly.bstride
        s.out_off, s.out_bstride = lx.off, lx.bstride
        s.ty, s.tx = ty, tx
    return d


class Program:
    """A static list of foreign calls.  `add(fn, *args)` marshals once; `run(stream)` replays."""

    def __init__(self, name: str = ""):
        self.name = name
        self.calls = []
        self.keep = []          # ctypes structs / tensors that must outlive the program

    def add(self, fn, *args, what: str = ""):
        conv = []
        for a, t in zip(args, fn.argtypes[:-1]):
            if isinstance(a, C.Structure):
                self.keep.append(a)
                conv.append(C.byref(a))
            elif isinstance(a, torch.Tensor):
                self.keep.append(a)
                conv.append(t(a.data_ptr()))
            elif a is None:
                conv.append(None)
            else:
                conv.append(t(a))
        assert len(args) == len(fn.argtypes) - 1, f"{fn.__name__}: {len(args)} args for {len(fn.argtypes) - 1}"
        self.calls.append((fn, tuple(conv), what or fn.__name__))

    def run(self, stream: int, start: int = 0, stop: Optional[int] = None):
        st = C.c_void_p(stream)
        calls = self.calls if (start == 0 and stop is None) else self.calls[start:stop]
        if os.environ.get("ZSG_DEBUG_SYNC"):          # locate a faulting launch: name it, run it, synchronise
            for fn, args, what in calls:
                print(f"[zsg] {self.name}/{what}", flush=True)
                rc = fn(*args, st)
                if rc:
                    raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")
                torch.cuda.synchronize()
            return
        for fn, args, what in calls:
            rc = fn(*args, st)
            if rc:
                raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")

    def __len__(self):
        return len(self.calls)
