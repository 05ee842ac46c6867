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
    taps that really contribute (no multiply-by-zero work).  Classes without any contributing tap (e.g. three of the
    four classes of a 1x1 stride-2 convolution) are dropped: `desc.zero_fill` tells the caller to clear dx first."""
    d = ConvDesc()
    d.B, d.C, d.N = dy.B, C_red, N
    d.src_ld, d.out_ld = dy.ld, dx.ld
    d.wR = d.wS = k
    d.wC, d.wc0 = C_red, 0
    d.wt_ld = k * k * C_red
    d.relu, d.merge_x, d.tile_hint = 0, 0, tile_hint
    assert len(dy.levels) == len(dx.levels)
    segs = []
    zero_fill = False
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
                if ny == 0 or nx == 0:
                    zero_fill = True
                    continue
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
    d.zero_fill = zero_fill
    return d


def tile_hint(bm: int, bn: int, splits: int = 0, w8: int = 0, k32: int = 0) -> int:
    """BM | BN<<8 | split_k<<16 | (8-wave workgroup)<<24 | (32-pixel K tiles, wgrad only)<<25"""
    return bm | (bn << 8) | (min(splits, 255) << 16) | (w8 << 24) | (k32 << 25)


def marshal(fn, args, keep: Optional[list] = None) -> tuple:
    """Convert (ctypes struct | tensor | None | scalar) arguments to the C types of fn (all but the trailing stream)."""
    assert len(args) == len(fn.argtypes) - 1, f"{fn.__name__}: {len(args)} args for {len(fn.argtypes) - 1}"
    conv = []
    for a, t in zip(args, fn.argtypes[:-1]):
        if isinstance(a, C.Structure):
            conv.append(C.byref(a))
        elif isinstance(a, torch.Tensor):
            conv.append(t(a.data_ptr()))
        elif a is None:
            conv.append(None)
        else:
            conv.append(t(a))
        if keep is not None and isinstance(a, (C.Structure, torch.Tensor)):
            keep.append(a)
    return tuple(conv)


class WinoJobs:
    """Job list of ONE zsg_wino_weights launch: U = G g G^T for every Winograd convolution of a program.
    A job reads source element (n, tap, c) at src + n*row_ld + tap*tap_ld + c (absolute device addresses)."""

    def __init__(self):
        self.jobs, self.blocks, self.keep = [], 0, []
        self.dev_blob = None

    def add(self, src_ptr: int, dst_ptr: int, N: int, Cred: int, row_ld: int, tap_ld: int, flip: bool):
        chunks, npad = (Cred + 7) // 8, (N + 63) // 64 * 64
        self.jobs.append((int(src_ptr), int(dst_ptr), N, Cred, row_ld, tap_ld, int(flip), npad, chunks, self.blocks))
        self.blocks += (chunks * npad * 8 + 255) // 256

    def finish(self, device):
        import struct
        blob = b"".join(struct.pack("<qqiiiiiiii", *j) for j in self.jobs)
        self.dev_blob = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        return self.dev_blob

    def launch(self, stream: int):
        check(lib.zsg_wino_weights(self.dev_blob.data_ptr(), len(self.jobs), self.blocks, C.c_void_p(stream)), "wino_weights")


def wino_ok(k: int, stride: int, pad: int, dil: int) -> bool:
    return k == 3 and stride == 1 and pad == 1 and dil == 1


SIDE_STREAM = os.environ.get("ZSG_SIDE_STREAM", "1") != "0"
HIP_GRAPH = os.environ.get("ZSG_HIP_GRAPH", "0")     # replay launch ranges as hipGraphs once they are warm ("0", "1", or program names "fwd,bwd")
HIP_GRAPH = False if HIP_GRAPH == "0" else (True if HIP_GRAPH == "1" else tuple(HIP_GRAPH.split(",")))
GRAPH_WARMUP = 2                                             # eager replays of a range before it is captured


# Backward: a side-stream (weight-gradient) launch is released only after this many further main-stream convolutions have
# been enqueued.  1 = the weight gradient of a layer starts when that layer's data gradient has finished, i.e. it runs under
# the BatchNorm backward of the next layer down instead of splitting the CUs with the data gradient; 0 = it starts beside its own
# layer's data gradient.  Round 2 (markers on every release, slower weight gradients): 1 won, 15.27 -> 15.12 ms.  Round 4, one
# box each, tools/ab_env.sh (4 interleaved runs): configs[1] 13.64 -> 13.54 ms with 0 (+ the early laterals, mdl.py), ResNet-18
# 7.61 -> 7.46, SSD-VGG B=32 38.32 -> 38.16; ResNet-101 @600^2 B=32 (launches of several hundred us, every one fills the chip)
# 105.2 -> 107.3 with 0.  Round 5: with the main chain's kernels at wave priority 3 (common.h ZSG_MAIN_PRIO) and the faster dense
# 1x1 weight gradients, 1 wins again on configs[1] (two boxes: 13.32 -> 13.24, 13.46 -> 13.40 ms; 2 = 1, 3 slower;
# profiles/r05_release_policy_prio.txt), so every plan uses 1 (Program.side_defer) unless ZSG_SIDE_DEFER says otherwise.
SIDE_DEFER = int(os.environ.get("ZSG_SIDE_DEFER", "-1"))
# ... and only at every n-th main-stream convolution: every release costs the main stream an event record, i.e. a marker packet
# the next kernel has to wait for (~4 us each: doubling the ~70 records of a ResNet-50 backward costs 0.28 ms).  Measured on
# configs[1]: n = 1 / 2 / 3 / 4 / 5 / 6 -> 14.59 / 14.66 / 14.40 / 14.56 / 14.41 / 14.51 ms; SSD-VGG B=32 39.49 -> 39.15 ms;
# ResNet-18 neutral; ResNet-101 @600^2 B=32 (launches of several hundred us: overlap matters, markers do not) 110.4 -> 111.9 ms,
# hence the plan picks 3 for small launches and 1 for large ones unless ZSG_SIDE_BATCH says otherwise (Program.side_batch).
SIDE_BATCH = int(os.environ.get("ZSG_SIDE_BATCH", "0"))


# Cross-stream edges ride on the producing launch's own completion signal (zsg_set_completion_event / zsg_stream_wait_event, zsg.h)
# instead of an event-record marker in the producing stream's queue; ZSG_COMPLETION_EVENTS=0 restores the marker path.
COMPLETION_EVENTS = os.environ.get("ZSG_COMPLETION_EVENTS", "1") != "0"


_MAIN_CONVS = (lib.zsg_conv_igemm, lib.zsg_conv_wino, lib.zsg_conv_igemm_bnb, lib.zsg_conv_wino_bnb, lib.zsg_conv_igemm_bnstat,
               lib.zsg_conv_wino_bnstat, lib.zsg_conv_igemm_bnb_tail, lib.zsg_conv_wino_bnb_tail)


_SIDE = {}


def shared_side_stream():
    """ONE side stream per device for every Program (forward, backward, the backward's weight preparation): ROCm multiplexes HIP
    streams onto 4 hardware queues per process, and two streams that share a queue serialise — with a stream per Program plus a
    preparation stream, RCCL's stream (DDP) pushed the count past four and the 'concurrent' weight gradients queued behind the
    critical chain."""
    dev = torch.cuda.current_device()
    if dev not in _SIDE:
        _SIDE[dev] = make_side_stream()
        ensure_stream_scratch(_SIDE[dev].cuda_stream)
    return _SIDE[dev]


def make_side_stream():
    return torch.cuda.Stream()


# ---- per-stream scratch of the stream-K convolutions (zsg_set_stream_workspace) -----------------------------------------------------
# 16 KB of hand-off flags + one partial accumulator tile per workgroup: 512 workgroups x 128 x 128 floats (the largest stream-K
# candidate the tuner offers) = 32 MB.  One buffer per (device, stream) that ever replays a program, registered on first use.
SCRATCH_BYTES = (16 << 10) + 512 * 128 * 128 * 4
_SCRATCH = {}


_PRIO_APPLIED = set()


def apply_main_priority_env():
    """ZSG_MAIN_PRIO=0|3 (wave priority of the dependent chain's kernels, zsg_set_main_priority) applied to the current device, once —
    when its first launch plan is built (nothing touches the GPU at import time: a DDP rank has not picked its device yet)."""
    v = os.environ.get("ZSG_MAIN_PRIO")
    dev = torch.cuda.current_device()
    if v is None or dev in _PRIO_APPLIED:
        return
    _PRIO_APPLIED.add(dev)
    check(lib.zsg_set_main_priority(int(v)), "set_main_priority")


def ensure_stream_scratch(stream: int):
    """Register this process's scratch buffer for `stream` (a hipStream_t as an integer) with the library, once."""
    key = (torch.cuda.current_device(), int(stream or 0))
    if key not in _SCRATCH:
        buf = torch.zeros(SCRATCH_BYTES // 4, dtype=torch.float32, device="cuda")
        check(lib.zsg_set_stream_workspace(C.c_void_p(stream), buf.data_ptr(), SCRATCH_BYTES), "set_stream_workspace")
        _SCRATCH[key] = buf
    return _SCRATCH[key]


class Program:
    """A static list of foreign calls.  `add(fn, *args)` marshals once; `run(stream)` replays."""

    def __init__(self, name: str = ""):
        self.name = name
        self.calls = []
        self.lanes = []         # 0 = the caller's stream; 1 = side stream (work whose results are not read before the next
                                # join); 2 = the caller's stream after it has waited for the side stream (a join)
        self.keep = []          # ctypes structs / tensors that must outlive the program
        self._side = None
        self.side_batch = 1     # deferred side launches are released at every side_batch-th main-stream convolution
        self.side_defer = 1     # ... and only after this many further main-stream convolutions (backward program; SIDE_DEFER above)
        self._side_busy = False
        self._graphs = {}       # (start, stop, side-stream mode) -> [eager replays so far, captured graph | None]
        self._sched = {}        # (start, stop, join, side stream busy at entry, release policy) -> compiled lane schedule
        self._ce_pool = []      # completion events (hipEvent_t behind the C ABI)

    def add(self, fn, *args, what: str = "", lane: int = 0):
        self.calls.append((fn, marshal(fn, args, self.keep), what or fn.__name__))
        self.lanes.append(lane)

    def _run_lanes(self, stream: int, start: int, stop: int, join: bool = True):
        """Replay with lane-1 launches on the side HIP stream: each one waits for everything enqueued on the main stream
        before its release (its inputs), and the main stream re-joins the side stream at lane-2 launches and — unless
        join=False (a DDP bucket boundary: the collective waits for both streams instead) — at the end of the range.
        Weight-gradient kernels are leaves of the backward graph, so they fill the CUs the critical chain's small launches
        leave idle; in the backward program their release is deferred (SIDE_DEFER) and batched (side_batch): every release
        costs the main stream one event record."""
        main = torch.cuda.current_stream()
        assert main.cuda_stream == stream, "Program.run expects torch's current stream"
        if self._side is None:
            self._side = shared_side_stream()
            self._ev_pool = []
        side = self._side
        st0, st1 = C.c_void_p(stream), C.c_void_p(side.cuda_stream)
        dirty, nev = True, 0
        defer = (SIDE_DEFER if SIDE_DEFER >= 0 else self.side_defer) if self.name == "bwd" else 0
        pending = []            # deferred lane-1 launches: [index, main-stream convolutions still to enqueue before it]
        nconv, batch = 0, max(1, SIDE_BATCH or self.side_batch)

        def side_launch(i):
            nonlocal dirty, nev
            fn, args, what = self.calls[i]
            if dirty:
                if nev == len(self._ev_pool):
                    self._ev_pool.append(torch.cuda.Event())
                ev = self._ev_pool[nev]
                nev += 1
                ev.record(main)
                side.wait_event(ev)
                dirty = False
            self._side_busy = True              # (survives the call: a later range may have to join what this one started)
            rc = fn(*args, st1)
            if rc:
                raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")

        for i in range(start, stop):
            fn, args, what = self.calls[i]
            if self.lanes[i] == 2 and (self._side_busy or pending):
                for j, _ in pending:
                    side_launch(j)
                pending.clear()
                main.wait_stream(side)
                self._side_busy = False
            if self.lanes[i] == 1:
                if defer:
                    pending.append([i, defer])
                else:
                    side_launch(i)
                continue
            dirty = True
            rc = fn(*args, st0)
            if rc:
                raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")
            if pending and fn in _MAIN_CONVS:
                for e in pending:
                    e[1] -= 1
                nconv += 1
                if nconv % batch == 0:
                    while pending and pending[0][1] <= 0:
                        side_launch(pending.pop(0)[0])
        for j, _ in pending:
            side_launch(j)
        if join and self._side_busy:
            main.wait_stream(side)
            self._side_busy = False

    def _schedule(self, start: int, stop: int, join: bool, side_busy: bool):
        """The lane scheduler of _run_lanes as a dry run: a flat list of operations for [start, stop) —
             ('m', i, k)  launch i on the main stream  (k >= 0: the launch carries completion event k)
             ('s', i, k)  launch i on the side stream  (      "      )
             ('ws', k) / ('wm', k)   the side / main stream waits for event k
             ('rm', k) / ('rs', k)   event k recorded on the main / side stream (a marker: only where no launch of this range precedes
                                     the edge)
        Same release policy as _run_lanes (deferral, batching); a release is attached to the last main-stream launch in front of it, a
        join to the last side-stream launch."""
        ops, nev = [], 0
        dirty, last_main, last_side = True, None, None
        defer = (SIDE_DEFER if SIDE_DEFER >= 0 else self.side_defer) if self.name == "bwd" else 0
        pending, nconv, batch = [], 0, max(1, SIDE_BATCH or self.side_batch)

        def side_launch(i):
            nonlocal dirty, nev, last_side, side_busy, last_main
            if dirty:
                k, nev = nev, nev + 1
                if last_main is not None and ops[last_main][2] < 0:
                    ops[last_main] = ("m", ops[last_main][1], k)
                else:
                    ops.append(("rm", k))
                ops.append(("ws", k))
                dirty = False
            ops.append(("s", i, -1))
            last_side, side_busy = len(ops) - 1, True

        def join_side():
            nonlocal nev, last_side, side_busy
            k, nev = nev, nev + 1
            if last_side is not None and ops[last_side][2] < 0:
                ops[last_side] = ("s", ops[last_side][1], k)
            else:
                ops.append(("rs", k))
            ops.append(("wm", k))
            last_side, side_busy = None, False

        for i in range(start, stop):
            fn = self.calls[i][0]
            if self.lanes[i] == 2 and (side_busy or pending):
                for j, _ in pending:
                    side_launch(j)
                pending.clear()
                join_side()
            if self.lanes[i] == 1:
                if defer:
                    pending.append([i, defer])
                else:
                    side_launch(i)
                continue
            dirty = True
            ops.append(("m", i, -1))
            last_main = len(ops) - 1
            if pending and fn in _MAIN_CONVS:
                for e in pending:
                    e[1] -= 1
                nconv += 1
                if nconv % batch == 0:
                    while pending and pending[0][1] <= 0:
                        side_launch(pending.pop(0)[0])
        for j, _ in pending:
            side_launch(j)
        if join and side_busy:
            join_side()
        return ops, nev, side_busy

    def _run_lanes_ce(self, stream: int, start: int, stop: int, join: bool = True):
        """_run_lanes with the cross-stream edges on completion events of the producing launches (no marker packet in the producing
        stream's queue: ~4.3 us of main-stream time per release, ~35 releases per step)."""
        main = torch.cuda.current_stream()
        assert main.cuda_stream == stream, "Program.run expects torch's current stream"
        if self._side is None:
            self._side = shared_side_stream()
            self._ev_pool = []
        key = (start, stop, join, self._side_busy, SIDE_DEFER, self.side_defer, SIDE_BATCH or self.side_batch)
        sched = self._sched.get(key)
        if sched is None:
            sched = self._sched[key] = self._schedule(start, stop, join, self._side_busy)
        ops, nev, busy_out = sched
        while len(self._ce_pool) < nev:
            e = lib.zsg_event_create()
            if not e:
                raise ZsgError(f"event_create failed: {lib.zsg_last_error().decode()}")
            self._ce_pool.append(C.c_void_p(e))
        evs = self._ce_pool
        st0, st1 = C.c_void_p(stream), C.c_void_p(self._side.cuda_stream)
        arm, rec, wait = lib.zsg_set_completion_event, lib.zsg_event_record, lib.zsg_stream_wait_event
        calls = self.calls
        for op in ops:
            t = op[0]
            if t == "m" or t == "s":
                fn, args, what = calls[op[1]]
                st = st0 if t == "m" else st1
                k = op[2]
                if k < 0:
                    rc = fn(*args, st)
                else:
                    arm(evs[k])
                    rc = fn(*args, st)
                    if arm(None) == 0:          # the call launched nothing that could carry the event: a marker instead
                        rec(evs[k], st)
                if rc:
                    arm(None)
                    raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")
            elif t == "ws":
                wait(st1, evs[op[1]])
            elif t == "wm":
                wait(st0, evs[op[1]])
            elif t == "rm":
                rec(evs[op[1]], st0)
            else:
                rec(evs[op[1]], st1)
        self._side_busy = busy_out

    def run(self, stream: int, start: int = 0, stop: Optional[int] = None, graph: bool = True, join: bool = True):
        """Replay calls[start:stop] on `stream` (torch's current stream), eagerly by default (3.4-3.9 us of host time per
        launch: the host stays ahead of the GPU).  With HIP_GRAPH on, a range that has been replayed GRAPH_WARMUP times is
        captured into a hipGraph (both lanes, with their event edges) and launched as ONE graph from then on — measured
        slower than eager replay on ROCm 7.2, hence opt-in."""
        stop = len(self.calls) if stop is None else stop
        if os.environ.get("ZSG_DEBUG_SYNC"):          # locate a faulting launch: name it, run it, synchronise
            st = C.c_void_p(stream)
            for fn, args, what in self.calls[start:stop]:
                print(f"[zsg] {self.name}/{what}", flush=True)
                rc = fn(*args, st)
                if rc:
                    raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")
                torch.cuda.synchronize()
            return
        if not (HIP_GRAPH and graph) or stop - start < 8 or (isinstance(HIP_GRAPH, tuple) and self.name not in HIP_GRAPH):
            return self._run_eager(stream, start, stop, join)
        key = (start, stop, SIDE_STREAM)
        ent = self._graphs.setdefault(key, [0, None])
        if ent[1] is not None:
            ent[1].replay()
            return
        if ent[0] < GRAPH_WARMUP:
            ent[0] += 1
            return self._run_eager(stream, start, stop)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            self._run_eager(torch.cuda.current_stream().cuda_stream, start, stop)
        ent[1] = g
        g.replay()

    def _run_eager(self, stream: int, start: int, stop: int, join: bool = True):
        if SIDE_STREAM and any(self.lanes[start:stop]):
            if COMPLETION_EVENTS and not torch.cuda.is_current_stream_capturing():
                return self._run_lanes_ce(stream, start, stop, join)
            return self._run_lanes(stream, start, stop, join)
        st = C.c_void_p(stream)
        for fn, args, what in self.calls[start:stop]:
            rc = fn(*args, st)
            if rc:
                raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")

    def profile(self, stream: int):
        """[(what, ms)] per launch, timed with HIP events on the launch stream (developer tool, bench --per-op)."""
        st = C.c_void_p(stream)
        evs = []
        for fn, args, what in self.calls:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args, st)
            b.record()
            if rc:
                raise ZsgError(f"{self.name}/{what} failed ({rc}): {lib.zsg_last_error().decode()}")
            evs.append((what, fn.__name__, a, b))
        torch.cuda.synchronize()
        return [(w, f, a.elapsed_time(b)) for w, f, a, b in evs]

    def __len__(self):
        return len(self.calls)


# ---------------------------------------------------------------------------------------------------------------------
# On-device autotuning of the tile shape / split-K factor of a convolution launch ("measure, don't guess").
# ---------------------------------------------------------------------------------------------------------------------
_TUNE_CACHE = {}
_TUNE_DIRTY = False


def load_tune_cache(path: str) -> int:
    """Optional persistence of the autotuner's choices (ZSG_TUNE_CACHE=<file>): profiling runs then contain no tuning
    launches.  Keys are geometry signatures, values tile hints."""
    import ast
    import json
    if not path or not os.path.exists(path):
        return 0
    with open(path) as f:
        for k, v in json.load(f).items():
            _TUNE_CACHE[ast.literal_eval(k)] = int(v)
    return len(_TUNE_CACHE)


def save_tune_cache(path: str):
    import json
    if path and _TUNE_DIRTY:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump({repr(k): v for k, v in _TUNE_CACHE.items()}, f)


if os.environ.get("ZSG_TUNE_CACHE"):
    import atexit
    load_tune_cache(os.environ["ZSG_TUNE_CACHE"])
    atexit.register(lambda: save_tune_cache(os.environ.get("ZSG_TUNE_CACHE", "")))


# ---- the shipped tuning table ------------------------------------------------------------------------------------------------
# Tile choices for the BASELINE.json shapes, made once on an MI355X by tools/make_tuning_table.py (median-of-5 interleaved tuner) and
# committed as zsgnet-pytorch_amd/tuning/gfx950.json together with the sha256 stamp of the kernel sources they were timed on.  A
# process preloads them when the stamp matches the sources it runs (else nothing is loaded and every shape is tuned on first use),
# so two fresh processes lower the same launch programs: the headline number does not depend on the tuner's luck (round 3: fresh
# tunings of one build spanned 13.89-14.21 ms).  Shapes that are not in the table are tuned as before.  ZSG_SHIPPED_TUNE=0 disables.
SHIPPED_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gfx950.json")
TUNE_INFO = {"table": None, "stamp": None, "table_stamp": None, "loaded": 0, "tuned_now": 0}


def files_stamp() -> str:
    """sha256 over the kernel sources lying next to the package (csrc/*.hip, *.h, *.cpp) — what csrc/stamp.py compiles into the library"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "*"))):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def source_stamp() -> str:
    """The stamp of the sources the LOADED libzsg.so was built from (zsg_source_stamp(): the library may be stale against the files,
    or another build selected with ZSG_LIB_PATH — ADVICE r04): what a tuning table / a rocprof summary was measured on."""
    return lib.zsg_source_stamp().decode()


def load_shipped_table(path: str = SHIPPED_TABLE) -> int:
    """Preload the shipped tile choices — only when they were measured on the library that is loaded (its embedded source stamp) and on
    the architecture this process drives (gfx950: the table's name); an entry that does not parse is skipped, never fatal."""
    import ast
    import json
    TUNE_INFO["stamp"] = source_stamp()
    TUNE_INFO["files_stamp"] = files_stamp()          # != stamp: libzsg.so is stale against csrc/ (or ZSG_LIB_PATH selects another build)
    if os.environ.get("ZSG_SHIPPED_TUNE", "1") == "0" or not os.path.exists(path):
        return 0
    try:
        tj = json.load(open(path))
    except Exception:
        return 0
    TUNE_INFO["table"], TUNE_INFO["table_stamp"] = os.path.relpath(path, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), tj.get("source_stamp")
    if tj.get("source_stamp") != TUNE_INFO["stamp"]:
        return 0
    n = 0
    for k, v in tj.get("entries", {}).items():
        try:
            key = ast.literal_eval(k)
            val = int(v)
        except Exception:
            TUNE_INFO["bad_entries"] = TUNE_INFO.get("bad_entries", 0) + 1
            continue
        if key not in _TUNE_CACHE:                 # (an explicit ZSG_TUNE_CACHE wins)
            _TUNE_CACHE[key] = val
            _SHIPPED_KEYS.add(key)
            n += 1
    TUNE_INFO["loaded"] = n
    return n


_SHIPPED_KEYS = set()
_ARCH_CHECKED = False


def _check_table_arch():
    """First tuner call (the device is initialised by then — nothing here touches the GPU at import time, where a DDP rank has not
    picked its device yet): the shipped table was measured on gfx950; on any other architecture its entries are dropped."""
    global _ARCH_CHECKED
    if _ARCH_CHECKED or not torch.cuda.is_available():
        return
    _ARCH_CHECKED = True
    arch = getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "gcnArchName", "")
    TUNE_INFO["device_arch"] = arch
    if _SHIPPED_KEYS and "gfx950" not in arch:
        for k in _SHIPPED_KEYS:
            _TUNE_CACHE.pop(k, None)
        TUNE_INFO["loaded"] = 0
        _SHIPPED_KEYS.clear()


def _sig(kind, d: ConvDesc, extra) -> tuple:
    segs = tuple((d.seg[i].rows_y, d.seg[i].rows_x, d.seg[i].src_H, d.seg[i].src_W, d.seg[i].sy, d.seg[i].osy,
                  d.seg[i].ty.n, d.seg[i].tx.n) for i in range(d.nseg))
    return (kind, d.B, d.C, d.N, d.src_ld, d.out_ld, d.wR, d.wC, d.wt_ld, d.relu, d.merge_x, segs, extra)


load_shipped_table()


def _time_launch(fn, args, stream, reps=6):
    """one timing sample: the mean of `reps` back-to-back launches (inf when the library refuses the configuration)"""
    st = C.c_void_p(stream)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        if fn(*args, st):
            return float("inf")
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


TUNE_ROUNDS = int(os.environ.get("ZSG_TUNE_ROUNDS", "5"))


def _pick_best(trials, set_hint, stream, penalty):
    """Median of TUNE_ROUNDS INTERLEAVED samples per candidate (round r times every surviving candidate once before round r + 1
    starts, so clock / thermal drift and a noisy neighbour hit all candidates alike; one batch of 10 launches per candidate picked
    tiles that differed from process to process and moved the step by +-1 %).  Candidates more than 25 % behind the leader after
    the first round are dropped.  trials: [(fn, marshalled args, hint, flag)]; returns hint | flag of the winner."""
    live, samples = [], []
    for f, conv, h, flag in trials:
        set_hint(h)
        if f(*conv, C.c_void_p(stream)):            # warm-up launch; a refused configuration is not a candidate
            continue
        _time_launch(f, conv, stream, reps=1)
        live.append((f, conv, h, flag))
        samples.append([])
    for r in range(TUNE_ROUNDS):
        for i, (f, conv, h, flag) in enumerate(live):
            if samples[i] is None:
                continue
            set_hint(h)
            samples[i].append(_time_launch(f, conv, stream) + penalty(h))
        if r == 0 and live:
            lead = min(s[0] for s in samples if s)
            for i, s_ in enumerate(samples):
                if s_[0] > 1.25 * lead:
                    samples[i] = None
    best, best_t = 0, float("inf")
    ranked = []
    for (f, conv, h, flag), s_ in zip(live, samples):
        if s_:
            t = sorted(s_)[len(s_) // 2]
            ranked.append((t, h | flag))
            if t < best_t:
                best, best_t = h | flag, t
    _LAST_RANKING[:] = sorted(ranked)
    return best


def _verify_candidates(kind, key, trials, set_hint, out, add_src, stream):
    """ZSG_TUNE_VERIFY=1 (developer check; a candidate that computes something else than its siblings is a bug the timing cannot see):
    every candidate of the launch is run once from the SAME state of the output buffer (split-K candidates of a non-accumulating launch
    from zeros, as the plan's prepared launch would) and its stored output compared with the first candidate's: fp32 summation-order
    differences only (2e-3 of the largest magnitude; a plain candidate starts from a NaN-poisoned buffer, so an element it fails to write
    counts).  Prints one line per outlier and keeps a count in TUNE_INFO['verify_bad']."""
    if not isinstance(out, torch.Tensor):
        return
    st = C.c_void_p(stream)
    snap = out.clone()
    accumulate = add_src is not None and isinstance(add_src, torch.Tensor) and add_src.data_ptr() == out.data_ptr()
    ref, ref_h = None, 0
    region = None              # the elements the launch writes (what the first candidate left non-NaN in a poisoned buffer)
    for f, conv, h, flag in trials:
        set_hint(h)
        split = kind == "igemm" and ((h >> 16) & 0xff) > 1
        if accumulate:
            out.copy_(snap)                         # the launch adds to what is there: every candidate from the same state
        elif split:
            out.zero_()                             # atomic split-K accumulates into a prepared (zeroed) output
        else:
            out.fill_(float("nan"))                 # a plain launch must WRITE every element of its region
        if f(*conv, st):
            continue
        torch.cuda.synchronize()
        res = out.clone()
        if ref is None:
            if split:
                continue                            # (the reference is the first plain candidate)
            ref, ref_h = res, h | flag
            region = ~torch.isnan(ref) if not accumulate else torch.ones_like(ref, dtype=torch.bool)
            continue
        a, b = res[region], ref[region]
        scale = float(b.abs().max()) + 1e-30
        err = float((a - b).abs().max()) if a.numel() else 0.0
        bad = not (err <= float(os.environ.get("ZSG_TUNE_VERIFY_TOL", "2e-3")) * scale)          # (NaN: not <=)
        TUNE_INFO["verify_n"] = TUNE_INFO.get("verify_n", 0) + 1
        if bad:
            TUNE_INFO["verify_bad"] = TUNE_INFO.get("verify_bad", 0) + 1
            if TUNE_INFO["verify_bad"] <= 20:
                print(f"[zsg tune-verify] {kind} {key[1:6]} hint {hex(h | flag)} vs {hex(ref_h)}: max |diff| {err:.3e} (scale {scale:.3e})", flush=True)
    out.copy_(snap)


_LAST_RANKING = []           # [(median ms incl. penalty, hint | flag)] of the last _pick_best call, fastest first
_TUNE_ALTS = {}              # cache key -> that ranking, for the shapes THIS process tuned: what refine_in_step() may try inside the step


WINO_FLAG = 1 << 30          # tuner result: the Winograd kernel won (its own tile hint in the low bits)


def wino_mode() -> str:
    """ZSG_WINO: '1' (default) the autotuner times the Winograd kernel next to the direct one for every 3x3/s1 convolution
    and keeps the faster; '0' direct only; 'force' Winograd wherever it applies (parity tests of that path)."""
    return os.environ.get("ZSG_WINO", "1")


K64_FLAG = 1 << 27           # tile_hint bit (igemm): 64-deep K tiles — half the K steps (barriers) at twice the LDS per block
SK_SHIFT = 28                # tile_hint bits 28-29 (igemm): stream-K with that many workgroups per CU (csrc/igemm.hip, template flag SK)


def sk_cands(d: ConvDesc, rows: int) -> list:
    """Stream-K candidates of an implicit-GEMM launch whose tile grid is below one round of (256 x workgroups-per-CU) workgroups —
    layer3 / layer4's 1x1 convolutions, the strided 3x3 ones, the small pyramid levels.  The library refuses what it cannot run
    (more tiles than workgroups, several segments): a refused candidate is simply not timed."""
    if d.nseg != 1 or d.merge_x or os.environ.get("ZSG_SK", "1") == "0" or os.environ.get("ZSG_SK_IGEMM", "1") == "0" or HIP_GRAPH:
        return []          # (hipGraph replay, opt-in: a captured range runs on the capture stream, which has no registered scratch)
    out = []
    for bm, bn, w8, k64 in ((64, 64, 0, 0), (64, 64, 1, 0), (64, 64, 1, 1), (128, 64, 1, 0), (128, 128, 1, 0), (128, 128, 0, 0)):
        if (bn == 128 and d.N <= 64) or (k64 and d.C % 64):
            continue
        tiles = ((rows + bm - 1) // bm) * ((d.N + bn - 1) // bn)
        for bpc in ((1, 2, 3) if bm == 64 else (1, 2)):
            if tiles <= 256 * bpc and not (tiles <= 256 * (bpc - 1) and bpc > 2):
                out.append(tile_hint(bm, bn, 1, w8) | (K64_FLAG if k64 else 0) | (bpc << SK_SHIFT))
    return out


def deterministic() -> bool:
    """ZSG_DETERMINISTIC=1: no launch may combine partial results with fp32 atomics (split-K candidates are not offered),
    so two processes that use the same tile choices (ZSG_TUNE_CACHE) produce bit-identical results."""
    return os.environ.get("ZSG_DETERMINISTIC", "0") == "1"


def _wino_cands(d: ConvDesc, allow_sk: bool = True) -> list:
    tiles = sum(d.B * ((d.seg[i].src_H + 1) // 2) * ((d.seg[i].src_W + 1) // 2) for i in range(d.nseg))
    # (tiles per block, channels per block, split-K, four position groups instead of two = twice the waves per SIMD)
    cands = [tile_hint(64, 64, 1), tile_hint(32, 64, 1), tile_hint(64, 64, 1, 1), tile_hint(32, 64, 1, 1), tile_hint(32, 32, 1, 1)]
    # stream-K (csrc/wino.hip, template flag SK): the 32 x 64 four-group tile over 256 workgroups, for grids below one round
    if allow_sk and d.nseg == 1 and ((tiles + 31) // 32) * ((d.N + 63) // 64) <= 256 and os.environ.get("ZSG_SK", "1") != "0" and os.environ.get("ZSG_SK_WINO", "1") != "0" and not HIP_GRAPH:
        cands.append(tile_hint(32, 64, 1, 1) | (1 << SK_SHIFT))
    s0 = d.seg[0]
    dense = (d.nseg == 1 and not d.relu and d.out_ld == d.N and s0.out_bstride == s0.rows_y * s0.rows_x * d.N)
    if dense and not deterministic():
        for tb, bn in ((64, 64), (32, 64)):
            blocks = ((tiles + tb - 1) // tb) * ((d.N + bn - 1) // bn)
            for sp in (2, 4, 8):
                if blocks * sp <= 1024 and sp * 4 <= (d.C + 7) // 8:
                    cands.append(tile_hint(tb, bn, sp))
                    cands.append(tile_hint(tb, bn, sp, 1))
    return cands


def wino_default_hint(d: ConvDesc) -> int:
    tiles = sum(d.B * ((d.seg[i].src_H + 1) // 2) * ((d.seg[i].src_W + 1) // 2) for i in range(d.nseg))
    blocks = ((tiles + 63) // 64) * ((d.N + 63) // 64)
    return tile_hint(64, 64, 1) if blocks >= 384 else tile_hint(32, 64, 1)


def pw_cands(d: ConvDesc) -> list:
    """tile hints (BM = 32, BN = unit width) of the filter-resident streaming kernel (csrc/pw.hip) for a 1x1 / stride-1 convolution
    whose filter — whole, or cut into 2 / 4 / 8 panels of output channels — fits a CU's LDS next to the eight wave buffers; the
    library decides (zsg_conv_igemm_partial_rows returns -1 where the kernel does not apply)."""
    keep = d.tile_hint
    if d.merge_x:       # the network's first convolution: the streaming kernel of csrc/mx.hip behind the same BM = 32 hint
        d.tile_hint = tile_hint(32, 64, 1)
        ok = d.nseg == 1 and lib.zsg_conv_igemm_partial_rows(C.byref(d)) > 0
        d.tile_hint = keep
        return [tile_hint(32, 64, 1)] if ok else []
    if d.nseg != 1 or d.C % 64:
        return []
    out = []
    for uw in (32, 64, 128):
        if d.N % uw == 0:
            d.tile_hint = tile_hint(32, uw, 1)
            if lib.zsg_conv_igemm_partial_rows(C.byref(d)) > 0:
                out.append(d.tile_hint)
    d.tile_hint = keep
    return out


def igemm_partial_rows(d: ConvDesc) -> int:
    """rows of BatchNorm partials the zsg_conv_igemm / zsg_conv_igemm_bnb launch of d (with its tile_hint) writes"""
    n = int(lib.zsg_conv_igemm_partial_rows(C.byref(d)))
    assert n > 0, "no partial-row count for this descriptor / tile hint"
    return n


def autotune_conv(kind: str, fn, d: ConvDesc, args: Sequence, stream: int, ws_bytes: int = 0, split_penalty_ms: float = 0.0,
                  wino_args: Optional[Sequence] = None, wino_fn=None, allow_sk: bool = True) -> int:
    """Pick d.tile_hint for `fn(d, *args, stream)` (kind: 'igemm' | 'wgrad') by timing the candidates on the real
    buffers.  Results are cached per geometry.  ZSG_AUTOTUNE=0 keeps the library heuristic.
    split_penalty_ms: what a split-K choice costs elsewhere (a convolution feeding BatchNorm loses the statistics fused
    in its epilogue: a separate statistics pass + finalize launch), added to the measured time of split candidates.
    wino_args: the same launch through zsg_conv_wino (args with the transformed filter image in place of the weight):
    its tile candidates are timed too; the result carries WINO_FLAG and d.use_wino is set when one of them wins.
    wino_fn: the Winograd entry point that goes with `fn` (default zsg_conv_wino).
    allow_sk: offer the stream-K candidates (sk_cands).  The backward's data gradients pass False: a stream-K launch fills every CU with
    equal shares of the work, which pays where the chain has the GPU to itself (the forward: 4.71 -> 4.68 ms) and costs where the other
    stream's weight gradients would have used the CUs a one-round grid leaves idle — the backward is bound by the CU-time of BOTH
    streams, and the hand-off adds to it (configs[1]: 13.21 -> 13.27 ms with stream-K data gradients, profiles/r06_ab_sk_igemm.txt)."""
    d.use_wino = False
    mode = wino_mode() if wino_args is not None else "0"
    if mode == "0":
        wino_args = None
    no_tune = os.environ.get("ZSG_AUTOTUNE", "1") == "0" or not torch.cuda.is_available()
    if wino_args is not None and (mode == "force" and no_tune):
        d.tile_hint, d.use_wino = (wino_default_hint(d) if kind == "igemm" else 0), True
        return d.tile_hint | WINO_FLAG
    if no_tune:
        return 0
    _check_table_arch()
    add_src, mask = (args[4], args[5]) if kind == "igemm" else (None, None)
    key = _sig(kind, d, (add_src is not None, mask is not None, add_src is not None and add_src is args[2], split_penalty_ms > 0,
                         mode if wino_args is not None else "", deterministic(), "fp32", fn.__name__,
                         os.environ.get("ZSG_PW", "1") != "0" and not (d.merge_x and os.environ.get("ZSG_MX", "1") == "0"),
                         allow_sk and os.environ.get("ZSG_SK", "1") != "0"))
    d._tune_key = key
    if key in _TUNE_CACHE:
        v = _TUNE_CACHE[key]
        d.tile_hint, d.use_wino = v & ~WINO_FLAG, bool(v & WINO_FLAG)
        return v
    rows = sum(d.B * d.seg[i].rows_y * d.seg[i].rows_x for i in range(d.nseg))
    cands = []
    if kind == "igemm":
        tiles = [(64, 64), (128, 64)] + ([(128, 128)] if d.N > 64 else [])
        s0 = d.seg[0]
        dense = (d.nseg == 1 and not d.relu and d.out_ld == d.N and s0.osy == 1 and s0.osx == 1 and s0.out_W == s0.rows_x
                 and s0.out_bstride == s0.rows_y * s0.rows_x * d.N)
        for bm, bn in tiles:
            cands.append(tile_hint(bm, bn, 1))
            if not d.merge_x and (bm == 128 or bn == 64):
                cands.append(tile_hint(bm, bn, 1, 1))          # 8-wave workgroup (64x64: two K groups)
        if not d.merge_x and d.C % 64 == 0 and os.environ.get("ZSG_K64", "1") != "0":
            cands += [tile_hint(bm, bn, 1, w8) | K64_FLAG for bm, bn in tiles for w8 in (0, 1) if not (bm == 128 and bn == 128 and not w8)]
        if fn is lib.zsg_conv_igemm and os.environ.get("ZSG_PW", "1") != "0" and not (d.merge_x and os.environ.get("ZSG_MX", "1") == "0"):
            cands += pw_cands(d)          # (ZSG_MX=0: A/B switch for the streaming first-layer kernel alone)
        if allow_sk and fn is not lib.zsg_conv_igemm_bnpre:
            ensure_stream_scratch(stream)
            cands += sk_cands(d, rows)
        blocks64 = ((rows + 63) // 64) * ((d.N + 63) // 64)
        n_it = s0.ty.n * s0.tx.n * ((d.C + 31) // 32)
        if dense and blocks64 < 1024 and not deterministic():
            for sp in (2, 3, 4, 6, 8, 12, 16, 24, 32):
                if sp <= n_it and blocks64 * sp <= 3072:
                    cands.append(tile_hint(64, 64, sp))
                    if d.C % 64 == 0 and 2 * sp <= n_it and os.environ.get("ZSG_K64", "1") != "0":
                        cands.append(tile_hint(64, 64, sp, 1) | K64_FLAG)
                    if blocks64 * sp < 256:
                        cands.append(tile_hint(128, 64, sp))
    else:
        ncols = d.seg[0].ty.n * d.seg[0].tx.n * d.C
        for bm in ((64, 128) if d.N > 64 else (64,)):
            # (bn 255 = the 256-column tile of the 64-channel layers: every column from one block, dY read once)
            for bn in (((64, 128) if ncols > 64 else (64,)) + ((255,) if (bm == 64 and d.N <= 64 and 128 < ncols <= 256 and d.out_ld % 4 == 0) else ())):
                nmn = ((d.N + bm - 1) // bm) * ((ncols + (256 if bn == 255 else bn) - 1) // (256 if bn == 255 else bn))
                seen = set()
                for target in (256, 384, 512, 768, 1024, 2048):
                    sp = max(1, min(target // nmn, rows // 64, 255))
                    if sp not in seen and sp * d.N * ncols * 4 <= ws_bytes:
                        seen.add(sp)
                        cands.append(tile_hint(bm, bn, sp))
                        if bm == 128 and bn == 128:
                            cands.append(tile_hint(bm, bn, sp, 0, 1))
                            cands.append(tile_hint(bm, bn, sp, 1, 0))          # 8-wave workgroup
                            cands.append(tile_hint(bm, bn, sp, 1, 1))
    trials = [] if mode == "force" else [(fn, marshal(fn, (d,) + tuple(args)), h, 0) for h in cands]
    if wino_args is not None and kind == "igemm":
        wfn = wino_fn or lib.zsg_conv_wino
        wconv = marshal(wfn, (d,) + tuple(wino_args))
        if allow_sk:
            ensure_stream_scratch(stream)
        trials += [(wfn, wconv, h, WINO_FLAG) for h in _wino_cands(d, allow_sk)]
    elif wino_args is not None:           # weight gradient: zsg_conv_wgrad_wino, split-K over 8-tile stages
        wconv = marshal(lib.zsg_conv_wgrad_wino, (d,) + tuple(wino_args))
        tiles = sum(d.B * ((d.seg[i].src_H + 1) // 2) * ((d.seg[i].src_W + 1) // 2) for i in range(d.nseg))
        nmn = ((d.N + 63) // 64) * ((d.C + 63) // 64)
        seen = set()
        for target in (128, 192, 256, 384, 512, 768):
            sp = max(1, min(target // nmn, tiles // 16, 255))
            if sp not in seen and sp * d.N * 9 * d.C * 4 <= ws_bytes:
                seen.add(sp)
                trials.append((lib.zsg_conv_wgrad_wino, wconv, tile_hint(64, 64, sp), WINO_FLAG))
                if sp > 1 and nmn * sp >= 128:        # block order "whole K slices per XCD" (csrc/winowg.hip: xmap)
                    trials.append((lib.zsg_conv_wgrad_wino, wconv, tile_hint(64, 64, sp, 1), WINO_FLAG))
    def set_hint(h):
        d.tile_hint = h

    best = _pick_best(trials, set_hint, stream, lambda h: split_penalty_ms if (kind == "igemm" and ((h >> 16) & 0xff) > 1) else 0.0)
    if os.environ.get("ZSG_TUNE_VERIFY") and fn in (lib.zsg_conv_igemm, lib.zsg_conv_wgrad):
        _verify_candidates(kind, key, trials, set_hint, args[2], args[4] if kind == "igemm" else None, stream)
    d.tile_hint, d.use_wino = best & ~WINO_FLAG, bool(best & WINO_FLAG)
    _TUNE_CACHE[key] = best
    _TUNE_ALTS[key] = list(_LAST_RANKING)
    global _TUNE_DIRTY
    _TUNE_DIRTY = True
    TUNE_INFO["tuned_now"] += 1
    return best


def autotune_wgrad_batch(d: ConvDesc, njobs: int, wino: bool, srcs, dys, scratch_dw: torch.Tensor, ws: torch.Tensor, ws_bytes: int, stream: int) -> int:
    """Split-K choice of a job-batched Winograd weight gradient (zsg_conv_wgrad_wino_batched: njobs convolutions of descriptor d's
    geometry in one launch), tuned like every other launch shape: median of interleaved samples, cached under its own key
    (it goes into ZSG_TUNE_CACHE / the shipped table with the rest).  The trial launches write `scratch_dw` (every job the same scratch
    image — timing only), never a gradient.  Returns the tile hint (also left in d.tile_hint)."""
    global _TUNE_DIRTY
    key = _sig("wgradb", d, (njobs, bool(wino), deterministic()))
    if key in _TUNE_CACHE:
        d.tile_hint = _TUNE_CACHE[key]
        return d.tile_hint
    assert wino, "job batches exist for the Winograd weight gradient only (the direct kernel's measured no gain in the step: profiles/r06_wgrad_batching.txt)"
    fn = lib.zsg_conv_wgrad_wino_batched
    VP = C.c_void_p * njobs
    a_src, a_dy, a_dw = VP(*[t.data_ptr() for t in srcs]), VP(*[t.data_ptr() for t in dys]), VP(*([scratch_dw.data_ptr()] * njobs))
    conv = (C.byref(d), C.c_int32(njobs), C.cast(a_src, C.c_void_p), C.cast(a_dy, C.c_void_p), C.cast(a_dw, C.c_void_p), C.c_int32(0),
            C.c_void_p(ws.data_ptr()), C.c_size_t(ws_bytes))
    trials, seen = [], set()
    if wino:
        tiles = sum(d.B * ((d.seg[i].src_H + 1) // 2) * ((d.seg[i].src_W + 1) // 2) for i in range(d.nseg))
        nmn = ((d.N + 63) // 64) * ((d.C + 63) // 64) * njobs
        for target in (128, 192, 256, 384, 512, 768):
            sp = max(1, min(target // nmn, tiles // 16, 255))
            if sp not in seen and njobs * sp * d.N * 9 * d.C * 4 <= ws_bytes:
                seen.add(sp)
                trials.append((fn, conv, tile_hint(64, 64, sp), 0))

    def set_hint(h):
        d.tile_hint = h
    best = _pick_best(trials, set_hint, stream, lambda h: 0.0)
    if not best:
        raise ZsgError(f"wgrad batch: no candidate ran ({lib.zsg_last_error().decode()})")
    d.tile_hint = best
    _TUNE_CACHE[key] = best
    _TUNE_DIRTY = True
    TUNE_INFO["tuned_now"] += 1
    return best


# ---------------------------------------------------------------------------------------------------------------------
# In-step refinement of the tuner's near-ties (round 6).
# ---------------------------------------------------------------------------------------------------------------------
# The tuner above ranks the candidates of ONE launch by their latency alone on the GPU.  Inside the step a launch runs between its real
# neighbours, beside the other stream's kernels: round 5 found ONE entry — the head's Winograd data gradient, two position groups
# against four — that ties in isolation and is worth 0.11 ms of the 13.2 ms step (the shipped table had to be seeded by hand from the best
# of six fresh tunings); round 6 found the stream-K data gradients, faster alone and slower in the step.  refine_in_step() closes that
# loop without a hand: for every tuned shape whose runner-up candidates lie within REFINE_WINDOW of the winner AND can replace it in the
# lowered plan as it stands (same kernel family, same partial-row / ticket counts — only the tile hint changes, in place), the
# alternative is run in the REAL step (forward + backward of the plan, both streams, a fixed incoming gradient) and kept when the step
# gets faster by more than the measurement noise, twice.
REFINE_WINDOW = float(os.environ.get("ZSG_REFINE_WINDOW", "1.08"))           # runner-ups within this factor of the winner's isolated time ...
REFINE_WINDOW_WAVES = float(os.environ.get("ZSG_REFINE_WINDOW_WAVES", "1.30"))   # ... a wider one for the SAME tile with the other wave count
REFINE_NOISE = float(os.environ.get("ZSG_REFINE_NOISE", "0.003"))        # fraction of the measured step a change has to beat, twice


def hint_compatible(d: ConvDesc, kind: str, cur: int, alt: int) -> bool:
    """May `alt` (hint | WINO_FLAG) replace `cur` in an already lowered launch of descriptor d?  Only what leaves every buffer size and
    every sibling launch of the plan untouched: the same kernel family, no split-K on either side (zero fills, lost epilogue fusions),
    the same number of BatchNorm partial rows and of in-kernel-finalize tickets."""
    if (cur ^ alt) & WINO_FLAG:
        return False
    hc, ha = cur & ~WINO_FLAG, alt & ~WINO_FLAG
    if kind == "wgrad":
        return True                      # (slabs live in the one shared workspace the candidates were sized for)
    if ((hc >> 16) & 0xff) > 1 or ((ha >> 16) & 0xff) > 1:
        return False
    keep = d.tile_hint
    try:
        if cur & WINO_FLAG:
            if (hc & 0xff) != (ha & 0xff):
                return False
            d.tile_hint = hc
            tc = lib.zsg_conv_bn_tail_tickets(C.byref(d), 1)
            d.tile_hint = ha
            return tc == lib.zsg_conv_bn_tail_tickets(C.byref(d), 1)
        d.tile_hint = hc
        rc, tc = lib.zsg_conv_igemm_partial_rows(C.byref(d)), lib.zsg_conv_bn_tail_tickets(C.byref(d), 0)
        d.tile_hint = ha
        return rc == lib.zsg_conv_igemm_partial_rows(C.byref(d)) and tc == lib.zsg_conv_bn_tail_tickets(C.byref(d), 0)
    finally:
        d.tile_hint = keep


def refine_in_step(descs: Sequence[ConvDesc], measure, log=None, max_alts: int = 3) -> dict:
    """descs: the convolution descriptors of a lowered plan (each carries _tune_key from autotune_conv; launches that share a key are
    switched together).  measure() -> milliseconds of the real step (median of a few replays).  An alternative is tried when the tuner
    timed it within REFINE_WINDOW of the current choice alone on the GPU — or within REFINE_WINDOW_WAVES when it is the same tile with
    the other wave count (tile_hint bit 24: 8-wave workgroups / four Winograd position groups): those change how a launch shares a CU
    with the other stream's blocks, which a single-launch timing cannot see (round 5: the head's data gradient, 15 % slower alone with
    four groups, 0.11 ms faster in the step).  It is kept when it beats the incumbent by more than REFINE_NOISE in two measurements
    that bracket a fresh measurement of the incumbent (clock drift cannot fake a win).  The winning choices go into the tuning cache
    (and from there into ZSG_TUNE_CACHE / the shipped table)."""
    global _TUNE_DIRTY
    groups = {}
    for d in descs:
        k = getattr(d, "_tune_key", None)
        if k is not None and k in _TUNE_ALTS:
            groups.setdefault(k, []).append(d)
    base = min(measure(), measure())
    t_start, tried, kept = base, 0, 0

    def put(ds, h):
        for d in ds:
            d.tile_hint = h & ~WINO_FLAG
    for key, ds in groups.items():
        rank, cur, kind = _TUNE_ALTS[key], _TUNE_CACHE[key], key[0]
        t_cur = next((t for t, h in rank if h == cur), rank[0][0])
        alts = []
        for t, h in rank:
            if h == cur:
                continue
            waves_only = ((h ^ cur) & ~(1 << 24)) == 0
            if t <= (REFINE_WINDOW_WAVES if waves_only else REFINE_WINDOW) * t_cur and all(hint_compatible(d, kind, cur, h) for d in ds):
                alts.append(h)
        for alt in alts[:max_alts]:
            put(ds, alt)
            t = measure()
            tried += 1
            verdict = ""
            if t < base * (1 - REFINE_NOISE):
                put(ds, cur)
                b2 = measure()                      # the incumbent again, now: drift cannot fake a win
                put(ds, alt)
                t2 = measure()
                if max(t, t2) < min(base, b2) * (1 - REFINE_NOISE / 2):
                    cur, base, kept, verdict = alt, (t + t2) / 2, kept + 1, " -> kept"
                    _TUNE_CACHE[key] = alt
                    _TUNE_DIRTY = True
                else:
                    base, verdict = (base + b2) / 2, f" (incumbent again {b2:.3f}, alternative again {t2:.3f}: not kept)"
            if log:
                log(f"refine {kind} {str(key[1:5])} {hex(alt)}: {t:.3f} ms vs {base:.3f}{verdict}")
            put(ds, cur)
    TUNE_INFO["refined"] = {"shapes": len(groups), "tried": tried, "kept": kept, "ms_before": round(t_start, 4), "ms_after": round(base, 4)}
    return TUNE_INFO["refined"]
