"""Developer tool (GPU box): stress the stream-K hand-off (csrc/igemm.hip, csrc/wino.hip: partial tile by write-through stores -> flag ->
finisher) for a RARE visibility race.  Per shape / candidate: one reference launch, then N launches whose output buffer and fused
BatchNorm partial rows are poisoned (NaN) first and compared with the reference ON THE DEVICE (no host sync per launch), while a second
stream keeps the memory system busy with an unrelated kernel mix.  A stale or missing partial tile shows as a mismatching element.
usage: python tools/sk_stress.py [launches per candidate, default 20000] [shape names]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_sk import SHAPES, WINO_SHAPES, name_of
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr


def stress(tag, launch, out, part, n, side, noise):
    """launch() enqueues one launch writing `out` (+ `part`); returns the number of launches (of n) with any differing element"""
    out.fill_(float("nan"))
    part.fill_(float("nan"))
    check(launch(), tag)
    torch.cuda.synchronize()
    ref_o, ref_p = out.clone(), part.clone()
    assert not torch.isnan(ref_o).any(), tag + ": reference launch left NaNs"
    bad = torch.zeros(1, device="cuda", dtype=torch.int64)
    main = torch.cuda.current_stream()
    for i in range(n):
        if i % 8 == 0:          # the neighbour: a fill and a copy of 64 MB each, re-issued while the main stream works
            with torch.cuda.stream(side):
                noise[0].fill_(float(i))
                noise[1].copy_(noise[0])
        out.fill_(float("nan"))
        part.fill_(float("nan"))
        rc = launch()
        if rc:
            raise RuntimeError(f"{tag}: launch {i} failed ({rc}): {lib.zsg_last_error().decode()}")
        bad += ((out != ref_o).any() | (part != ref_p).any()).to(torch.int64)
        if i % 2048 == 2047:
            main.synchronize()
    torch.cuda.synchronize()
    return int(bad.item())


def main():
    args = [a for a in sys.argv[1:]]
    n = int(args[0]) if args and args[0].isdigit() else 20000
    only = [a for a in args if not a.isdigit()]
    st = stream_ptr()
    ops.ensure_stream_scratch(st)
    side = torch.cuda.Stream()
    noise = [torch.empty(16 << 20, device="cuda"), torch.empty(16 << 20, device="cuda")]
    total_bad = 0
    for name, B, Ci, Co, H, W in WINO_SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, 3, 3, Ci, device="cuda") * 0.05
        y = torch.empty(B, H, W, Co, device="cuda")
        U = torch.empty(int(lib.zsg_wino_u_elems(Ci, Co)), device="cuda")
        jobs = ops.WinoJobs()
        jobs.add(w.data_ptr(), U.data_ptr(), Co, Ci, 9 * Ci, Ci, 0)
        jobs.finish("cuda")
        jobs.launch(st)
        tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
        part = torch.empty(tiles // 32 + 2, 2, Co, device="cuda")
        xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        yv = ops.TView(y.view(-1), B, Co, Co, [ops.Level(0, H, W, H * W * Co)])
        d0 = ops.fwd_desc(xv, yv, Ci, Co, 3, 1, 1, 1, wC=Ci)
        for h in ops._wino_cands(d0):
            if not (h >> ops.SK_SHIFT) & 3:
                continue
            d = ops.fwd_desc(xv, yv, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=h)
            rows = lib.zsg_conv_igemm_partial_rows  # noqa: F841 (wino rows: tiles / 32)
            pr = part[:(tiles + 31) // 32]
            b = stress(f"wino {name} {name_of(h)}", lambda: lib.zsg_conv_wino(C.byref(d), x.data_ptr(), U.data_ptr(), y.data_ptr(), None, None, None, pr.data_ptr(), st),
                       y, pr, n, side, noise)
            total_bad += b
            print(f"wino  {name:10s} {name_of(h):14s}: {b} of {n} launches differ", flush=True)
    for name, B, Ci, Co, H, W, k, s, p in SHAPES:
        if only and name not in only:
            continue
        Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
        rows = B * Ho * Wo
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
        y = torch.empty(B, Ho, Wo, Co, device="cuda")
        xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        yv = ops.TView(y.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
        d0 = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci)
        for h in ops.sk_cands(d0, rows)[:4]:
            d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=h)
            part = torch.empty(int(ops.igemm_partial_rows(d)), 2, Co, device="cuda")
            try:
                b = stress(f"igemm {name} {name_of(h)}", lambda: lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), st),
                           y, part, n, side, noise)
            except RuntimeError as e:
                if "launch 0" in str(e) or "refus" in str(e):
                    continue
                raise
            except Exception as e:          # a refused candidate (more tiles than workgroups): the reference launch fails
                print(f"igemm {name:10s} {name_of(h):14s}: skipped ({type(e).__name__})")
                continue
            total_bad += b
            print(f"igemm {name:10s} {name_of(h):14s}: {b} of {n} launches differ", flush=True)
    print("TOTAL differing launches:", total_bad)


if __name__ == "__main__":
    main()
