"""Developer tool: the filter-resident streaming kernel (csrc/pw.hip, tile_hint BM = 32) against the implicit-GEMM tiles on the
1x1 shapes of the trunk's first stages — single launches separated by a synchronise, plain and with the fused BatchNorm
statistics, plus a bit-level comparison of the results with the 64x64 tile's (same products, different summation order: the
difference is reported, not asserted)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr
from tools.igemm_model import t_single


def main():
    st = stream_ptr()
    shapes = ((90000, 64, 256), (90000, 256, 64), (90000, 64, 64), (23104, 128, 512), (23104, 512, 128), (90000, 256, 128), (5776, 256, 1024))
    if os.environ.get("SHAPES"):                     # e.g. SHAPES="90000x64x256"
        shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(","))
    print("time of ONE launch in us (median of 15, device idle before each); '+st' = with fused BatchNorm statistics")
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * 0.05
        y = torch.empty(M, N, device="cuda")
        part = torch.empty(4 << 20, device="cuda")
        xv = ops.TView(x.view(-1), 1, K, K, [ops.Level(0, 1, M, M * K)])
        yv = ops.TView(y.view(-1), 1, N, N, [ops.Level(0, 1, M, M * N)])
        gf = 2.0 * M * N * K / 1e9
        hbm = (M * K + M * N + N * K) * 4
        line = f"M={M:6d} K={K:4d} N={N:4d} mfma {gf / 157.3 * 1e3:5.1f}us hbm@5.5TB/s {hbm / 5.5e6:5.1f}us |"
        hints = [("64x64", ops.tile_hint(64, 64, 1)), ("128x64w", ops.tile_hint(128, 64, 1, 1))]
        if N >= 128:
            hints.append(("128x128w", ops.tile_hint(128, 128, 1, 1)))
        probe = ops.fwd_desc(xv, yv, K, N, 1, 1, 0, 1, wC=K)
        hints += [(f"pw{(h >> 8) & 0xff}", h) for h in ops.pw_cands(probe)]
        ref = None
        for name, h in hints:
            d = ops.fwd_desc(xv, yv, K, N, 1, 1, 0, 1, wC=K, tile_hint=h)
            y.fill_(float("nan"))
            us = t_single(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st)))
            us2 = t_single(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), st)))
            if ref is None:
                ref = y.clone()
                err = 0.0
            else:
                err = float((y - ref).abs().max())
            line += f" {name} {us:5.1f} +st {us2:5.1f} (d {err:.1e})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
