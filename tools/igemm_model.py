"""Developer tool: fixed overhead vs per-K-step time of the implicit-GEMM tile variants (1x1 convolutions, single launches
separated by a synchronise: no overlap between consecutive launches)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr


def t_single(fn, n=15):
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


def main():
    st = stream_ptr()
    variants = [(64, 64, 0), (64, 64, 1), (128, 64, 0), (128, 64, 1), (128, 128, 0), (128, 128, 1)]
    bx = int(os.environ.get("K64", "0")) << 27          # variant bit: 64-deep K tiles
    print("time of ONE launch in us (median of 15, device idle before each)")
    shapes = ((5776, 256), (1600, 512), (23104, 128), (5776, 1024), (8192, 256), (16384, 256))
    ks = (64, 256, 1024, 2048)
    if os.environ.get("SHAPES"):                     # e.g. SHAPES="1600x512,5776x256" KS="2048"
        shapes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(","))
    if os.environ.get("KS"):
        ks = tuple(int(v) for v in os.environ["KS"].split(","))
    for (M, N) in shapes:
        for K in ks:
            pad = int(os.environ.get("PAD", "0"))          # row pitch = K + PAD floats for both operands (L2 channel camping experiment)
            Kp = K + pad
            x = torch.randn(M, Kp, device="cuda")
            w = torch.randn(N, Kp, device="cuda") * 0.05
            y = torch.empty(M, N, device="cuda")
            xv = ops.TView(x.view(-1), 1, K, Kp, [ops.Level(0, 1, M, M * Kp)])
            yv = ops.TView(y.view(-1), 1, N, N, [ops.Level(0, 1, M, M * N)])
            gf = 2.0 * M * N * K / 1e9
            line = f"M={M:6d} N={N:5d} K={K:5d} ideal {gf / 157.3 * 1e3:6.1f}us |"
            for bm, bn, w8 in variants:
                if bn == 128 and N < 128:
                    line += "             "
                    continue
                d = ops.fwd_desc(xv, yv, K, N, 1, 1, 0, 1, wC=Kp, tile_hint=ops.tile_hint(bm, bn, 1, w8) | bx)
                us = t_single(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st)))
                blk = -(-M // bm) * -(-N // bn)
                line += f" {bm}x{bn}{'w' if w8 else ' '}[{blk:4d}]{us:6.1f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
