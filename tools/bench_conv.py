"""Developer micro-benchmark: time conv fwd / dgrad / wgrad launches for given shapes and tile hints.
usage: python tools/bench_conv.py            (built-in ResNet-50/FPN/head shape list at B=16)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(64 << 20, device="cuda")

# name, B, Cin, Cout, H, W, k, s, p
SHAPES = [
    ("head3x3_38", 16, 256, 256, 38, 38, 3, 1, 1),
    ("l1_conv2", 16, 64, 64, 75, 75, 3, 1, 1),
    ("l1_conv1", 16, 256, 64, 75, 75, 1, 1, 0),
    ("l1_conv3", 16, 64, 256, 75, 75, 1, 1, 0),
    ("l2_conv2", 16, 128, 128, 38, 38, 3, 1, 1),
    ("l2_conv3", 16, 128, 512, 38, 38, 1, 1, 0),
    ("l3_conv2", 16, 256, 256, 19, 19, 3, 1, 1),
    ("l3_conv1", 16, 1024, 256, 19, 19, 1, 1, 0),
    ("l3_conv3", 16, 256, 1024, 19, 19, 1, 1, 0),
    ("l4_conv2", 16, 512, 512, 10, 10, 3, 1, 1),
    ("l4_conv3", 16, 512, 2048, 10, 10, 1, 1, 0),
    ("l3_ds", 16, 512, 1024, 38, 38, 1, 2, 0),
    ("l3_c2s2", 16, 256, 256, 38, 38, 3, 2, 1),
    ("P6", 16, 2048, 256, 10, 10, 3, 2, 1),
]
TILES = [0, 128 | (128 << 8), 128 | (64 << 8), 64 | (64 << 8)]
TILES_IG = TILES + [128 | (128 << 8) | (1 << 24), 128 | (64 << 8) | (1 << 24), 64 | (64 << 8) | (1 << 24)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    only = sys.argv[1:] if len(sys.argv) > 1 else None
    st = stream_ptr()
    for name, B, Ci, Co, H, W, k, s, p in SHAPES:
        if only and name not in only:
            continue
        Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
        wt = torch.randn(Ci, k, k, Co, device="cuda") * 0.05
        y = torch.empty(B, Ho, Wo, Co, device="cuda")
        dy = torch.randn(B, Ho, Wo, Co, device="cuda")
        dx = torch.empty(B, H, W, Ci, device="cuda")
        dw = torch.zeros(Co, k, k, Ci, device="cuda")
        xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        yv = ops.TView(y.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
        dyv = ops.TView(dy.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
        dxv = ops.TView(dx.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        gf = 2.0 * B * Ho * Wo * Co * Ci * k * k / 1e9
        line = f"{name:12s} M={B * Ho * Wo:6d} N={Co:4d} K={Ci * k * k:5d} {gf:7.2f} GF |"
        for t in TILES_IG:
            d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=t)
            ms = timeit(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st)))
            line += f" f[{t & 0xff}x{(t >> 8) & 0xff}{chr(119) if (t >> 24) & 1 else chr(32)}] {gf / ms:5.1f}"
        line += " |"
        for t in TILES_IG:
            d = ops.dgrad_desc(dyv, dxv, Co, Ci, k, s, p, 1, tile_hint=t)
            ms = timeit(lambda: check(lib.zsg_conv_igemm(C.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st)))
            line += f" d[{t & 0xff}x{(t >> 8) & 0xff}{chr(119) if (t >> 24) & 1 else chr(32)}] {gf / ms:5.1f}"
        line += " | wgrad"
        for t in [] if os.environ.get("NO_WGRAD") else TILES + [128 | (128 << 8) | (1 << 25), 128 | (128 << 8) | (1 << 24), 128 | (128 << 8) | (1 << 24) | (1 << 25)]:
            d = ops.fwd_desc(xv, dyv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=t)
            ms = timeit(lambda: check(lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st)))
            line += f" [{t & 0xff}x{(t >> 8) & 0xff}{'w8' if (t >> 24) & 1 else ''}{'k32' if (t >> 25) & 1 else ''}] {gf / ms:6.1f}"
        if B * Ho * Wo <= 2048:
            line += " | fwd split-K"
            for sp in (2, 4, 8, 16):
                d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))
                ms = timeit(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st)))
                line += f" [{sp}] {gf / ms:6.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
