"""Developer micro-benchmark: stream-K candidates of the implicit GEMM next to the plain tiles, single launches back to back.
usage: python tools/bench_sk.py [shape names]      (default: the under-filled grids of ResNet-50 + FPN at B = 16)
Prints per shape the best plain (unsplit, deterministic) tile, the best atomic split-K, and every stream-K candidate (us per launch,
TFLOP/s), forward form with the fused BatchNorm partial rows where the plain launch has them."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

# name, B, Cin, Cout, H, W, k, s, p
SHAPES = [
    ("l3_conv1", 16, 1024, 256, 19, 19, 1, 1, 0),      # = layer3 conv3's data gradient
    ("l3_conv3", 16, 256, 1024, 19, 19, 1, 1, 0),      # = layer3 conv1's data gradient
    ("l3_0_conv1", 16, 512, 256, 38, 38, 1, 1, 0),
    ("l4_conv1", 16, 2048, 512, 10, 10, 1, 1, 0),
    ("l4_conv3", 16, 512, 2048, 10, 10, 1, 1, 0),
    ("l4_0_conv1", 16, 1024, 512, 19, 19, 1, 1, 0),
    ("l3_ds", 16, 512, 1024, 38, 38, 1, 2, 0),
    ("l4_ds", 16, 1024, 2048, 19, 19, 1, 2, 0),
    ("l3_c2s2", 16, 256, 256, 38, 38, 3, 2, 1),
    ("l4_c2s2", 16, 512, 512, 19, 19, 3, 2, 1),
    ("P5_1", 16, 2048, 256, 10, 10, 1, 1, 0),
    ("P4_1", 16, 1024, 256, 19, 19, 1, 1, 0),
    ("P6", 16, 2048, 256, 10, 10, 3, 2, 1),
    ("l2_conv1", 16, 512, 128, 38, 38, 1, 1, 0),
]


def timeit(fn, n=30):
    for _ in range(3):
        if fn():
            return float("inf")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def name_of(h):
    s = f"{h & 0xff}x{(h >> 8) & 0xff}"
    if (h >> 24) & 1:
        s += "w"
    if (h >> 27) & 1:
        s += "k"
    if (h >> 16) & 0xff > 1:
        s += f"/s{(h >> 16) & 0xff}"
    if (h >> 28) & 3:
        s += f"+sk{(h >> 28) & 3}"
    return s


WINO_SHAPES = [
    ("l3_conv2", 16, 256, 256, 19, 19),
    ("l4_conv2", 16, 512, 512, 10, 10),
    ("l2_conv2", 16, 128, 128, 38, 38),
    ("P5_2", 16, 256, 256, 10, 10),
]


def wino_main(only, st):
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
        gf = 2.0 * B * H * W * Co * Ci * 9 / 1e9
        d0 = ops.fwd_desc(xv, yv, Ci, Co, 3, 1, 1, 1, wC=Ci)
        res = []
        for h in ops._wino_cands(d0):
            if ((h >> 16) & 0xff) > 1:
                continue
            d = ops.fwd_desc(xv, yv, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=h)
            res.append((timeit(lambda: lib.zsg_conv_wino(C.byref(d), x.data_ptr(), U.data_ptr(), y.data_ptr(), None, None, None, part.data_ptr(), st)), h))
        res.sort()
        print(f"{name:11s} tiles={tiles:5d} N={Co:4d} C={Ci:4d} {gf:6.2f} GF (direct) ideal {gf * 4 / 9 / 157.3 * 1e3:5.1f} us executed")
        print("   winograd: " + "  ".join(f"{name_of(h)} {t:5.1f}" for t, h in res), flush=True)


def main():
    only = sys.argv[1:]
    st = stream_ptr()
    ops.ensure_stream_scratch(st)
    wino_main(only, st)
    for name, B, Ci, Co, H, W, k, s, p in SHAPES:
        if only and name not in only:
            continue
        Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
        rows = B * Ho * Wo
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
        y = torch.empty(B, Ho, Wo, Co, device="cuda")
        part = torch.empty(rows // 64 + 2, 2, Co, device="cuda")
        xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        yv = ops.TView(y.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
        gf = 2.0 * rows * Co * Ci * k * k / 1e9
        plain = []
        for bm, bn in ((64, 64), (128, 64), (128, 128)):
            for w8 in (0, 1):
                for k64 in (0, 1):
                    if (bm, bn, w8) == (128, 128, 0) and k64:
                        continue
                    plain.append(ops.tile_hint(bm, bn, 1, w8) | (k64 << 27))
        split = [ops.tile_hint(64, 64, sp) for sp in (2, 3, 4, 6, 8)] + [ops.tile_hint(64, 64, sp, 1) | ops.K64_FLAG for sp in (2, 3, 4)]
        d0 = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci)
        sk = ops.sk_cands(d0, rows)

        def run(h, with_part):
            d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=h)
            return timeit(lambda: lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None,
                                                     part.data_ptr() if with_part else None, st))

        res_p = sorted((run(h, True), h) for h in plain)
        res_s = sorted((run(h, False), h) for h in split)
        res_k = sorted((run(h, True), h) for h in sk)
        print(f"{name:11s} M={rows:6d} N={Co:4d} K={Ci * k * k:5d} {gf:6.2f} GF ideal {gf / 157.3 * 1e3:5.1f} us")
        print("   plain   : " + "  ".join(f"{name_of(h)} {t:5.1f}" for t, h in res_p[:5]))
        print("   split-K : " + "  ".join(f"{name_of(h)} {t:5.1f}" for t, h in res_s[:3]) + "   (atomics, no fused statistics)")
        print("   stream-K: " + "  ".join(f"{name_of(h)} {t:5.1f}" for t, h in res_k[:8]))
        if res_k and res_p:
            print(f"   best stream-K / best plain: {res_k[0][0]:.1f} / {res_p[0][0]:.1f} us = {res_k[0][0] / res_p[0][0]:.3f}   "
                  f"({gf / res_k[0][0] * 1e3 / 157.3:.3f} vs {gf / res_p[0][0] * 1e3 / 157.3:.3f} of the fp32-MFMA peak)", flush=True)


if __name__ == "__main__":
    main()
