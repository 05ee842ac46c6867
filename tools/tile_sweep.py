"""Developer tool: forward-conv TFLOP/s over (tile, waves, split-K) for the mid-size encoder shapes (B=16)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import SHAPES, timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr


def main():
    names = sys.argv[1:] or ["l3_conv2", "l2_conv2", "l4_conv2", "l3_conv1", "l3_conv3", "l1_conv2"]
    st = stream_ptr()
    for name, B, Ci, Co, H, W, k, s, p in SHAPES:
        if name not in names:
            continue
        Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
        x = torch.randn(B, H, W, Ci, device="cuda")
        w = torch.randn(Co, k, k, Ci, device="cuda") * 0.05
        y = torch.empty(B, Ho, Wo, Co, device="cuda")
        xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
        yv = ops.TView(y.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
        M = B * Ho * Wo
        gf = 2.0 * M * Co * Ci * k * k / 1e9
        print(f"{name}: M={M} N={Co} K={Ci * k * k} {gf:.2f} GF  (ideal {gf / 157.3 * 1e3:.1f} us)")
        for bm, bn, w8 in ((64, 64, 0), (128, 64, 0), (128, 64, 1), (128, 128, 0), (128, 128, 1)):
            blk = -(-M // bm) * -(-Co // bn)
            line = f"   {bm}x{bn}{'w8' if w8 else '  '} blocks={blk:4d}:"
            for sp in (1, 2, 3, 4, 5, 6, 8, 12):
                d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=ops.tile_hint(bm, bn, sp, w8))
                ms = timeit(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st)))
                line += f"  s{sp}:{gf / ms:6.1f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
