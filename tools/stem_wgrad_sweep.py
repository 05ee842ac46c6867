"""Developer tool: the stem's weight gradient (7x7x4 -> 64, 360 000 output pixels at B=16) over tile / split choices."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

B, H, W = 16, 300, 300
x = torch.randn(B, H, W, 4, device="cuda")
dy = torch.randn(B, 150, 150, 64, device="cuda")
dw = torch.zeros(64, 7, 7, 4, device="cuda")
ws = torch.empty(64 << 20, device="cuda")
xv = ops.TView(x.view(-1), B, 4, 4, [ops.Level(0, H, W, H * W * 4)])
dv = ops.TView(dy.view(-1), B, 64, 64, [ops.Level(0, 150, 150, 150 * 150 * 64)])
gf = 2.0 * B * 150 * 150 * 64 * 196 / 1e9
for bn in (64, 128, 255):
    line = f"64x{256 if bn == 255 else bn}:"
    for sp in (32, 64, 96, 128, 192, 255):
        d = ops.fwd_desc(xv, dv, 4, 64, 7, 2, 3, 1, wC=4, tile_hint=ops.tile_hint(64, bn, sp))
        ms = timeit(lambda: check(lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, stream_ptr())))
        line += f"  s{sp}: {ms * 1e3:6.1f}us"
    print(line, flush=True)
print(f"({gf:.2f} GF: {gf / 157.3 * 1e3:.1f} us at the fp32-MFMA peak)")
