"""Developer tool (GPU box): time zsg_conv_wgrad_wino (kernel + slab reduction) on the bench shapes of tools/bench_wino.py.
usage: python tools/bench_winowg.py [shape:splits ...]     (default: head:16 P3_2:16 l3_conv2:16 l2_conv2:16 l4_conv2:4)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import SHAPES, timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(128 << 20, device="cuda")
st = stream_ptr()
out = []
for spec in (sys.argv[1:] or ["head:16", "P3_2:16", "l3_conv2:16", "l2_conv2:16", "l4_conv2:4"]):
    name, sp = spec.split(":")
    _, B, Ci, Co, sizes = [x for x in SHAPES if x[0] == name][0]
    lv_in, lv_out, oi, oo = [], [], 0, 0
    for (h, w) in sizes:
        lv_in.append(ops.Level(oi, h, w, h * w * Ci))
        lv_out.append(ops.Level(oo, h, w, h * w * Co))
        oi += B * h * w * Ci
        oo += B * h * w * Co
    x, dy = torch.randn(oi, device="cuda"), torch.randn(oo, device="cuda")
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    d = ops.fwd_desc(ops.TView(x, B, Ci, Ci, lv_in), ops.TView(dy, B, Co, Co, lv_out), Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, int(sp)))
    t = timeit(lambda: check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgw"), n=30)
    gf = 2.0 * sum(B * h * ww for h, ww in sizes) * Co * 9 * Ci / 1e9
    out.append(f"{name}/{sp} {t * 1e3:6.1f}us {gf / t * 4 / 9:5.1f}TF")
print(" | ".join(out))
