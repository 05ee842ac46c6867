"""Developer tool (GPU box): write the Winograd weight gradient of two bench shapes (fixed seed) to a file, so that two builds of libzsg
(ZSG_LIB_PATH) can be compared bit for bit.   python tools/dev_ww_bits.py <out.pt>"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import SHAPES
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(128 << 20, device="cuda")
res = {}
for name, sp in (("head", 16), ("l3_conv2", 16), ("l4_conv2", 4), ("l3_conv2", 1)):
    _, B, Ci, Co, sizes = [x for x in SHAPES if x[0] == name][0]
    lv_in, lv_out, oi, oo = [], [], 0, 0
    for (h, w) in sizes:
        lv_in.append(ops.Level(oi, h, w, h * w * Ci))
        lv_out.append(ops.Level(oo, h, w, h * w * Co))
        oi += B * h * w * Ci
        oo += B * h * w * Co
    g = torch.Generator(device="cuda").manual_seed(5)
    x, dy = torch.randn(oi, device="cuda", generator=g), torch.randn(oo, device="cuda", generator=g)
    dw = torch.ones(Co, 3, 3, Ci, device="cuda")
    d = ops.fwd_desc(ops.TView(x, B, Ci, Ci, lv_in), ops.TView(dy, B, Co, Co, lv_out), Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))
    check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, WS.data_ptr(), WS.numel() * 4, stream_ptr()), "wgw")
    torch.cuda.synchronize()
    res[f"{name}/{sp}"] = dw.cpu()
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    for k in res:
        print(k, "bit-identical" if torch.equal(res[k], other[k]) else f"DIFFERENT max {float((res[k] - other[k]).abs().max()):.3e}")
torch.save(res, sys.argv[1])
