"""Developer tool: launch ONE Winograd convolution configuration a few times (for rocprofv3 --pmc runs).
usage: python tools/one_wino.py <shape of tools/bench_wino.py> <TB> <BN> [splits] [ps4] [wgrad]      (wgrad = 1: zsg_conv_wgrad_wino, splits = its K slices)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import SHAPES
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

name, TB, BN = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sp = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ps4 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
wg = int(sys.argv[6]) if len(sys.argv) > 6 else 0
_, B, Ci, Co, sizes = [x for x in SHAPES if x[0] == name][0]
lv_in, lv_out, oi, oo = [], [], 0, 0
for (h, w) in sizes:
    lv_in.append(ops.Level(oi, h, w, h * w * Ci))
    lv_out.append(ops.Level(oo, h, w, h * w * Co))
    oi += B * h * w * Ci
    oo += B * h * w * Co
x = torch.randn(oi, device="cuda")
y = torch.empty(oo, device="cuda")
w = torch.randn(Co, 3, 3, Ci, device="cuda") * 0.05
src, out = ops.TView(x, B, Ci, Ci, lv_in), ops.TView(y, B, Co, Co, lv_out)
U = torch.empty(int(lib.zsg_wino_u_elems(Ci, Co)), device="cuda")
jobs = ops.WinoJobs()
jobs.add(w.data_ptr(), U.data_ptr(), Co, Ci, 9 * Ci, Ci, False)
jobs.finish("cuda")
st = stream_ptr()
jobs.launch(st)
d = ops.fwd_desc(src, out, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=TB | (BN << 8) | (sp << 16) | (ps4 << 24))
if wg:
    ws = torch.empty(64 << 20, device="cuda")
    dy = torch.randn(oo, device="cuda")
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    dyv = ops.TView(dy, B, Co, Co, lv_out)
    d = ops.fwd_desc(src, dyv, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))
for _ in range(8):
    if wg:
        check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, st), "wino wgrad")
    else:
        check(lib.zsg_conv_wino(C.byref(d), x.data_ptr(), U.data_ptr(), y.data_ptr(), None, None, None, None, st), "wino")
torch.cuda.synchronize()
