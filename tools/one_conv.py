"""Developer tool: launch ONE conv configuration a few times (for rocprofv3 --pmc runs).
usage: python tools/one_conv.py <shape> <fwd|dgrad|wgrad> <BM> <BN> [w8] [splits]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_conv import SHAPES
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(64 << 20, device="cuda")

name, mode, bm, bn = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
w8 = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sp = int(sys.argv[6]) if len(sys.argv) > 6 else 0
_, B, Ci, Co, H, W, k, s, p = [x for x in SHAPES if x[0] == name][0]
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
hint = ops.tile_hint(bm, bn, sp, w8)
st = stream_ptr()
for _ in range(8):
    if mode == "fwd":
        d = ops.fwd_desc(xv, yv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=hint)
        check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st))
    elif mode == "dgrad":
        d = ops.dgrad_desc(dyv, dxv, Co, Ci, k, s, p, 1, tile_hint=hint)
        check(lib.zsg_conv_igemm(C.byref(d), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st))
    else:
        d = ops.fwd_desc(xv, dyv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=hint)
        check(lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st))
torch.cuda.synchronize()
