"""Developer tool (GPU box): upper bound of batching identical-shape DIRECT (1x1) weight gradients — J separate zsg_conv_wgrad calls at
their best tile / split-K against ONE call over J x the input channels (optimistic: dY shared between the emulated jobs)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(1024 << 20, device="cuda")
st = stream_ptr()


def run(B, Ci, Co, hw, hint):
    h, w = hw
    x, dy = torch.randn(B * h * w * Ci, device="cuda"), torch.randn(B * h * w * Co, device="cuda")
    dw = torch.zeros(Co, Ci, device="cuda")
    d = ops.fwd_desc(ops.TView(x, B, Ci, Ci, [ops.Level(0, h, w, h * w * Ci)]), ops.TView(dy, B, Co, Co, [ops.Level(0, h, w, h * w * Co)]),
                     Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
    t = timeit(lambda: lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, WS.data_ptr(), WS.numel() * 4, st), n=20)
    return t * 1e3


def best(B, Ci, Co, hw, sps):
    res = {}
    for bm, bn in ((64, 64), (128, 64), (128, 128)):
        for sp in sps:
            t = run(B, Ci, Co, hw, ops.tile_hint(bm, bn, sp))
            if t == t and t < 1e8:
                res[(bm, bn, sp)] = t
    k = min(res, key=res.get)
    return k, res[k]


for name, J, Ci, Co, hw in (("l3_conv1", 5, 1024, 256, (19, 19)), ("l3_conv3", 5, 256, 1024, (19, 19)), ("l2_conv1", 3, 512, 128, (38, 38)), ("l2_conv3", 3, 128, 512, (38, 38)),
                            ("l4_conv1", 2, 2048, 512, (10, 10)), ("l4_conv3", 2, 512, 2048, (10, 10)), ("l1_conv1", 2, 256, 64, (75, 75)), ("l1_conv3", 2, 64, 256, (75, 75))):
    k1, t1 = best(16, Ci, Co, hw, (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64))
    kj, tj = best(16, Ci * J, Co, hw, (1, 2, 3, 4, 6, 8, 12, 16, 24, 32))
    print(f"{name} x{J}: separate {J} x {t1:.1f} = {J * t1:.1f} us at {k1} | batched-emulated {tj:.1f} us at {kj} | ratio {tj / (J * t1):.2f}", flush=True)
