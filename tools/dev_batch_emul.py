"""Developer tool (GPU box): upper bound of BATCHING identical-shape Winograd weight gradients into one launch — J separate calls at their
best split-K against ONE call over J x the input channels (J x the (n, c) blocks, the same K range, fewer splits): what a job-batched
launch would cost if its blocks behaved like additional channel blocks (the emulation shares dY between the jobs: optimistic)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(512 << 20, device="cuda")
st = stream_ptr()


def run(B, Ci, Co, hw, sp):
    h, w = hw
    x, dy = torch.randn(B * h * w * Ci, device="cuda"), torch.randn(B * h * w * Co, device="cuda")
    dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
    d = ops.fwd_desc(ops.TView(x, B, Ci, Ci, [ops.Level(0, h, w, h * w * Ci)]), ops.TView(dy, B, Co, Co, [ops.Level(0, h, w, h * w * Co)]),
                     Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))
    return timeit(lambda: check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, WS.data_ptr(), WS.numel() * 4, st), "wgw"), n=30) * 1e3


for name, J, Ci, Co, hw in (("l3_conv2", 5, 256, 256, (19, 19)), ("l2_conv2", 3, 128, 128, (38, 38)), ("l4_conv2", 2, 512, 512, (10, 10)),
                            ("l1_conv2", 2, 64, 64, (75, 75))):
    one = {sp: run(16, Ci, Co, hw, sp) for sp in (1, 2, 4, 8, 16, 32)}
    bat = {sp: run(16, Ci * J, Co, hw, sp) for sp in (1, 2, 3, 4, 6, 8, 16)}
    b1, bj = min(one.values()), min(bat.values())
    print(f"{name} x{J}: separate {J} x {b1:.1f} = {J * b1:.1f} us (splits: " + " ".join(f"{k}:{v:.1f}" for k, v in one.items()) + f") | batched-emulated {bj:.1f} us ("
          + " ".join(f"{k}:{v:.1f}" for k, v in bat.items()) + f") | ratio {bj / (J * b1):.2f}", flush=True)
