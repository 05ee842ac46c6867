"""Developer tool (GPU box): zsg_conv_wgrad_wino_batched against J separate zsg_conv_wgrad_wino calls on J DIFFERENT operand sets (the
real thing, unlike tools/dev_batch_emul.py): time (kernel + slab reductions) and agreement of every job's gradient."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_wino import timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

WS = torch.empty(2048 << 20, device="cuda")
st = stream_ptr()
VP = C.c_void_p * 8

for name, J, Ci, Co, hw in (("l3_conv2", 5, 256, 256, (19, 19)), ("l2_conv2", 3, 128, 128, (38, 38)), ("l4_conv2", 2, 512, 512, (10, 10)),
                            ("l1_conv2", 2, 64, 64, (75, 75))):
    h, w = hw
    B = 16
    xs = [torch.randn(B * h * w * Ci, device="cuda") for _ in range(J)]
    dys = [torch.randn(B * h * w * Co, device="cuda") for _ in range(J)]
    dws = [torch.zeros(Co, 3, 3, Ci, device="cuda") for _ in range(J)]
    dwb = [torch.zeros(Co, 3, 3, Ci, device="cuda") for _ in range(J)]

    def desc(sp):
        return ops.fwd_desc(ops.TView(xs[0], B, Ci, Ci, [ops.Level(0, h, w, h * w * Ci)]), ops.TView(dys[0], B, Co, Co, [ops.Level(0, h, w, h * w * Co)]),
                            Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))

    def separate(sp):
        d = desc(sp)
        for j in range(J):
            check(lib.zsg_conv_wgrad_wino(C.byref(d), xs[j].data_ptr(), dys[j].data_ptr(), dws[j].data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgw")

    def batched(sp):
        d = desc(sp)
        a, b, c = VP(*[x.data_ptr() for x in xs]), VP(*[x.data_ptr() for x in dys]), VP(*[x.data_ptr() for x in dwb])
        check(lib.zsg_conv_wgrad_wino_batched(C.byref(d), J, a, b, c, 0, WS.data_ptr(), WS.numel() * 4, st), "wgwb")

    one = {sp: timeit(lambda: separate(sp), n=20) * 1e3 for sp in (4, 8, 16, 32, 48, 64, 96, 128, 192)}
    bat = {sp: timeit(lambda: batched(sp), n=20) * 1e3 for sp in (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96)}
    s1, sj = min(one, key=one.get), min(bat, key=bat.get)
    separate(s1)
    batched(sj)
    torch.cuda.synchronize()
    err = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(dwb, dws))
    print(f"{name} x{J}: separate {one[s1]:.1f} us at /{s1} (" + " ".join(f"{k}:{v:.1f}" for k, v in one.items()) + f") | batched {bat[sj]:.1f} us at /{sj} ("
          + " ".join(f"{k}:{v:.1f}" for k, v in bat.items()) + f") | ratio {bat[sj] / one[s1]:.2f} | max rel diff {err:.1e}", flush=True)
