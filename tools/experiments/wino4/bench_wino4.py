"""Developer tool (GPU box): single-launch times of zsg_conv_wino4 next to zsg_conv_wino's best tile on the shapes F(4x4,3x3) is offered
for (pyramid output P3_2, the head's levels): us per launch (mean of 20 back-to-back launches) and executed / algorithmic TFLOP/s.
usage: ZSG_LIB_PATH=<EXPERIMENTS=1 build> python tools/experiments/wino4/bench_wino4.py"""
import ctypes as C
import os
import sys
import torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), os.path.dirname(os.path.abspath(__file__))]
from zsgnet_pytorch_amd import ops                                   # noqa: E402
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr
from bind import bind
bind()           # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def levels(B, C_, sizes):
    lv, off = [], 0
    for (h, w) in sizes:
        lv.append(ops.Level(off, h, w, h * w * C_))
        off += B * h * w * C_
    return lv, off


def main():
    B, Ci, Co = 16, 256, 256
    st = C.c_void_p(stream_ptr())
    w = torch.randn(Co, 3, 3, Ci, device="cuda") / (9 * Ci) ** 0.5
    bias = torch.randn(Co, device="cuda")
    shapes = {"P3_2 (38^2)": [(38, 38)], "head big (38^2 + 19^2)": [(38, 38), (19, 19)], "head all six levels": [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)],
              "head small (10^2 .. 1)": [(10, 10), (5, 5), (3, 3), (1, 1)], "19^2": [(19, 19)]}
    for name, sizes in shapes.items():
        lv, n = levels(B, Ci, sizes)
        x = torch.randn(n, device="cuda").clamp_min(0)
        y = torch.empty(n, device="cuda")
        src, dst = ops.TView(x, B, Ci, Ci, lv), ops.TView(y, B, Co, Co, lv)
        fl = sum(2.0 * B * h * ww * Co * 9 * Ci for h, ww in sizes)
        # F(4x4)
        U4 = torch.empty(int(lib.zsg_wino4_u_elems(Ci, Co)), device="cuda")
        j = ops.WinoJobs()
        j.add(w.data_ptr(), U4.data_ptr(), Co, Ci, 9 * Ci, Ci, 0)
        blob = j.finish("cuda")
        check(lib.zsg_wino4_weights(blob.data_ptr(), 1, j.blocks, st), "u4")
        d4 = ops.fwd_desc(src, dst, Ci, Co, 3, 1, 1, 1, wC=Ci, relu=True)
        t4 = timeit(lambda: check(lib.zsg_conv_wino4(C.byref(d4), x.data_ptr(), U4.data_ptr(), y.data_ptr(), bias.data_ptr(), None, None, st), "w4"))
        tiles4 = sum(B * ((h + 3) // 4) * ((ww + 3) // 4) for h, ww in sizes)
        # F(2x2): best of its tile candidates
        U2 = torch.empty(int(lib.zsg_wino_u_elems(Ci, Co)), device="cuda")
        j2 = ops.WinoJobs()
        j2.add(w.data_ptr(), U2.data_ptr(), Co, Ci, 9 * Ci, Ci, 0)
        j2.finish("cuda")
        j2.launch(stream_ptr())
        best = (1e9, 0)
        for hint in (ops.tile_hint(64, 64, 1), ops.tile_hint(32, 64, 1), ops.tile_hint(64, 64, 1, 1), ops.tile_hint(32, 64, 1, 1)):
            d2 = ops.fwd_desc(src, dst, Ci, Co, 3, 1, 1, 1, wC=Ci, relu=True, tile_hint=hint)
            t2 = timeit(lambda: check(lib.zsg_conv_wino(C.byref(d2), x.data_ptr(), U2.data_ptr(), y.data_ptr(), bias.data_ptr(), None, None, None, st), "w2"))
            best = min(best, (t2, hint))
        blocks4 = ((tiles4 + 31) // 32) * ((Co + 63) // 64)
        print(f"{name:28s} F(4x4): {t4:7.1f} us ({blocks4} blocks, {fl / t4 * 1e-6:6.1f} TF/s algorithmic, {fl / t4 * 1e-6 / 4:5.1f} executed)   "
              f"F(2x2) best: {best[0]:7.1f} us (hint {best[1]:#x}, {fl / best[0] * 1e-6:6.1f} algorithmic, {fl / best[0] * 1e-6 * 4 / 9:5.1f} executed)", flush=True)


if __name__ == "__main__":
    main()
