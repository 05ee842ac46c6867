"""Developer micro-benchmark: Winograd F(2x2,3x3) launches vs the direct implicit GEMM on the network's 3x3/s1 shapes.
usage: python tools/bench_wino.py [shape names]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

# name, B, Cin, Cout, sizes
SHAPES = [
    ("head", 16, 256, 256, [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]),
    ("P3_2", 16, 256, 256, [(38, 38)]),
    ("P4_2", 16, 256, 256, [(19, 19)]),
    ("l1_conv2", 16, 64, 64, [(75, 75)]),
    ("l2_conv2", 16, 128, 128, [(38, 38)]),
    ("l3_conv2", 16, 256, 256, [(19, 19)]),
    ("l4_conv2", 16, 512, 512, [(10, 10)]),
    ("head5", 16, 256, 45, [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]),
]


WS = torch.empty(64 << 20, device="cuda") if torch.cuda.is_available() else None


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    only = sys.argv[1:] or None
    st = stream_ptr()
    for name, B, Ci, Co, sizes in SHAPES:
        if only and name not in only:
            continue
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
        t_w = timeit(lambda: jobs.launch(st))
        gf = 2.0 * sum(B * h * ww for h, ww in sizes) * Co * 9 * Ci / 1e9
        line = f"{name:10s} {gf:7.2f} GF | U-transform {t_w * 1e3:6.1f} us |"
        for hint in (0, 64 | (64 << 8), 128 | (64 << 8), 128 | (128 << 8) | (1 << 24)):
            d = ops.fwd_desc(src, out, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=hint)
            t = timeit(lambda: check(lib.zsg_conv_igemm(C.byref(d), x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, st), "ig"))
            line += f" ig[{hint & 0xff}x{(hint >> 8) & 0xff}] {t * 1e3:6.1f}us {gf / t:6.1f}"
        line += " |"
        dense = len(sizes) == 1
        cands = [(64, 64, 1, 0), (32, 64, 1, 0), (32, 32, 1, 0), (64, 64, 1, 1), (32, 64, 1, 1), (64, 32, 1, 1), (32, 32, 1, 1)]
        if dense:
            cands += [(64, 64, 2, 0), (64, 64, 4, 0), (64, 64, 2, 1), (32, 64, 2, 1), (64, 64, 4, 1)]
        for TB, BN, sp, ps4 in cands:
            d = ops.fwd_desc(src, out, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=TB | (BN << 8) | (sp << 16) | (ps4 << 24))
            t = timeit(lambda: check(lib.zsg_conv_wino(C.byref(d), x.data_ptr(), U.data_ptr(), y.data_ptr(), None, None, None, None, st), "wino"))
            line += f" wn[{TB}x{BN}/{sp}{'q' if ps4 else ''}] {t * 1e3:6.1f}us {gf / t:6.1f}"
        # weight gradient: direct (heuristic / a few splits) vs Winograd F(3x3,2x2)
        dy = torch.randn(oo, device="cuda")
        dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        dyv = ops.TView(dy, B, Co, Co, lv_out)
        if Co % 4 == 0:
            line += " || wgrad"
            best = 1e9
            for hint in (0, ops.tile_hint(128, 128, 4, 1, 0), ops.tile_hint(128, 128, 8, 1, 0), ops.tile_hint(64, 64, 32), ops.tile_hint(128, 64, 16)):
                d = ops.fwd_desc(src, dyv, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=hint)
                t = timeit(lambda: check(lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wg"))
                best = min(best, t)
            line += f" direct-best {best * 1e3:6.1f}us {gf / best:6.1f} |"
            nmn = ((Co + 63) // 64) * ((Ci + 63) // 64)
            for target in (128, 256, 384, 512):
                sp = max(1, min(target // nmn, 255))
                d = ops.fwd_desc(src, dyv, Ci, Co, 3, 1, 1, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, sp))
                t = timeit(lambda: check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgw"))
                line += f" ww[/{sp}] {t * 1e3:6.1f}us {gf / t:6.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
