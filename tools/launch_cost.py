"""Developer tool: host cost of one launch through the C ABI (ctypes + library prologue + hipLaunchKernel)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import _lib as L, ops


def timeit(name, fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:28s} host {1e6 * (t1 - t0) / n:7.2f} us/launch   wall {1e6 * (t2 - t0) / n:7.2f} us/launch")


def main():
    st = C.c_void_p(L.stream_ptr())
    x = torch.zeros(1024, device="cuda")
    y = torch.zeros(1024, device="cuda")
    px, py = C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())
    timeit("relu_fwd (1 kernel)", lambda: L.lib.zsg_relu_fwd(px, 1024, py, st))
    timeit("memset_f32", lambda: L.lib.zsg_memset_f32(px, 1024, C.c_float(0.0), st))
    B, H, W, Ci, Co = 1, 8, 8, 64, 64
    src = torch.zeros(B * H * W * Ci, device="cuda")
    wt = torch.zeros(Co * Ci, device="cuda")
    out = torch.zeros(B * H * W * Co, device="cuda")
    sv = ops.TView(src, B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
    ov = ops.TView(out, B, Co, Co, [ops.Level(0, H, W, H * W * Co)])
    d = ops.fwd_desc(sv, ov, Ci, Co, 1, 1, 0, 1, wC=Ci)
    d.tile_hint = ops.tile_hint(64, 64, 1)
    args = ops.marshal(L.lib.zsg_conv_igemm, (d, src, wt, out, None, None, None, None))
    timeit("conv_igemm 64x64 tiny", lambda: L.lib.zsg_conv_igemm(*args, st))
    a, b = torch.cuda.Event(), torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def evpair():
        a.record(main_s)
        b.wait_event(a)
    timeit("event record + wait", evpair)
    timeit("torch add_ (reference)", lambda: x.add_(1.0))


if __name__ == "__main__":
    main()
