"""Developer tool (GPU box): zsg_conv_wino4 launch time against the number of 8-channel chunks (P3_2's geometry): t = fixed + per-chunk."""
import ctypes as C, os, sys, torch
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), os.path.dirname(os.path.abspath(__file__))]
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr
from bind import bind
bind()
from bench_wino4 import timeit, levels
B, Co = 16, 256
st = C.c_void_p(stream_ptr())
for Ci in (8, 64, 128, 256, 512):
    lv, n = levels(B, Ci, [(38, 38)])
    lvo, no = levels(B, Co, [(38, 38)])
    x = torch.randn(n, device="cuda"); y = torch.empty(no, device="cuda")
    w = torch.randn(Co, 3, 3, Ci, device="cuda")
    U4 = torch.empty(int(lib.zsg_wino4_u_elems(Ci, Co)), device="cuda")
    j = ops.WinoJobs(); j.add(w.data_ptr(), U4.data_ptr(), Co, Ci, 9 * Ci, Ci, 0); blob = j.finish("cuda")
    check(lib.zsg_wino4_weights(blob.data_ptr(), 1, j.blocks, st), "u4")
    d = ops.fwd_desc(ops.TView(x, B, Ci, Ci, lv), ops.TView(y, B, Co, Co, lvo), Ci, Co, 3, 1, 1, 1, wC=Ci)
    t = timeit(lambda: check(lib.zsg_conv_wino4(C.byref(d), x.data_ptr(), U4.data_ptr(), y.data_ptr(), None, None, None, st), "w4"))
    print(f"Ci={Ci:4d} chunks={Ci // 8:3d}  {t:7.1f} us", flush=True)
