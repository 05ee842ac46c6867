import ctypes as C, os, sys, torch
sys.path.insert(0, os.getcwd())
from tools.bench_conv import SHAPES, timeit
from zsgnet_pytorch_amd import ops
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr
WS = torch.empty(128 << 20, device="cuda")
st = stream_ptr()
for name in ("l1_conv2", "l1_conv1"):
    _, B, Ci, Co, H, W, k, s, p = [x for x in SHAPES if x[0] == name][0]
    Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    x = torch.randn(B, H, W, Ci, device="cuda"); dy = torch.randn(B, Ho, Wo, Co, device="cuda"); dw = torch.zeros(Co, k, k, Ci, device="cuda")
    xv = ops.TView(x.view(-1), B, Ci, Ci, [ops.Level(0, H, W, H * W * Ci)])
    dyv = ops.TView(dy.view(-1), B, Co, Co, [ops.Level(0, Ho, Wo, Ho * Wo * Co)])
    gf = 2.0 * B * Ho * Wo * Co * Ci * k * k / 1e9
    print(name, f"N={Co} ncols={k*k*Ci} {gf:.2f} GF")
    for bn in (64, 128, 255):
        line = f"  64x{256 if bn == 255 else bn}:"
        for sp in (16, 32, 64, 96, 128, 192, 255):
            d = ops.fwd_desc(xv, dyv, Ci, Co, k, s, p, 1, wC=Ci, tile_hint=ops.tile_hint(64, bn, sp))
            ms = timeit(lambda: check(lib.zsg_conv_wgrad(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st)))
            line += f" s{sp}:{gf / ms:6.1f}"
        print(line, flush=True)
