"""Developer tool (GPU box): zsg_bn_relu_maxpool_bwd on the stem shape (fixed seed) -> file, so that two builds of libzsg (ZSG_LIB_PATH)
can be compared bit for bit; prints the call's time.   python tools/dev_stem_bits.py <out.pt> [other.pt]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr

res = {}
for (B, C, H, W) in ((16, 64, 150, 150), (2, 64, 37, 41)):
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B, H, W, C, device="cuda", generator=g) * 1.5 + 0.4
    mean, invstd = x.view(-1, C).mean(0), 1 / torch.sqrt(x.view(-1, C).var(0, unbiased=False) + 1e-5)
    gam, bet = torch.rand(C, device="cuda", generator=g) + 0.5, 0.3 * torch.randn(C, device="cuda", generator=g)
    out = torch.empty(B, Ho, Wo, C, device="cuda")
    idx = torch.zeros(B * Ho * Wo * C, dtype=torch.uint8, device="cuda")
    st = stream_ptr()
    check(lib.zsg_bn_relu_maxpool_fwd(x.data_ptr(), B, H, W, C, mean.data_ptr(), invstd.data_ptr(), gam.data_ptr(), bet.data_ptr(), 3, 2, 1, Ho, Wo,
                                      out.data_ptr(), idx.data_ptr(), st), "fwd")
    gy = torch.randn(B, Ho, Wo, C, device="cuda", generator=g)
    dx = torch.empty(B, H, W, C, device="cuda")
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    wsb = lib.zsg_bn_workspace_bytes(B * Ho * Wo, C)
    ws = torch.empty(wsb // 4 + 16, device="cuda")

    def run():
        check(lib.zsg_bn_relu_maxpool_bwd(gy.data_ptr(), idx.data_ptr(), x.data_ptr(), B, H, W, C, mean.data_ptr(), invstd.data_ptr(), gam.data_ptr(),
                                          bet.data_ptr(), 3, 2, 1, Ho, Wo, dx.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), 0, ws.data_ptr(), wsb, st), "bwd")
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        run()
    torch.cuda.synchronize()
    print(f"B={B} {H}x{W}: zsg_bn_relu_maxpool_bwd {(time.perf_counter() - t0) / 30 * 1e6:.1f} us per call")
    res[f"{B}x{H}"] = (dx.cpu(), dgam.cpu(), dbet.cpu())
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    for k in res:
        print(k, "bit-identical" if all(torch.equal(a, b) for a, b in zip(res[k], other[k])) else "DIFFERENT")
torch.save(res, sys.argv[1])
