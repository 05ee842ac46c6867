"""The bf16x6 matrix path (fp32 operands split exactly into three bf16 terms, six bf16 MFMAs per product block, fp32
accumulation) must be an fp32-grade computation: measured against an fp64 reference its error may not exceed the native
fp32-MFMA path's by more than the stated factor, on every tile variant and on the awkward shapes (K tails, strides, dilation,
split-K, fused epilogues)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import Z, dev, nhwc, ohwi, pad4, view_of  # noqa: F401

pytestmark = pytest.mark.gpu

BX = 1 << 26
# (BM, BN, two K groups)
VARIANTS = [(64, 64, 0), (64, 64, 1), (128, 64, 0), (128, 64, 1), (128, 128, 0), (128, 128, 1)]

CASES = [
    # B, Ci, Co, H, W, k, s, p, d, bias, relu
    (2, 64, 256, 19, 19, 1, 1, 0, 1, False, False),
    (2, 300, 96, 10, 13, 1, 1, 0, 1, True, True),        # K tail (300 = 9 x 32 + 12), N tail
    (2, 128, 128, 21, 21, 3, 2, 1, 1, False, False),
    (1, 516, 256, 10, 10, 3, 1, 1, 1, True, True),
    (1, 64, 96, 12, 12, 3, 1, 6, 6, True, False),        # dilation 6
    (3, 2048, 64, 5, 5, 1, 1, 0, 1, False, False),       # long K
    (2, 40, 45, 9, 7, 3, 1, 1, 1, True, False),
]


def rel_err(got, ref):
    return float((got.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("case", CASES, ids=[f"x{i}" for i in range(len(CASES))])
def test_bf16x6_matches_fp32_grade(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, k, s, p, d, bias, relu = case
    g = torch.Generator().manual_seed(11 + Ci + Co)
    # wide dynamic range + a few exact powers of two and tiny values: the split has to be exact for all of them
    x = torch.randn(B, Ci, H, W, generator=g) * torch.exp(2.0 * torch.randn(B, Ci, H, W, generator=g))
    x.view(-1)[:8] = torch.tensor([1.0, -2.0, 2.0 ** -20, 3.0 * 2.0 ** -30, 1.0 + 2.0 ** -23, -(1.0 - 2.0 ** -24), 65504.0, 1e-30])
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if bias else None
    xr = x.double().requires_grad_()
    y64 = F.conv2d(xr, w.double(), b.double() if bias else None, s, p, d)
    pre64 = y64
    if relu:
        y64 = F.relu(y64)
    Ho, Wo = y64.shape[2:]
    gy = torch.randn(y64.shape, generator=g)
    gpre = gy * (pre64 > 0) if relu else gy
    y64.backward(gy.double())
    y64, dx64 = y64.detach(), xr.grad
    cp, Cop = pad4(Ci), pad4(Co)
    st = L.stream_ptr()
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    bd = dev(b) if bias else None
    src = view_of(ops, xd, B, H, W, cp)
    dyd = dev(nhwc(gpre.float(), Cop))
    dyv = view_of(ops, dyd, B, Ho, Wo, Cop)
    wt = torch.empty((cp, k, k, Cop), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, k * k, cp, Cop, st), "transpose_w")

    def fwd(hint):
        out = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
        desc = ops.fwd_desc(src, view_of(ops, out, B, Ho, Wo, Co), cp, Co, k, s, p, d, wC=cp, relu=relu and ((hint >> 16) & 0xff) <= 1, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(desc), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr() if bias else None, None, None, None, st), "igemm")
        return out.permute(0, 3, 1, 2)

    def dgrad(hint):
        dx = torch.zeros((B, H, W, cp), device="cuda")
        desc = ops.dgrad_desc(dyv, view_of(ops, dx, B, H, W, cp), Cop, cp, k, s, p, d, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(desc), dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st), "dgrad")
        return dx[..., :Ci].permute(0, 3, 1, 2)

    e_nat_f = rel_err(fwd(ops.tile_hint(64, 64, 1)), y64)
    e_nat_d = rel_err(dgrad(ops.tile_hint(64, 64, 1)), dx64)
    assert e_nat_f < 1e-5 and e_nat_d < 1e-5
    for bm, bn, w8 in VARIANTS:
        h = ops.tile_hint(bm, bn, 1, w8) | BX
        e_f = rel_err(fwd(h), y64)
        e_d = rel_err(dgrad(h), dx64)
        # fp32-grade: no worse than 1.5x the native fp32-MFMA error (+ one ulp of slack for tiny problems)
        assert e_f <= 1.5 * e_nat_f + 1.2e-7, f"fwd {bm}x{bn}/{w8}: bf16x6 err {e_f:.3g} vs native {e_nat_f:.3g}"
        assert e_d <= 1.5 * e_nat_d + 1.2e-7, f"dgrad {bm}x{bn}/{w8}: bf16x6 err {e_d:.3g} vs native {e_nat_d:.3g}"
    if not relu and s == 1:
        for sp in (2, 5):
            e_f = rel_err(fwd(ops.tile_hint(64, 64, sp) | BX), y64)
            assert e_f <= 1.5 * e_nat_f + 2.4e-7, f"split-K {sp}: {e_f:.3g} vs {e_nat_f:.3g}"
    torch.cuda.synchronize()


def test_bf16x6_fused_bn_statistics_and_mask(Z):
    """the epilogue (BatchNorm partials, accumulate + ReLU mask) is shared with the native path: same results within fp32 rounding"""
    L, ops = Z
    g = torch.Generator().manual_seed(5)
    B, Ci, Co, H, W = 4, 256, 128, 19, 19
    x, w = torch.randn(B, H, W, Ci, generator=g), torch.randn(Co, 1, 1, Ci, generator=g) / 16
    xd, wd = dev(x), dev(w)
    st = L.stream_ptr()
    src = view_of(ops, xd, B, H, W, Ci)
    res = {}
    for name, hint in (("nat", ops.tile_hint(128, 64, 1)), ("bx", ops.tile_hint(128, 64, 1) | BX), ("bx8", ops.tile_hint(128, 64, 1, 1) | BX)):
        out = torch.empty((B, H, W, Co), device="cuda")
        chunks = (B * H * W + 127) // 128
        part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
        d = ops.fwd_desc(src, view_of(ops, out, B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "igemm+stats")
        prev, mask = dev(torch.randn(B, H, W, Co, generator=torch.Generator().manual_seed(1))), dev(torch.randn(B, H, W, Co, generator=torch.Generator().manual_seed(2)))
        L.check(L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), prev.data_ptr(), None, prev.data_ptr(), mask.data_ptr(), None, st), "igemm+acc+mask")
        res[name] = (out.cpu(), part.cpu(), prev.cpu())
    for name in ("bx", "bx8"):
        for a, b_ in zip(res[name], res["nat"]):
            assert torch.allclose(a, b_, rtol=2e-5, atol=2e-5 * float(b_.abs().max()))
        assert torch.equal(res[name][2] == 0, res["nat"][2] == 0)          # the mask zeros exactly the same elements
