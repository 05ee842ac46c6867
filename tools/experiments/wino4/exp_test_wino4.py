"""EXPERIMENT (tools/experiments/README.md).  Winograd F(4x4,3x3) convolution (wino4.hip: zsg_wino4_weights + zsg_conv_wino4) through the raw C ABI, forward and data
gradient, against torch in float64 — the pyramid-output / shared-head convolutions of fpn_resnet.py:157-172 and mdl.py:211-244.
Numeric gate of VERDICT r04 item 7: op-level error against fp64 <= 1e-5 of the output range on head-shaped operands (C = 256, post-ReLU
activations, He-scaled weights); the general cases are held to 5e-5 of the output range (random-sign inputs of smaller fan-in).
Edge cases as the reference has them: pyramid levels 38 / 19 / 10 / 5 / 3 / 1 in ONE launch (segments), sizes that are no multiple of
4, the 45-channel head output (scalar store path), C % 8 == 4, bias / ReLU / accumulate / ReLU mask epilogues."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))]
from test_gpu_ops import Z, dev, nhwc, ohwi, pad4, view_of  # noqa: E402,F401
from bind import bind  # noqa: E402

bind()      # (the experiment's three symbols on the loaded library: a build made with EXPERIMENTS=1, see ../README.md)

pytestmark = pytest.mark.gpu


def make_u4(L, ops, wd, N, Cred, row_ld, tap_ld, flip, wc0=0):
    U = torch.full((int(L.lib.zsg_wino4_u_elems(Cred, N)),), float("nan"), device="cuda")
    jobs = ops.WinoJobs()
    jobs.add(wd.data_ptr() + 4 * wc0, U.data_ptr(), N, Cred, row_ld, tap_ld, flip)
    blob = jobs.finish("cuda")
    L.check(L.lib.zsg_wino4_weights(blob.data_ptr(), 1, jobs.blocks, C.c_void_p(L.stream_ptr())), "wino4_weights")
    return U


CASES = [
    # B, Ci, Co, H, W, bias, relu
    (2, 64, 64, 19, 19, False, False),
    (2, 64, 128, 20, 17, True, True),
    (3, 128, 64, 7, 10, False, False),
    (2, 48, 256, 10, 10, True, False),
    (2, 256, 45, 10, 10, True, False),        # ragged N: scalar store path
    (1, 260, 256, 5, 5, True, True),          # C % 8 == 4
    (2, 256, 256, 1, 1, True, True),
    (2, 8, 64, 9, 9, False, False),           # one chunk
    (2, 12, 64, 6, 11, True, False),
    (16, 256, 256, 38, 38, True, True),       # the head's largest level at the bench shape
    (4, 256, 256, 19, 19, True, False),
]


@pytest.mark.parametrize("case", CASES, ids=[f"f{i}" for i in range(len(CASES))])
def test_wino4_fwd_dgrad(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, bias, relu = case
    g = torch.Generator().manual_seed(17 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g) if bias else None
    xr = x.double().requires_grad_()
    y_ref = F.conv2d(xr, w.double(), b.double() if bias else None, 1, 1)
    if relu:
        y_ref = F.relu(y_ref)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy.double())
    cp = pad4(Ci)
    st = L.stream_ptr()
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    U = make_u4(L, ops, wd, Co, cp, 9 * cp, cp, False)
    assert not torch.isnan(U).any()
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    src, ov = view_of(ops, xd, B, H, W, cp), view_of(ops, out, B, H, W, Co)
    desc = ops.fwd_desc(src, ov, cp, Co, 3, 1, 1, 1, wC=cp, relu=relu)
    bd = dev(b) if bias else None
    L.check(L.lib.zsg_conv_wino4(C.byref(desc), xd.data_ptr(), U.data_ptr(), out.data_ptr(), bd.data_ptr() if bias else None, None, None, st), "wino4")
    yr = y_ref.detach()
    e = float((out.permute(0, 3, 1, 2).double().cpu() - yr).abs().max()) / float(yr.abs().max())
    print(f"F(4x4,3x3) forward: max error / max|y| = {e:.2e}")
    assert e <= 5e-5, e
    # two launches are bit-identical (no atomics, fixed order)
    out2 = torch.full_like(out, float("nan"))
    d2 = ops.fwd_desc(src, view_of(ops, out2, B, H, W, Co), cp, Co, 3, 1, 1, 1, wC=cp, relu=relu)
    L.check(L.lib.zsg_conv_wino4(C.byref(d2), xd.data_ptr(), U.data_ptr(), out2.data_ptr(), bd.data_ptr() if bias else None, None, None, st), "wino4")
    assert torch.equal(out, out2)
    # data gradient: the same kernel on dy with the rotated, transposed filter image
    gpre = gy * (yr > 0) if relu else gy
    Cop = pad4(Co)
    dyd = dev(nhwc(gpre.float(), Cop))
    dyv = view_of(ops, dyd, B, H, W, Cop)
    wt = torch.full((cp, 3, 3, Cop), float("nan"), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, 9, cp, Cop, st), "transpose_w")
    Ut = make_u4(L, ops, wt, cp, Cop, 9 * Cop, Cop, True)
    dx = torch.full((B, H, W, cp), float("nan"), device="cuda")
    dxv = view_of(ops, dx, B, H, W, cp)
    ddesc = ops.dgrad_desc(dyv, dxv, Cop, cp, 3, 1, 1, 1)
    L.check(L.lib.zsg_conv_wino4(C.byref(ddesc), dyd.data_ptr(), Ut.data_ptr(), dx.data_ptr(), None, None, None, st), "wino4 dgrad")
    gr = xr.grad
    e = float((dx[..., :Ci].permute(0, 3, 1, 2).double().cpu() - gr).abs().max()) / float(gr.abs().max())
    print(f"F(4x4,3x3) data gradient: max error / max|dx| = {e:.2e}")
    assert e <= 5e-5, e
    if cp > Ci:
        assert float(dx[..., Ci:].abs().max()) == 0.0
    # accumulate + relu-mask epilogue: out = (prev + acc) * (mask > 0)
    prev = torch.randn(B, H, W, cp, generator=g)
    maskv = torch.randn(B, H, W, cp, generator=g)
    acc = dev(prev.clone())
    mk = dev(maskv)
    adesc = ops.dgrad_desc(dyv, view_of(ops, acc, B, H, W, cp), Cop, cp, 3, 1, 1, 1)
    L.check(L.lib.zsg_conv_wino4(C.byref(adesc), dyd.data_ptr(), Ut.data_ptr(), acc.data_ptr(), None, acc.data_ptr(), mk.data_ptr(), st), "wino4 acc+mask")
    want = (prev.cuda() + dx) * (mk > 0)
    assert torch.allclose(acc, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_wino4_gate_head_operands(Z):
    """VERDICT r04 item 7's op-level gate on head-like data: C = 256 post-ReLU activations, He-scaled weights, 38x38 — error against
    fp64 <= 1e-5 of the output range (measured on the CPU in emulated fp32: 6.5e-6, tools/wino_f4_error.py)."""
    L, ops = Z
    B, Ci, Co, H, W = 4, 256, 256, 38, 38
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g).clamp_min(0)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5
    yr = F.conv2d(x.double(), w.double(), None, 1, 1)
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    U = make_u4(L, ops, wd, Co, Ci, 9 * Ci, Ci, False)
    out = torch.empty(B, H, W, Co, device="cuda")
    desc = ops.fwd_desc(view_of(ops, xd, B, H, W, Ci), view_of(ops, out, B, H, W, Co), Ci, Co, 3, 1, 1, 1, wC=Ci)
    L.check(L.lib.zsg_conv_wino4(C.byref(desc), xd.data_ptr(), U.data_ptr(), out.data_ptr(), None, None, None, L.stream_ptr()), "wino4")
    e = float((out.permute(0, 3, 1, 2).double().cpu() - yr).abs().max()) / float(yr.abs().max())
    print(f"gate: F(4x4,3x3) max error / max|y| = {e:.2e} (bound 1e-5)")
    assert e <= 1e-5, e


def test_wino4_pyramid_levels_in_one_launch(Z):
    """The shared head's launch shape: six pyramid levels (38, 19, 10, 5, 3, 1) packed level-major in one buffer, one segment each,
    shared weights, bias + ReLU — against torch level by level."""
    L, ops = Z
    B, Ci, Co = 2, 64, 64
    sizes = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]
    g = torch.Generator().manual_seed(3)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    xs = [torch.randn(B, Ci, h, ww, generator=g) for h, ww in sizes]
    lv, off = [], 0
    for (h, ww) in sizes:
        lv.append(ops.Level(off, h, ww, h * ww * Ci))
        off += B * h * ww * Ci
    buf = torch.zeros(off, device="cuda")
    for x, l in zip(xs, lv):
        buf[l.off:l.off + x.numel()] = dev(nhwc(x)).reshape(-1)
    outb = torch.full((off,), float("nan"), device="cuda")
    src = ops.TView(buf, B, Ci, Ci, lv)
    dst = ops.TView(outb, B, Co, Co, lv)
    wd, bd = dev(ohwi(w)), dev(b)
    U = make_u4(L, ops, wd, Co, Ci, 9 * Ci, Ci, False)
    desc = ops.fwd_desc(src, dst, Ci, Co, 3, 1, 1, 1, wC=Ci, relu=True)
    L.check(L.lib.zsg_conv_wino4(C.byref(desc), buf.data_ptr(), U.data_ptr(), outb.data_ptr(), bd.data_ptr(), None, None, L.stream_ptr()), "wino4 levels")
    assert not torch.isnan(outb).any()
    for x, l in zip(xs, lv):
        yr = F.relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1))
        got = outb[l.off:l.off + x.numel()].view(B, l.H, l.W, Co).permute(0, 3, 1, 2).double().cpu()
        assert float((got - yr).abs().max()) <= 5e-5 * float(yr.abs().max()) + 1e-6, (l.H, l.W)
