"""GPU parity of the Winograd F(2x2,3x3) convolution (zsg_conv_wino + zsg_wino_weights, called through the C ABI)
against torch-CPU fp32 conv2d.  Tolerance: rel 3e-4 / abs 3e-4 of the output scale — fp32 throughout; the transforms
add a handful of roundings per output (constants 0, +-1, +-1/2), same class as the direct kernel's 2e-4."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_gpu_ops import Z, assert_close, dev, nhwc, ohwi, pad4, view_of  # noqa: E402,F401


def make_u(L, ops, wd, N, Cred, row_ld, tap_ld, flip, wc0=0):
    U = torch.full((int(L.lib.zsg_wino_u_elems(Cred, N)),), float("nan"), device="cuda")
    jobs = ops.WinoJobs()
    jobs.add(wd.data_ptr() + 4 * wc0, U.data_ptr(), N, Cred, row_ld, tap_ld, flip)
    jobs.finish("cuda")
    jobs.launch(L.stream_ptr())
    return U


WINO_CASES = [
    # B, Ci, Co, H, W, bias, relu, TB, BN, splits
    (2, 64, 64, 19, 19, False, False, 64, 64, 1),
    (2, 64, 128, 20, 17, True, True, 64, 64, 1),
    (3, 128, 64, 7, 10, False, False, 32, 64, 1),
    (2, 48, 256, 10, 10, True, False, 64, 32, 1),
    (2, 256, 45, 10, 10, True, False, 64, 64, 1),
    (1, 516, 256, 5, 5, True, True, 32, 32, 1),
    (2, 256, 256, 1, 1, True, True, 32, 32, 1),
    (2, 128, 128, 9, 9, False, False, 32, 32, 3),
    (2, 512, 96, 10, 10, True, False, 64, 64, 4),
    (16, 64, 64, 38, 38, False, False, 64, 64, 1),
    # four position groups (tile_hint bit 24: splits field + 256)
    (2, 64, 128, 20, 17, True, True, 64, 64, 1 + 256),
    (3, 128, 64, 7, 10, False, False, 32, 64, 1 + 256),
    (2, 48, 256, 10, 10, True, False, 64, 32, 1 + 256),
    (1, 516, 256, 5, 5, True, True, 32, 32, 1 + 256),
    (2, 512, 96, 10, 10, True, False, 64, 64, 4 + 256),
    (16, 64, 64, 38, 38, False, False, 64, 64, 1 + 256),
    # pipeline fill / drain and channel tails of the K loop: 1, 2, 3, 5 chunks of 8 channels, C % 8 == 4 (last chunk half dead)
    (2, 8, 64, 9, 9, False, False, 32, 64, 1 + 256),
    (2, 16, 64, 9, 9, True, True, 32, 64, 1 + 256),
    (2, 24, 128, 7, 10, False, False, 64, 64, 1 + 256),
    (3, 40, 64, 7, 10, True, False, 32, 64, 1),
    (2, 12, 64, 9, 9, True, False, 32, 64, 1 + 256),
    (2, 20, 64, 9, 9, False, False, 64, 64, 1 + 256),
    (1, 516, 256, 5, 5, True, True, 32, 64, 1 + 256),
    (2, 256, 45, 10, 10, True, False, 32, 64, 1 + 256),
    (2, 512, 96, 10, 10, True, False, 32, 64, 4 + 256),
    (4, 256, 256, 19, 19, False, False, 32, 64, 1 + 256),
]


@pytest.mark.parametrize("case", WINO_CASES, ids=[f"w{i}" for i in range(len(WINO_CASES))])
def test_wino_fwd_dgrad(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, bias, relu, TB, BN, splits = case
    ps4, splits = (splits >> 8) & 1, splits & 0xff
    g = torch.Generator().manual_seed(7 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone()
    y_ref = F.conv2d(xr, wr, b, 1, 1)
    if relu:
        y_ref = F.relu(y_ref)
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    cp = pad4(Ci)
    st = L.stream_ptr()
    hint = TB | (BN << 8) | (splits << 16) | (ps4 << 24)
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    U = make_u(L, ops, wd, Co, cp, 9 * cp, cp, False)
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    src, ov = view_of(ops, xd, B, H, W, cp), view_of(ops, out, B, H, W, Co)
    desc = ops.fwd_desc(src, ov, cp, Co, 3, 1, 1, 1, wC=cp, relu=relu and splits == 1, tile_hint=hint)
    bd = dev(b) if bias else None
    L.check(L.lib.zsg_conv_wino(C.byref(desc), xd.data_ptr(), U.data_ptr(), out.data_ptr(), bd.data_ptr() if bias else None, None,
                                None, None, st), "wino")
    got = F.relu(out) if (relu and splits > 1) else out
    assert_close(got.permute(0, 3, 1, 2), y_ref, 3e-4, 3e-4, "wino fwd")
    if not bias and not relu and splits == 1 and Co % 4 == 0:
        tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
        chunks = (tiles + TB - 1) // TB
        part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
        L.check(L.lib.zsg_conv_wino(C.byref(desc), xd.data_ptr(), U.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "wino+stats")
        yf = y_ref.detach().permute(0, 2, 3, 1).reshape(-1, Co).double()
        assert_close(part[:, 0].double().sum(0), yf.sum(0), 1e-4, 1e-4 * float(yf.abs().sum(0).max()), "bn partial sums")
        assert_close(part[:, 1].double().sum(0), (yf * yf).sum(0), 2e-4, 1e-6, "bn partial sums of squares")
    # data gradient: the same kernel on dy with the rotated, transposed filter image
    gpre = gy * (y_ref > 0) if relu else gy
    Cop = pad4(Co)
    dyd = dev(nhwc(gpre, Cop))
    dyv = view_of(ops, dyd, B, H, W, Cop)
    wt = torch.full((cp, 3, 3, Cop), float("nan"), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, 9, cp, Cop, st), "transpose_w")
    Ut = make_u(L, ops, wt, cp, Cop, 9 * Cop, Cop, True)
    dx = torch.full((B, H, W, cp), float("nan"), device="cuda")
    dxv = view_of(ops, dx, B, H, W, cp)
    ddesc = ops.dgrad_desc(dyv, dxv, Cop, cp, 3, 1, 1, 1, tile_hint=hint)
    L.check(L.lib.zsg_conv_wino(C.byref(ddesc), dyd.data_ptr(), Ut.data_ptr(), dx.data_ptr(), None, None, None, None, st), "wino dgrad")
    assert_close(dx[..., :Ci].permute(0, 3, 1, 2), xr.grad, 5e-4, 5e-4 * float(xr.grad.abs().max()), "wino dgrad")
    if cp > Ci:
        assert float(dx[..., Ci:].abs().max()) == 0.0
    if splits > 1:       # (split-K: the mask applies to the new term only, as in zsg_conv_igemm — never combined by the plans)
        return
    # accumulate + relu-mask epilogue: out = (prev + acc) * (mask > 0)
    prev = torch.randn(B, H, W, cp, generator=g)
    mask = torch.randn(B, H, W, cp, generator=g)
    dx2, maskd = dev(prev), dev(mask)
    L.check(L.lib.zsg_conv_wino(C.byref(ddesc), dyd.data_ptr(), Ut.data_ptr(), dx2.data_ptr(), None, dx2.data_ptr(), maskd.data_ptr(), None, st), "wino dgrad+")
    ref2 = (prev[..., :Ci] + xr.grad.permute(0, 2, 3, 1)) * (mask[..., :Ci] > 0)
    assert_close(dx2[..., :Ci], ref2, 5e-4, 5e-4 * float(ref2.abs().max()), "wino dgrad accumulate+mask")
    torch.cuda.synchronize()


def test_wino_multilevel_window(Z):
    """grouped launch over pyramid levels with a channel window of the weight (head conv0: features only) and an
    additive map, output scattered into a [B, P, N] buffer"""
    L, ops = Z
    g = torch.Generator().manual_seed(11)
    B, Cf, Cw, Co = 2, 64, 12, 64
    Ct = Cf + Cw
    sizes = [(7, 7), (4, 4), (3, 3), (2, 2), (1, 1)]
    w = torch.randn(Co, Ct, 3, 3, generator=g) / 24
    b = torch.randn(Co, generator=g)
    xs = [torch.randn(B, Cf, h, ww, generator=g) for h, ww in sizes]
    P = sum(h * ww for h, ww in sizes)
    addm = torch.randn(B, P, Co, generator=g)
    refs = [F.conv2d(x, w[:, :Cf], b, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Co) for x in xs]
    ref = F.relu(torch.cat(refs, dim=1) + addm)
    packed = torch.cat([nhwc(x).reshape(-1) for x in xs]).cuda()
    lv_in, lv_out, off_in, off_px = [], [], 0, 0
    for (h, ww) in sizes:
        lv_in.append(ops.Level(off_in, h, ww, h * ww * Cf))
        lv_out.append(ops.Level(off_px * Co, h, ww, P * Co))
        off_in += B * h * ww * Cf
        off_px += h * ww
    src = ops.TView(packed, B, Cf, Cf, lv_in)
    out = torch.full((B, P, Co), float("nan"), device="cuda")
    ov = ops.TView(out.view(-1), B, Co, Co, lv_out)
    wd, bd, ad = dev(ohwi(w)), dev(b), dev(addm)
    U = make_u(L, ops, wd, Co, Cf, 9 * Ct, Ct, False)
    for TB, BN, ps4 in ((64, 64, 0), (32, 64, 0), (64, 32, 0), (32, 32, 0), (64, 64, 1), (32, 64, 1), (32, 32, 1)):
        out.fill_(float("nan"))
        desc = ops.fwd_desc(src, ov, Cf, Co, 3, 1, 1, 1, wC=Ct, relu=True, tile_hint=TB | (BN << 8) | (1 << 16) | (ps4 << 24))
        L.check(L.lib.zsg_conv_wino(C.byref(desc), packed.data_ptr(), U.data_ptr(), out.data_ptr(), bd.data_ptr(), ad.data_ptr(), None, None,
                                    L.stream_ptr()), "wino")
        assert_close(out, ref, 3e-4, 3e-4, f"multi-level wino {TB}x{BN}")


WG_CASES = [
    # B, Ci, Co, H, W, splits, accumulate
    (2, 64, 64, 19, 19, 0, 0),
    (2, 64, 128, 20, 17, 3, 1),
    (3, 128, 64, 7, 10, 1, 0),
    (2, 48, 256, 10, 10, 4, 0),
    (2, 256, 45, 10, 10, 2, 1),
    (2, 256, 256, 1, 1, 1, 0),
    (16, 64, 64, 38, 38, 24, 0),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[f"g{i}" for i in range(len(WG_CASES))])
def test_wino_wgrad(Z, case):
    """zsg_conv_wgrad_wino (F(3x3,2x2)) vs autograd's conv2d weight gradient: rel 5e-4 of the gradient's max (the direct
    kernel's bound)."""
    L, ops = Z
    from test_gpu_ops import WS
    B, Ci, Co, H, W, splits, acc = case
    g = torch.Generator().manual_seed(31 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).requires_grad_()
    gy = torch.randn(B, Co, H, W, generator=g)
    F.conv2d(x, w, None, 1, 1).backward(gy)
    cp, Cop = pad4(Ci), pad4(Co)
    xd, dyd = dev(nhwc(x)), dev(nhwc(gy, Cop))
    src, dyv = view_of(ops, xd, B, H, W, cp), view_of(ops, dyd, B, H, W, Cop)
    first = None
    for xmap in (0, 1):       # tile_hint bit 24: block order "whole K slices per XCD" — the same sums in the same order
        d = ops.fwd_desc(src, dyv, cp, Co, 3, 1, 1, 1, wC=cp, tile_hint=ops.tile_hint(64, 64, splits, xmap))
        dw = torch.full((Co, 3, 3, cp), float(acc), device="cuda")
        L.check(L.lib.zsg_conv_wgrad_wino(C.byref(d), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), acc, WS.data_ptr(), WS.numel() * 4, L.stream_ptr()), "wgrad_wino")
        assert_close(dw[..., :Ci].permute(0, 3, 1, 2) - acc, w.grad, 5e-4, 5e-4 * float(w.grad.abs().max()), "wino wgrad")
        if cp > Ci:
            assert float((dw[..., Ci:] - acc).abs().max()) == 0.0
        if first is None:
            first = dw.clone()
        else:
            assert torch.equal(first, dw), "block order must not change the result"


def test_wino_wgrad_multilevel_window(Z):
    """all pyramid levels reduced in one launch into a channel window of a wider weight (head conv0: features only)"""
    L, ops = Z
    from test_gpu_ops import WS
    g = torch.Generator().manual_seed(12)
    B, Cf, Cw, Co = 2, 64, 12, 64
    Ct = Cf + Cw
    sizes = [(7, 7), (4, 4), (3, 3), (2, 2), (1, 1)]
    xs = [torch.randn(B, Cf, h, ww, generator=g) for h, ww in sizes]
    P = sum(h * ww for h, ww in sizes)
    gy = torch.randn(B, P, Co, generator=g)
    wr = (torch.randn(Co, Cf, 3, 3, generator=g) / 24).requires_grad_()
    offs, o = [], 0
    for h, ww in sizes:
        offs.append(o)
        o += h * ww
    tot = sum((F.conv2d(x, wr, None, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Co) * gy[:, o:o + x.shape[2] * x.shape[3]]).sum() for x, o in zip(xs, offs))
    tot.backward()
    packed = torch.cat([nhwc(x).reshape(-1) for x in xs]).cuda()
    lv_in, lv_dy, off_in, off_px = [], [], 0, 0
    for (h, ww) in sizes:
        lv_in.append(ops.Level(off_in, h, ww, h * ww * Cf))
        lv_dy.append(ops.Level(off_px * Co, h, ww, P * Co))
        off_in += B * h * ww * Cf
        off_px += h * ww
    src = ops.TView(packed, B, Cf, Cf, lv_in)
    gyd = dev(gy)
    dyv = ops.TView(gyd.view(-1), B, Co, Co, lv_dy)
    dw = torch.zeros(Co, 3, 3, Ct, device="cuda")
    for splits in (1, 2, 0):
        dw.fill_(2.0)
        d = ops.fwd_desc(src, dyv, Cf, Co, 3, 1, 1, 1, wC=Ct, wc0=0, tile_hint=ops.tile_hint(64, 64, splits))
        L.check(L.lib.zsg_conv_wgrad_wino(C.byref(d), packed.data_ptr(), gyd.data_ptr(), dw.data_ptr(), 1, WS.data_ptr(), WS.numel() * 4, L.stream_ptr()), "wgrad_wino")
        assert_close(dw[..., :Cf].permute(0, 3, 1, 2) - 2.0, wr.grad, 5e-4, 5e-4 * float(wr.grad.abs().max()), f"multi-level wino wgrad splits={splits}")
        assert float((dw[..., Cf:] - 2.0).abs().max()) == 0.0, "channels outside the window must stay untouched"


@pytest.mark.parametrize("case", [(2, 64, 128, 20, 17, 3, 1, 3), (3, 128, 64, 7, 10, 1, 0, 2), (16, 64, 64, 38, 38, 12, 1, 2), (2, 256, 45, 10, 10, 2, 0, 8)],
                         ids=["b0", "b1", "b2", "b3"])
def test_wino_wgrad_batched(Z, case):
    """zsg_conv_wgrad_wino_batched (round 6): J convolutions of one geometry in one launch = J x zsg_conv_wgrad_wino on the same
    operands, BIT FOR BIT at the same split-K factor (a job's blocks walk the same stages in the same order, its slabs are reduced in the
    same order), and autograd's weight gradient within the single launch's bound; accumulate adds to what the gradient image holds.
    Refusals: more than 8 jobs, a null operand -> -1, nothing falls back.  Reference: autograd through nn.Conv2d of
    fpn_resnet.py:86-100's identical bottlenecks."""
    L, ops = Z
    from test_gpu_ops import WS
    B, Ci, Co, H, W, splits, acc, J = case
    g = torch.Generator().manual_seed(77 + Ci + Co + H + J)
    cp, Cop = pad4(Ci), pad4(Co)
    xs, gys, refs = [], [], []
    for j in range(J):
        x = torch.randn(B, Ci, H, W, generator=g)
        w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).requires_grad_()
        gy = torch.randn(B, Co, H, W, generator=g)
        F.conv2d(x, w, None, 1, 1).backward(gy)
        xs.append(dev(nhwc(x)))
        gys.append(dev(nhwc(gy, Cop)))
        refs.append(w.grad)
    src, dyv = view_of(ops, xs[0], B, H, W, cp), view_of(ops, gys[0], B, H, W, Cop)
    d = ops.fwd_desc(src, dyv, cp, Co, 3, 1, 1, 1, wC=cp, tile_hint=ops.tile_hint(64, 64, splits))
    st = L.stream_ptr()
    one = [torch.full((Co, 3, 3, cp), float(acc), device="cuda") for _ in range(J)]
    for j in range(J):
        L.check(L.lib.zsg_conv_wgrad_wino(C.byref(d), xs[j].data_ptr(), gys[j].data_ptr(), one[j].data_ptr(), acc, WS.data_ptr(), WS.numel() * 4, st), "wgrad_wino")
    bat = [torch.full((Co, 3, 3, cp), float(acc), device="cuda") for _ in range(J)]
    VP = C.c_void_p * J
    a, b, c = VP(*[t.data_ptr() for t in xs]), VP(*[t.data_ptr() for t in gys]), VP(*[t.data_ptr() for t in bat])
    L.check(L.lib.zsg_conv_wgrad_wino_batched(C.byref(d), J, a, b, c, acc, WS.data_ptr(), WS.numel() * 4, st), "wgrad_wino_batched")
    torch.cuda.synchronize()
    for j in range(J):
        assert torch.equal(bat[j], one[j]), f"job {j}: the batched launch must reproduce the single launch bit for bit"
        assert_close(bat[j][..., :Ci].permute(0, 3, 1, 2) - acc, refs[j], 5e-4, 5e-4 * float(refs[j].abs().max()), f"batched wino wgrad job {j}")
    # refusals
    VP9 = C.c_void_p * 9
    nine = VP9(*([xs[0].data_ptr()] * 9))
    assert L.lib.zsg_conv_wgrad_wino_batched(C.byref(d), 9, nine, nine, nine, 0, WS.data_ptr(), WS.numel() * 4, st) != 0
    hole = VP(*([xs[0].data_ptr()] * (J - 1) + [None]))
    assert L.lib.zsg_conv_wgrad_wino_batched(C.byref(d), J, hole, b, c, 0, WS.data_ptr(), WS.numel() * 4, st) != 0
    assert L.lib.zsg_conv_wgrad_wino_batched(C.byref(d), J, a, b, c, 0, WS.data_ptr(), 1024, st) != 0 or splits <= 1, "a workspace too small for J x splits slabs is refused"

