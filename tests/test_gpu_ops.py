"""GPU parity of every HIP op (called through the C ABI) against the CPU oracle / torch-CPU fp32 references.
Tolerances are stated per test: index / mask outputs exact, fp32 sums rel 1e-4 (summation order), losses rel 1e-5."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import zsg_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def Z():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import zsgnet_pytorch_amd._lib as L
    import zsgnet_pytorch_amd.ops as ops
    stream_scratch(L)
    return L, ops


_SCRATCH = {}


def stream_scratch(L, stream=None, mb=40):
    """caller-owned scratch of the stream-K launches (zsg_set_stream_workspace), registered once per stream"""
    key = L.stream_ptr() if stream is None else stream
    if key not in _SCRATCH:
        _SCRATCH[key] = torch.zeros(mb << 18, device="cuda")
        L.check(L.lib.zsg_set_stream_workspace(C.c_void_p(key), _SCRATCH[key].data_ptr(), _SCRATCH[key].numel() * 4), "set_stream_workspace")
    return _SCRATCH[key]


def dev(t):
    return t.cuda().contiguous()


class _LazyWS:
    """split-K workspace of zsg_conv_wgrad (allocated on first use)"""
    t = None

    def _get(self):
        if _LazyWS.t is None:
            _LazyWS.t = torch.empty(64 << 20, device="cuda")
        return _LazyWS.t

    def data_ptr(self):
        return self._get().data_ptr()

    def numel(self):
        return self._get().numel()


WS = _LazyWS()


def pad4(n):
    return (n + 3) // 4 * 4


def nhwc(x, cpad=None):
    B, Cc, H, W = x.shape
    cpad = cpad or pad4(Cc)
    o = torch.zeros(B, H, W, cpad)
    o[..., :Cc] = x.permute(0, 2, 3, 1)
    return o


def ohwi(w, cpad=None):
    Co, Ci, k, _ = w.shape
    cpad = cpad or pad4(Ci)
    o = torch.zeros(Co, k, k, cpad)
    o[..., :Ci] = w.permute(0, 2, 3, 1)
    return o


def view_of(ops, t, B, H, W, Cc, ld=None):
    return ops.TView(t.view(-1), B, Cc, ld or Cc, [ops.Level(0, H, W, H * W * (ld or Cc))])


def assert_close(got, ref, rtol, atol, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax((err - tol).flatten()))
        idx = np.unravel_index(i, tuple(err.shape))
        raise AssertionError(f"{what}: {int(bad.sum())}/{err.numel()} mismatches; worst at {idx}: got {got.flatten()[i]:.6g} "
                             f"ref {ref.flatten()[i]:.6g} (max abs err {err.max():.3g}, ref scale {ref.abs().max():.3g})")


CONV_CASES = [
    # B, Ci, Co, H, W, k, s, p, d, bias, relu, merge_x, tile
    (2, 64, 64, 19, 19, 1, 1, 0, 1, False, False, False, 0),
    (2, 64, 128, 20, 17, 3, 1, 1, 1, True, True, False, 0),
    (2, 128, 128, 21, 21, 3, 2, 1, 1, False, False, False, 0),
    (3, 256, 64, 9, 11, 1, 2, 0, 1, False, False, False, 0),
    (2, 3, 64, 45, 37, 7, 2, 3, 1, False, False, True, 0),
    (2, 3, 64, 31, 30, 3, 1, 1, 1, True, True, True, 0),
    (1, 514, 256, 10, 10, 3, 1, 1, 1, True, True, False, 0),
    (2, 256, 45, 10, 10, 3, 1, 1, 1, True, False, False, 0),
    (1, 64, 96, 12, 12, 3, 1, 6, 6, True, False, False, 0),
    (2, 128, 256, 5, 5, 3, 1, 0, 1, True, True, False, 0),
    (2, 64, 256, 16, 16, 1, 1, 0, 1, False, False, False, 128 | (128 << 8)),
    (2, 64, 256, 16, 16, 3, 1, 1, 1, False, False, False, 128 | (64 << 8)),
    (2, 64, 256, 16, 16, 3, 1, 1, 1, False, False, False, 64 | (64 << 8)),
    (2, 64, 256, 16, 16, 3, 1, 1, 1, True, True, False, 128 | (128 << 8) | (1 << 24)),
    (2, 96, 64, 16, 16, 3, 1, 1, 1, True, False, False, 128 | (64 << 8) | (1 << 24)),
    (16, 300, 512, 1, 20, 1, 1, 0, 1, True, False, False, 0),
    # 64-deep K tiles (tile_hint bit 27): every tile variant, taps + padding + stride, K = 64 (one step) .. 576
    (2, 64, 256, 16, 16, 1, 1, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 27)),
    (2, 128, 256, 16, 16, 1, 1, 0, 1, True, True, False, 64 | (64 << 8) | (1 << 24) | (1 << 27)),
    (2, 192, 128, 17, 15, 3, 1, 1, 1, False, False, False, 128 | (64 << 8) | (1 << 27)),
    (2, 64, 192, 21, 21, 3, 2, 1, 1, True, False, False, 128 | (64 << 8) | (1 << 24) | (1 << 27)),
    (2, 256, 256, 9, 9, 1, 1, 0, 1, False, False, False, 128 | (128 << 8) | (1 << 24) | (1 << 27)),
    (3, 128, 64, 11, 13, 1, 2, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 27)),
    # stream-K (tile_hint bits 28-29 = workgroups per CU): one K tile, odd / even tile counts, taps + padding + stride, epilogue terms
    (2, 32, 64, 16, 16, 1, 1, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    (2, 64, 256, 16, 16, 1, 1, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    (2, 96, 128, 17, 15, 3, 1, 1, 1, True, True, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    (2, 64, 192, 21, 21, 3, 2, 1, 1, True, False, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    (1, 1024, 256, 19, 19, 1, 1, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    (3, 160, 64, 11, 13, 1, 2, 0, 1, False, False, False, 64 | (64 << 8) | (1 << 24) | (1 << 28)),
    # stream-K, two workgroups per CU: K of one / two / three / many tiles, every tile shape it exists for
    (2, 32, 64, 16, 16, 1, 1, 0, 1, False, False, False, 64 | (64 << 8) | (2 << 28)),
    (2, 64, 256, 16, 16, 1, 1, 0, 1, True, True, False, 64 | (64 << 8) | (1 << 24) | (1 << 27) | (2 << 28)),
    (2, 96, 128, 17, 15, 3, 1, 1, 1, True, True, False, 128 | (64 << 8) | (1 << 24) | (2 << 28)),
    (2, 64, 192, 21, 21, 3, 2, 1, 1, True, False, False, 128 | (128 << 8) | (1 << 24) | (2 << 28)),
    (1, 1024, 256, 19, 19, 1, 1, 0, 1, False, False, False, 128 | (128 << 8) | (2 << 28)),
    (3, 160, 64, 11, 13, 1, 2, 0, 1, False, False, False, 64 | (64 << 8) | (3 << 28)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_conv_fwd_dgrad_wgrad(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, k, s, p, d, bias, relu, mx, tile = case
    g = torch.Generator().manual_seed(100 + Ci + Co + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    y_ref = F.conv2d(xr, wr, br, s, p, d)
    if relu:
        y_ref = F.relu(y_ref)
    Ho, Wo = y_ref.shape[2:]
    gy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(gy)
    cp = pad4(Ci)
    st = L.stream_ptr()

    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    out = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    src = view_of(ops, xd, B, H, W, cp)
    ov = view_of(ops, out, B, Ho, Wo, Co)
    desc = ops.fwd_desc(src, ov, cp, Co, k, s, p, d, wC=cp, relu=relu, merge_x=mx, tile_hint=tile)
    bd = dev(b) if bias else None
    L.check(L.lib.zsg_conv_igemm(C.byref(desc), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr() if bias else None, None,
                                 None, None, st), "igemm")
    assert_close(out.permute(0, 3, 1, 2), y_ref, 2e-4, 2e-4, "conv fwd")
    if not bias and not relu:
        # fused BatchNorm statistics: per-tile (sum, sum^2) partials from the epilogue -> mean / invstd
        for bm, bn_, w8, k64 in ((64, 64, 0, 0), (128, 64, 0, 0), (128, 128, 1, 0), (64, 64, 1, 1), (128, 64, 1, 1), (128, 128, 1, 1), (64, 64, 1, 2), (64, 64, 0, 4), (128, 64, 1, 4)):
            if (bn_ == 128 and (Co <= 64 or mx)) or (k64 == 1 and (mx or cp % 64)) or (k64 in (2, 4) and mx):
                continue
            d4 = ops.fwd_desc(src, ov, cp, Co, k, s, p, d, wC=cp, merge_x=mx, tile_hint=ops.tile_hint(bm, bn_, 1, w8) | (k64 << 27))
            chunks = (B * Ho * Wo + bm - 1) // bm
            part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
            L.check(L.lib.zsg_conv_igemm(C.byref(d4), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "igemm+stats")
            yf = y_ref.detach().permute(0, 2, 3, 1).reshape(-1, Co).double()
            assert_close(part[:, 0].double().sum(0), yf.sum(0), 1e-4, 1e-4 * float(yf.abs().sum(0).max()), "bn partial sums")
            assert_close(part[:, 1].double().sum(0), (yf * yf).sum(0), 1e-4, 1e-6, "bn partial sums of squares")
            mean, invstd = torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
            L.check(L.lib.zsg_bn_stats_from_partials(part.data_ptr(), chunks, B * Ho * Wo, Co, mean.data_ptr(), invstd.data_ptr(), None, None,
                                                     0.1, 1e-5, st), "bn_from_partials")
            assert_close(mean, yf.mean(0), 1e-4, 1e-5, "fused bn mean")
            assert_close(invstd, 1 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5), 2e-4, 0, "fused bn invstd")
            if chunks <= L.lib.zsg_bn_inline_max_chunks():
                # finalize-free apply: every block reduces the few partial rows itself; must equal nn.BatchNorm2d(train) + ReLU
                gam, bet = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
                rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
                m2, i2 = torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
                yb = torch.empty_like(out)
                mask = torch.empty((out.numel() // 4 + 3) // 4 * 4, dtype=torch.uint8, device="cuda")
                gd, bd_ = dev(gam), dev(bet)
                L.check(L.lib.zsg_bn_apply_from_partials(out.data_ptr(), B * Ho * Wo, Co, part.data_ptr(), chunks, gd.data_ptr(), bd_.data_ptr(), None, 1,
                                                         yb.data_ptr(), mask.data_ptr(), m2.data_ptr(), i2.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, st),
                        "bn_apply_from_partials")
                ref_bn = F.relu(F.batch_norm(y_ref.detach(), None, None, gam, bet, True, 0.1, 1e-5))
                assert_close(yb.permute(0, 3, 1, 2), ref_bn, 5e-4, 5e-4, "bn apply from partials")
                assert_close(m2, mean, 1e-6, 1e-7, "inline mean == finalize mean")
                assert_close(rm, 0.1 * yf.mean(0), 1e-4, 1e-6, "running mean")
                assert_close(rv, 0.9 + 0.1 * yf.var(0, unbiased=True), 2e-4, 1e-6, "running var")

    # backward: dy is the gradient w.r.t. the pre-ReLU output
    gpre = gy * (y_ref > 0) if relu else gy
    Cop = pad4(Co)
    dyd = dev(nhwc(gpre, Cop))
    dyv = view_of(ops, dyd, B, Ho, Wo, Cop)
    # wgrad
    dw = torch.zeros(Co, k, k, cp, device="cuda")
    wdesc = ops.fwd_desc(src, dyv, cp, Co, k, s, p, d, wC=cp)
    L.check(L.lib.zsg_conv_wgrad(C.byref(wdesc), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgrad")
    assert_close(dw[..., :Ci].permute(0, 3, 1, 2), wr.grad, 5e-4, 5e-4 * float(wr.grad.abs().max()), "conv wgrad")
    for hint, acc in ((ops.tile_hint(64, 64, 1), 1), (ops.tile_hint(128, 64, 3), 1), (ops.tile_hint(64, 128, 7), 0),
                      (ops.tile_hint(128, 128, 2, 1, 0), 0), (ops.tile_hint(128, 128, 3, 1, 1), 1), (ops.tile_hint(128, 128, 1, 0, 1), 0)):
        dw2 = torch.full((Co, k, k, cp), 1.0, device="cuda")
        wd2 = ops.fwd_desc(src, dyv, cp, Co, k, s, p, d, wC=cp, tile_hint=hint)
        L.check(L.lib.zsg_conv_wgrad(C.byref(wd2), xd.data_ptr(), dyd.data_ptr(), dw2.data_ptr(), acc, WS.data_ptr(), WS.numel() * 4, st), "wgrad hint")
        assert_close(dw2[..., :Ci].permute(0, 3, 1, 2) - acc, wr.grad, 5e-4, 5e-4 * float(wr.grad.abs().max()), f"conv wgrad hint {hint:x} acc={acc}")
    if cp > Ci:
        assert float(dw[..., Ci:].abs().max()) == 0.0
    if bias:
        db = torch.zeros(Co, device="cuda")
        L.check(L.lib.zsg_colsum(dyd.data_ptr(), 1, 0, B * Ho * Wo, Cop, 0, Co, db.data_ptr(), 0, st), "colsum")
        assert_close(db, br.grad, 5e-4, 5e-4 * float(br.grad.abs().max()), "bias grad")
    # dgrad (skipped for RGB inputs: the image needs no gradient)
    if not mx:
        wt = torch.full((cp, k, k, Cop), float("nan"), device="cuda")
        L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, k * k, cp, Cop, st), "transpose_w")
        assert torch.equal(wt[..., :Co].cpu(), ohwi(w).permute(3, 1, 2, 0).contiguous())
        dx = torch.full((B, H, W, cp), float("nan"), device="cuda")
        dxv = view_of(ops, dx, B, H, W, cp)
        ddesc = ops.dgrad_desc(dyv, dxv, Cop, cp, k, s, p, d)
        if ddesc.zero_fill:              # parity classes without any tap are not launched: the caller clears dx
            dx.zero_()
        L.check(L.lib.zsg_conv_igemm(C.byref(ddesc), dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st), "dgrad")
        assert_close(dx[..., :Ci].permute(0, 3, 1, 2), xr.grad, 5e-4, 5e-4 * float(xr.grad.abs().max()), "conv dgrad")
        # accumulate + relu-mask epilogue: out = (prev + acc) * (mask > 0)
        prev = torch.randn(B, H, W, cp, generator=g)
        mask = torch.randn(B, H, W, cp, generator=g)
        dx2, maskd = dev(prev), dev(mask)
        L.check(L.lib.zsg_conv_igemm(C.byref(ddesc), dyd.data_ptr(), wt.data_ptr(), dx2.data_ptr(), None, dx2.data_ptr(), maskd.data_ptr(), None, st), "dgrad+")
        ref2 = (prev[..., :Ci] + xr.grad.permute(0, 2, 3, 1)) * (mask[..., :Ci] > 0)
        if not ddesc.zero_fill:          # (dropped parity classes are neither accumulated nor masked: never needed)
            assert_close(dx2[..., :Ci], ref2, 5e-4, 5e-4 * float(ref2.abs().max()), "dgrad accumulate+mask")
        # split-K variants (fp32 atomics) of the same data gradient
        for sp in (2, 5):
            dx3 = torch.full((B, H, W, cp), float("nan"), device="cuda")
            if ddesc.nseg == 1 and s == 1:
                d3 = ops.dgrad_desc(dyv, view_of(ops, dx3, B, H, W, cp), Cop, cp, k, s, p, d, tile_hint=ops.tile_hint(64, 64, sp))
                L.check(L.lib.zsg_conv_igemm(C.byref(d3), dyd.data_ptr(), wt.data_ptr(), dx3.data_ptr(), None, None, None, None, st), "dgrad split-K")
                assert_close(dx3[..., :Ci].permute(0, 3, 1, 2), xr.grad, 5e-4, 5e-4 * float(xr.grad.abs().max()), f"conv dgrad split-K {sp}")
    if not relu:
        for sp in (3, 8):
            out3 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
            d3 = ops.fwd_desc(src, view_of(ops, out3, B, Ho, Wo, Co), cp, Co, k, s, p, d, wC=cp, merge_x=mx, tile_hint=ops.tile_hint(64, 64, sp))
            L.check(L.lib.zsg_conv_igemm(C.byref(d3), xd.data_ptr(), wd.data_ptr(), out3.data_ptr(), bd.data_ptr() if bias else None, None, None, None, st), "fwd split-K")
            assert_close(out3.permute(0, 3, 1, 2), y_ref, 2e-4, 2e-4, f"conv fwd split-K {sp}")
    torch.cuda.synchronize()


PW_CASES = [
    # B, Ci, Co, H, W   (1x1 / stride 1; every unit width the geometry admits is run)
    (2, 64, 256, 19, 19),        # 23 row tiles, the last with 18 live rows; fewer units than waves
    (3, 256, 64, 10, 13),        # four K chunks per unit
    (2, 64, 64, 21, 21),
    (1, 128, 128, 37, 41),       # two K chunks, N = one 128-wide unit
    (5, 64, 192, 30, 30),        # N = 192 = 3 or 6 units: not a geometry of the kernel — the hint must be refused
    (2, 128, 512, 19, 19),       # filter cut into 4 panels of 128 rows (one per workgroup column block); dgrad: 512 -> 128 in panels of 32
    (1, 256, 128, 20, 21),       # 2 panels of 64 rows
    (16, 128, 512, 38, 38),      # layer2 conv3 at the bench shape: 64 row groups x 4 column blocks
    (16, 64, 256, 75, 75),       # layer1 conv3 at the bench shape: 2-3 units per wave, next-unit prefetch across units
    (16, 256, 64, 75, 75),       # layer1 conv1
]


@pytest.mark.parametrize("case", PW_CASES, ids=[f"p{i}" for i in range(len(PW_CASES))])
def test_conv_pw_streaming(Z, case):
    """The filter-resident streaming kernel behind zsg_conv_igemm's tile_hint BM = 32 (csrc/pw.hip) against torch-CPU fp32
    F.conv2d: forward with bias + ReLU, forward with the fused BatchNorm statistics (one partial row per workgroup,
    zsg_conv_igemm_partial_rows), data gradient plain and with accumulate + mask; the plain launches must also equal the 64x64
    tile's results to fp32 summation-order accuracy.  Tolerances as test_conv_fwd_dgrad_wgrad."""
    L, ops = Z
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(7 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    b = torch.randn(Co, generator=g)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    y_lin = F.conv2d(xr, wr)
    gy = torch.randn(y_lin.shape, generator=g)
    y_lin.backward(gy)
    y_ref = F.relu(y_lin.detach() + b.view(1, -1, 1, 1))
    st = L.stream_ptr()
    xd, wd, bd = dev(nhwc(x)), dev(ohwi(w)), dev(b)
    src = view_of(ops, xd, B, H, W, Ci)
    rows = B * H * W
    probe = ops.fwd_desc(src, view_of(ops, torch.empty(1), B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci)
    hints = ops.pw_cands(probe)
    if Co == 192:
        assert hints == [], "192 channels = 3 / 6 units: not a geometry of the streaming kernel"
        probe.tile_hint = ops.tile_hint(32, 64, 1)
        assert L.lib.zsg_conv_igemm_partial_rows(C.byref(probe)) == -1
        out = torch.zeros(B, H, W, Co, device="cuda")
        rc = L.lib.zsg_conv_igemm(C.byref(ops.fwd_desc(src, view_of(ops, out, B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=probe.tile_hint)),
                                  xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, None, st)
        assert rc != 0, "an inapplicable streaming hint must fail loudly, not fall back"
        return
    assert hints, "the streaming kernel must cover this geometry"
    yf = y_lin.detach().permute(0, 2, 3, 1).reshape(-1, Co).double()
    for hint in hints:
        out = torch.full((B, H, W, Co), float("nan"), device="cuda")
        ov = view_of(ops, out, B, H, W, Co)
        d1 = ops.fwd_desc(src, ov, Ci, Co, 1, 1, 0, 1, wC=Ci, relu=True, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(d1), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr(), None, None, None, st), "pw fwd")
        assert_close(out.permute(0, 3, 1, 2), y_ref, 2e-4, 2e-4, f"pw fwd bias+relu hint {hint:x}")
        # fused BatchNorm statistics
        d2 = ops.fwd_desc(src, ov, Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
        chunks = ops.igemm_partial_rows(d2)
        uw = (hint >> 8) & 0xff
        ncb = next(c for c in (1, 2, 4, 8) if Co % c == 0 and (Co // c) % uw == 0 and (Co // c) // uw in (1, 2, 4, 8)
                   and ((Co // c) * (Ci + 4) + 8 * 32 * 68) * 4 <= 160 * 1024)          # filter panels (column blocks of the grid)
        assert chunks == min(256 // ncb, ((rows + 31) // 32 * (Co // ncb // uw) + 7) // 8)
        part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
        out.fill_(float("nan"))
        L.check(L.lib.zsg_conv_igemm(C.byref(d2), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "pw fwd+stats")
        assert_close(out.permute(0, 3, 1, 2), y_lin.detach(), 2e-4, 2e-4, f"pw fwd hint {hint:x}")
        assert not torch.isnan(part).any()
        assert_close(part[:, 0].double().sum(0), yf.sum(0), 1e-4, 1e-4 * float(yf.abs().sum(0).max()), "pw bn partial sums")
        assert_close(part[:, 1].double().sum(0), (yf * yf).sum(0), 1e-4, 1e-6, "pw bn partial sums of squares")
        mean, invstd = torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
        L.check(L.lib.zsg_bn_stats_from_partials(part.data_ptr(), chunks, rows, Co, mean.data_ptr(), invstd.data_ptr(), None, None, 0.1, 1e-5, st), "finalize")
        assert_close(mean, yf.mean(0), 1e-4, 1e-5, "pw fused bn mean")
        assert_close(invstd, 1 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5), 2e-4, 0, "pw fused bn invstd")
        # the same launch twice: bit-identical (no atomics, fixed reduction order)
        part2, out2 = torch.full_like(part, float("nan")), torch.full_like(out, float("nan"))
        L.check(L.lib.zsg_conv_igemm(C.byref(d2), xd.data_ptr(), wd.data_ptr(), out2.data_ptr(), None, None, None, part2.data_ptr(), st), "pw again")
        assert torch.equal(out, out2) and torch.equal(part, part2)
    # data gradient: rows = dx pixels, reduction = dy channels
    dyd = dev(nhwc(gy))
    dyv = view_of(ops, dyd, B, H, W, Co)
    wt = torch.empty((Ci, 1, 1, Co), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, 1, Ci, Co, st), "transpose_w")
    probe = ops.dgrad_desc(dyv, view_of(ops, torch.empty(1), B, H, W, Ci), Co, Ci, 1, 1, 0, 1)
    dhints = ops.pw_cands(probe)
    assert dhints
    prev = torch.randn(B, H, W, Ci, generator=g)
    mask = torch.randn(B, H, W, Ci, generator=g)
    for hint in dhints:
        dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        dd = ops.dgrad_desc(dyv, view_of(ops, dx, B, H, W, Ci), Co, Ci, 1, 1, 0, 1, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(dd), dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st), "pw dgrad")
        assert_close(dx.permute(0, 3, 1, 2), xr.grad, 5e-4, 5e-4 * float(xr.grad.abs().max()), f"pw dgrad hint {hint:x}")
        dx2, maskd = dev(prev), dev(mask)
        d2 = ops.dgrad_desc(dyv, view_of(ops, dx2, B, H, W, Ci), Co, Ci, 1, 1, 0, 1, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(d2), dyd.data_ptr(), wt.data_ptr(), dx2.data_ptr(), None, dx2.data_ptr(), maskd.data_ptr(), None, st), "pw dgrad+")
        ref2 = (prev + xr.grad.permute(0, 2, 3, 1)) * (mask > 0)
        assert_close(dx2, ref2, 5e-4, 5e-4 * float(ref2.abs().max()), f"pw dgrad accumulate+mask hint {hint:x}")
    torch.cuda.synchronize()


MX_CASES = [
    # B, H, W, k, stride, pad, bias+relu
    (2, 45, 37, 7, 2, 3),          # the ResNet stem's window on an odd image: units that straddle image rows, a ragged last unit
    (3, 64, 64, 7, 2, 3),          # 3 x 32 x 32 = 96 units
    (1, 7, 9, 7, 2, 3),            # a window larger than the image on every side
    (2, 31, 30, 3, 1, 1),          # SSD-VGG conv1_1's window
    (16, 300, 300, 7, 2, 3),       # the bench shape: 11 250 units, 5-6 per wave, prefetch across units
]


@pytest.mark.parametrize("case", MX_CASES, ids=[f"m{i}" for i in range(len(MX_CASES))])
def test_conv_mx_streaming(Z, case):
    """The filter-resident streaming kernel of the network's first convolution (csrc/mx.hip: tile_hint BM = 32 on a merge_x
    descriptor) against torch-CPU fp32 F.conv2d: plain + fused BatchNorm statistics (one partial row per workgroup), bias + ReLU,
    two launches bit-identical, and against the implicit-GEMM merge_x tile.  Tolerances as test_conv_fwd_dgrad_wgrad."""
    L, ops = Z
    B, H, W, k, s, p = case
    Ci, Co = 3, 64
    g = torch.Generator().manual_seed(900 + H + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    y_lin = F.conv2d(x, w, None, s, p)
    Ho, Wo = y_lin.shape[2:]
    st = L.stream_ptr()
    xd, wd, bd = dev(nhwc(x)), dev(ohwi(w)), dev(b)
    src = view_of(ops, xd, B, H, W, 4)
    out = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    ov = view_of(ops, out, B, Ho, Wo, Co)
    hint = ops.tile_hint(32, 64, 1)
    probe = ops.fwd_desc(src, ov, 4, Co, k, s, p, 1, wC=4, merge_x=True)
    assert ops.pw_cands(probe) == [hint], "the streaming first-layer kernel must cover this geometry"
    d2 = ops.fwd_desc(src, ov, 4, Co, k, s, p, 1, wC=4, merge_x=True, tile_hint=hint)
    rows = B * Ho * Wo
    chunks = ops.igemm_partial_rows(d2)
    assert chunks == min(256, ((rows + 31) // 32 + 7) // 8)
    part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
    L.check(L.lib.zsg_conv_igemm(C.byref(d2), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "mx fwd+stats")
    assert_close(out.permute(0, 3, 1, 2), y_lin, 2e-4, 2e-4, "mx fwd")
    yf = y_lin.permute(0, 2, 3, 1).reshape(-1, Co).double()
    assert not torch.isnan(part).any()
    assert_close(part[:, 0].double().sum(0), yf.sum(0), 1e-4, 1e-4 * float(yf.abs().sum(0).max()), "mx bn partial sums")
    assert_close(part[:, 1].double().sum(0), (yf * yf).sum(0), 1e-4, 1e-6, "mx bn partial sums of squares")
    mean, invstd = torch.empty(Co, device="cuda"), torch.empty(Co, device="cuda")
    L.check(L.lib.zsg_bn_stats_from_partials(part.data_ptr(), chunks, rows, Co, mean.data_ptr(), invstd.data_ptr(), None, None, 0.1, 1e-5, st), "finalize")
    assert_close(mean, yf.mean(0), 1e-4, 1e-5, "mx fused bn mean")
    assert_close(invstd, 1 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5), 2e-4, 0, "mx fused bn invstd")
    part2, out2 = torch.full_like(part, float("nan")), torch.full_like(out, float("nan"))
    L.check(L.lib.zsg_conv_igemm(C.byref(d2), xd.data_ptr(), wd.data_ptr(), out2.data_ptr(), None, None, None, part2.data_ptr(), st), "mx again")
    assert torch.equal(out, out2) and torch.equal(part, part2)
    # the implicit-GEMM merge_x tile on the same operands: fp32 summation-order accuracy
    out3 = torch.full_like(out, float("nan"))
    d3 = ops.fwd_desc(src, view_of(ops, out3, B, Ho, Wo, Co), 4, Co, k, s, p, 1, wC=4, merge_x=True, tile_hint=ops.tile_hint(64, 64, 1))
    L.check(L.lib.zsg_conv_igemm(C.byref(d3), xd.data_ptr(), wd.data_ptr(), out3.data_ptr(), None, None, None, None, st), "igemm tile")
    assert_close(out, out3, 1e-5, 1e-5 * float(out3.abs().max()), "mx vs the 64x64 tile")
    # bias + ReLU
    out.fill_(float("nan"))
    d1 = ops.fwd_desc(src, ov, 4, Co, k, s, p, 1, wC=4, relu=True, merge_x=True, tile_hint=hint)
    L.check(L.lib.zsg_conv_igemm(C.byref(d1), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr(), None, None, None, st), "mx bias+relu")
    assert_close(out.permute(0, 3, 1, 2), F.relu(y_lin + b.view(1, -1, 1, 1)), 2e-4, 2e-4, "mx fwd bias+relu")
    # an add operand is not this kernel's: loud failure, no fallback
    rc = L.lib.zsg_conv_igemm(C.byref(d1), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, out.data_ptr(), None, None, st)
    assert rc != 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(2, 256, 64, 19, 19), (3, 512, 128, 10, 13), (16, 256, 64, 75, 75), (2, 1024, 256, 19, 19)], ids=["l1", "l2", "l1bench", "l3"])
def test_conv_igemm_bnpre(Z, case):
    """zsg_conv_igemm_bnpre — the next bottleneck's conv1 applying the previous block's bn3 + residual + ReLU in its operand loader and
    materialising the activation itself — against the unfused pair zsg_bn_apply -> zsg_conv_igemm on the same operands: the
    materialised activation and its packed ReLU bits must be zsg_bn_apply's (same arithmetic: equal to 1 ulp of fp32 contraction),
    the convolution output and its fused statistics equal to summation-order accuracy, for every tile the entry accepts, with and
    without the in-kernel finalize; and against torch-CPU fp32 for the whole chain."""
    L, ops = Z
    B, Ci, Co, H, W = case
    g = torch.Generator().manual_seed(31 + Ci + H)
    rows = B * H * W
    x = torch.randn(rows, Ci, generator=g) * 2 + 0.3
    res = torch.randn(rows, Ci, generator=g)
    gam, bet = torch.rand(Ci, generator=g) + 0.5, torch.randn(Ci, generator=g) * 0.2
    mean = x.mean(0)
    invstd = 1 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)
    w = torch.randn(Co, Ci, generator=g) / Ci ** 0.5
    y_t = torch.relu((x - mean) * (invstd * gam) + bet + res)
    o_t = y_t @ w.t()
    st = L.stream_ptr()
    xd, rd, gd, bd, md, isd, wd = dev(x), dev(res), dev(gam), dev(bet), dev(mean), dev(invstd), dev(w.view(Co, 1, 1, Ci))
    y0 = torch.full((rows, Ci), float("nan"), device="cuda")
    m0 = torch.zeros(rows * Ci // 4, dtype=torch.uint8, device="cuda")
    L.check(L.lib.zsg_bn_apply(xd.data_ptr(), rows, Ci, md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rd.data_ptr(), 1, y0.data_ptr(),
                               m0.data_ptr(), st), "bn_apply")
    assert_close(y0, y_t, 1e-5, 1e-5, "bn_apply vs torch")
    src0 = view_of(ops, y0, B, H, W, Ci)
    o0 = torch.full((rows, Co), float("nan"), device="cuda")
    d0 = ops.fwd_desc(src0, view_of(ops, o0, B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, 1))
    L.check(L.lib.zsg_conv_igemm(C.byref(d0), y0.data_ptr(), wd.data_ptr(), o0.data_ptr(), None, None, None, None, st), "plain conv")
    assert_close(o0, o_t, 2e-4, 2e-4 * float(o_t.abs().max()), "unfused chain vs torch")
    of = o0.double()
    for bm, bn_, w8 in ((64, 64, 0), (64, 64, 1), (128, 64, 0), (128, 64, 1), (128, 128, 0), (128, 128, 1)):
        if bn_ == 128 and Co < 128:
            continue
        hint = ops.tile_hint(bm, bn_, 1, w8)
        for tail in (False, True):
            y1 = torch.full((rows, Ci), float("nan"), device="cuda")
            m1 = torch.full((rows * Ci // 4,), 0xAA, dtype=torch.uint8, device="cuda")
            o1 = torch.full((rows, Co), float("nan"), device="cuda")
            d1 = ops.fwd_desc(view_of(ops, xd, B, H, W, Ci), view_of(ops, o1, B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
            chunks = ops.igemm_partial_rows(d1)
            part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
            tk = torch.zeros(64, dtype=torch.int32, device="cuda")
            mo, io = torch.full((Co,), float("nan"), device="cuda"), torch.full((Co,), float("nan"), device="cuda")
            ntk = int(L.lib.zsg_conv_bn_tail_tickets(C.byref(d1), 0))
            use_tail = tail and ntk > 0
            if tail and not use_tail:
                continue
            if (bm, bn_, w8) == (128, 128, 0):      # no BatchNorm-applying loader for the 4-wave 128x128 tile (it spilled: round 6): refused, loudly
                rc = L.lib.zsg_conv_igemm_bnpre(C.byref(d1), xd.data_ptr(), wd.data_ptr(), o1.data_ptr(), part.data_ptr(), None, mo.data_ptr(), io.data_ptr(),
                                                None, None, 0.1, 1e-5, md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y1.data_ptr(),
                                                m1.data_ptr(), st)
                assert rc == -1 and b"8-wave" in L.lib.zsg_last_error()
                continue
            L.check(L.lib.zsg_conv_igemm_bnpre(C.byref(d1), xd.data_ptr(), wd.data_ptr(), o1.data_ptr(), part.data_ptr(),
                                               tk.data_ptr() if use_tail else None, mo.data_ptr(), io.data_ptr(), None, None, 0.1, 1e-5,
                                               md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y1.data_ptr(), m1.data_ptr(), st),
                    f"bnpre {bm}x{bn_} w8={w8} tail={tail}")
            what = f"{bm}x{bn_} w8={w8} tail={use_tail}"
            assert not torch.isnan(y1).any() and not torch.isnan(o1).any()
            assert_close(y1, y0, 2e-7, 2e-7, "materialised activation " + what)
            flips = int((m1 != m0).sum())
            assert flips <= rows * Ci // 4 // 100000 + 2, f"ReLU bits {what}: {flips} bytes differ"      # (a value within an ulp of 0)
            assert_close(o1, o0, 1e-5, 1e-5 * float(o0.abs().max()), "convolution output " + what)
            assert_close(part[:, 0].double().sum(0), of.sum(0), 1e-4, 1e-4 * float(of.abs().sum(0).max()), "fused sums " + what)
            assert_close(part[:, 1].double().sum(0), (of * of).sum(0), 1e-4, 1e-6, "fused sums of squares " + what)
            if use_tail:
                assert_close(mo, of.mean(0), 1e-4, 1e-5, "in-kernel mean " + what)
                assert_close(io, 1 / torch.sqrt(of.var(0, unbiased=False) + 1e-5), 2e-4, 0, "in-kernel invstd " + what)
                assert int(tk.abs().sum()) == 0, "tickets must be zero again"
    # refused configurations fail loudly
    dbad = ops.fwd_desc(view_of(ops, xd, B, H, W, Ci), view_of(ops, o0, B, H, W, Co), Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, 1) | (1 << 27))      # (64-deep K tiles)
    rc = L.lib.zsg_conv_igemm_bnpre(C.byref(dbad), xd.data_ptr(), wd.data_ptr(), o0.data_ptr(), None, None, None, None, None, None, 0.1, 1e-5,
                                    md.data_ptr(), isd.data_ptr(), gd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y0.data_ptr(), None, st)
    assert rc != 0
    torch.cuda.synchronize()


def test_conv_multilevel_shared_weights(Z):
    """grouped launch over pyramid levels (shared head), output scattered into the [B, A, 5]-style buffer"""
    L, ops = Z
    g = torch.Generator().manual_seed(5)
    B, Ci, Co, k = 2, 64, 45, 3
    sizes = [(7, 7), (4, 4), (2, 2), (1, 1)]
    w = torch.randn(Co, Ci, k, k, generator=g) / 24
    b = torch.randn(Co, generator=g)
    xs = [torch.randn(B, Ci, h, ww, generator=g) for h, ww in sizes]
    refs = [F.conv2d(x, w, b, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Co) for x in xs]
    ref = torch.cat(refs, dim=1)                     # [B, P, Co]
    P = ref.shape[1]
    packed = torch.cat([nhwc(x).reshape(-1) for x in xs]).cuda()
    lv_in, lv_out, off_in, off_px = [], [], 0, 0
    for (h, ww) in sizes:
        lv_in.append(ops.Level(off_in, h, ww, h * ww * Ci))
        lv_out.append(ops.Level(off_px * Co, h, ww, P * Co))
        off_in += B * h * ww * Ci
        off_px += h * ww
    src = ops.TView(packed, B, Ci, Ci, lv_in)
    out = torch.full((B, P, Co), float("nan"), device="cuda")
    ov = ops.TView(out.view(-1), B, Co, Co, lv_out)
    desc = ops.fwd_desc(src, ov, Ci, Co, k, 1, 1, 1, wC=Ci)
    wd, bd = dev(ohwi(w)), dev(b)
    L.check(L.lib.zsg_conv_igemm(C.byref(desc), packed.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr(), None, None, None, L.stream_ptr()), "igemm")
    assert_close(out, ref, 2e-4, 2e-4, "multi-level conv")
    # wgrad accumulates over the levels
    gy = torch.randn(B, P, Co, generator=g)
    Cop = 48
    gyp = torch.zeros(B, P, Cop)
    gyp[..., :Co] = gy
    gyd = dev(gyp)
    lv_dy = []
    off_px = 0
    for (h, ww) in sizes:
        lv_dy.append(ops.Level(off_px * Cop, h, ww, P * Cop))
        off_px += h * ww
    dyv = ops.TView(gyd.view(-1), B, Cop, Cop, lv_dy)
    dw = torch.zeros(Co, k, k, Ci, device="cuda")
    wdesc = ops.fwd_desc(src, dyv, Ci, Co, k, 1, 1, 1, wC=Ci)
    L.check(L.lib.zsg_conv_wgrad(C.byref(wdesc), packed.data_ptr(), gyd.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, L.stream_ptr()), "wgrad")
    wr = w.clone().requires_grad_()
    tot = sum((F.conv2d(x, wr, None, 1, 1).permute(0, 2, 3, 1).reshape(B, -1, Co) * gy[:, o:o + x.shape[2] * x.shape[3]]).sum()
              for x, o in zip(xs, np.cumsum([0] + [h * ww for h, ww in sizes])[:-1]))
    tot.backward()
    assert_close(dw.permute(0, 3, 1, 2), wr.grad, 5e-4, 5e-4 * float(wr.grad.abs().max()), "multi-level wgrad")


@pytest.mark.parametrize("B,h,w", [(2, 5, 7), (3, 1, 1), (16, 2, 2)])
def test_head_conv0_decomposition(Z, B, h, w):
    """mdl.py:69-104 + :379 — conv0 over [feat | lang (constant over pixels) | grid (constant over batch)] equals the
    feature-window conv plus the additive map of zsg_head_lang_map; backward: the validity-masked sums of
    zsg_head_border_sums / zsg_batch_sum give the lang / grid columns of dW and d(we).  fp32 sums: rel 5e-4."""
    L, ops = Z
    g = torch.Generator().manual_seed(h * 10 + w)
    Cf, Cw, N, k = 32, 24, 64, 3
    Ct = Cf + Cw + 2
    cp = pad4(Ct)
    feat = torch.randn(B, Cf, h, w, generator=g)
    we = torch.randn(B, Cw, generator=g)
    grid = torch.from_numpy(O.create_grid(h, w).reshape(h, w, 2).astype(np.float32)).permute(2, 0, 1)
    x = torch.cat([feat, we[:, :, None, None].expand(B, Cw, h, w), grid[None].expand(B, 2, h, w)], 1).requires_grad_()
    wt = (torch.randn(N, Ct, k, k, generator=g) / 10).requires_grad_()
    bias = torch.randn(N, generator=g)
    y = F.relu(F.conv2d(x, wt, bias, 1, 1))
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dy_ref = (gy * (y > 0)).permute(0, 2, 3, 1).contiguous()            # gradient w.r.t. the pre-ReLU output, NHWC

    W0 = dev(ohwi(wt.detach(), cp))                                        # [N][3][3][cp]
    wed, featd, biasd = dev(we), dev(nhwc(feat)), dev(bias)
    st = L.stream_ptr()
    # V = W0[..., lang] . we  -> [B][N*9]
    V = torch.empty(B, N * 9, device="cuda")
    dv = ops.fwd_desc(view_of(ops, wed, B, 1, 1, Cw), view_of(ops, V, B, 1, 1, N * 9), Cw, N * 9, 1, 1, 0, 1, wC=cp, wt_ld=cp, wc0=Cf)
    L.check(L.lib.zsg_conv_igemm(C.byref(dv), wed.data_ptr(), W0.data_ptr(), V.data_ptr(), None, None, None, None, st), "V")
    Vref = torch.einsum("ntc,bc->bnt", ohwi(wt.detach(), cp).reshape(N, 9, cp)[:, :, Cf:Cf + Cw], we).reshape(B, N * 9)
    assert_close(V, Vref, 2e-4, 2e-4, "V")
    # G = conv(grid, W0[..., grid])
    gm = torch.zeros(h, w, 4)
    gm[..., :2] = grid.permute(1, 2, 0)
    gmd = dev(gm)
    G = torch.empty(h * w, N, device="cuda")
    dg = ops.fwd_desc(view_of(ops, gmd, 1, h, w, 4), view_of(ops, G, 1, h, w, N), 4, N, 3, 1, 1, 1, wC=cp, wc0=Cf + Cw)
    L.check(L.lib.zsg_conv_igemm(C.byref(dg), gmd.data_ptr(), W0.data_ptr(), G.data_ptr(), None, None, None, None, st), "G")
    Gref = F.conv2d(grid[None], wt.detach()[:, Cf + Cw:], None, 1, 1)[0].permute(1, 2, 0).reshape(h * w, N)
    assert_close(G, Gref, 2e-4, 2e-4, "G")
    lmap = torch.empty(B, h, w, N, device="cuda")
    L.check(L.lib.zsg_head_lang_map(V.data_ptr(), G.data_ptr(), B, h, w, N, lmap.data_ptr(), st), "lang_map")
    lref = F.conv2d(x.detach()[:, Cf:], wt.detach()[:, Cf:], None, 1, 1).permute(0, 2, 3, 1)
    assert_close(lmap, lref, 2e-4, 2e-4, "lang map")
    out = torch.empty(B, h, w, N, device="cuda")
    d0 = ops.fwd_desc(view_of(ops, featd, B, h, w, Cf), view_of(ops, out, B, h, w, N), Cf, N, 3, 1, 1, 1, wC=cp, wc0=0, relu=True)
    L.check(L.lib.zsg_conv_igemm(C.byref(d0), featd.data_ptr(), W0.data_ptr(), out.data_ptr(), biasd.data_ptr(), lmap.data_ptr(), None, None, st), "conv0")
    assert_close(out, y.permute(0, 2, 3, 1), 3e-4, 3e-4, "conv0 = window conv + map")

    # ---- backward ---------------------------------------------------------------------------------------------------
    dyd = dev(dy_ref)
    dW = torch.full((N, 3, 3, cp), float("nan"), device="cuda")
    dyv = view_of(ops, dyd, B, h, w, N)
    dwf = ops.fwd_desc(view_of(ops, featd, B, h, w, Cf), dyv, Cf, N, 3, 1, 1, 1, wC=cp, wc0=0)
    L.check(L.lib.zsg_conv_wgrad(C.byref(dwf), featd.data_ptr(), dyd.data_ptr(), dW.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgrad feat")
    S = torch.full((2 * B * 9 * N,), float("nan"), device="cuda")
    S2p = S[B * 9 * N:]
    Q = torch.zeros(9 * B * N, device="cuda")
    bg = torch.zeros(N, device="cuda")                          # the finalize kernel accumulates (+=) the bias gradient
    L.check(L.lib.zsg_head_border_sums(dyd.data_ptr(), B, h, w, N, Q.data_ptr(), st), "border sums")
    L.check(L.lib.zsg_head_border_finalize(Q.data_ptr(), B, N, S.data_ptr(), S2p.data_ptr(), bg.data_ptr(), st), "border finalize")
    assert_close(bg, dy_ref.sum((0, 1, 2)), 2e-4, 2e-4 * float(dy_ref.abs().max()) * B, "bias gradient from the image sums")
    valid = torch.zeros(9, h, w)
    for r in range(3):
        for q in range(3):
            ys, xs_ = torch.arange(h) + r - 1, torch.arange(w) + q - 1
            valid[r * 3 + q] = ((ys >= 0) & (ys < h))[:, None] * ((xs_ >= 0) & (xs_ < w))[None, :]
    S1ref = torch.einsum("bhwn,thw->bnt", dy_ref, valid).reshape(B, N * 9)
    assert_close(S[:B * 9 * N].view(B, N * 9), S1ref, 2e-4, 2e-4 * float(S1ref.abs().max()), "S1")
    assert_close(S2p.view(N * 9, B), S1ref.t(), 2e-4, 2e-4 * float(S1ref.abs().max()), "S2")
    S1v = view_of(ops, S[:B * 9 * N], B, 1, 1, N * 9)
    dwl = ops.fwd_desc(view_of(ops, wed, B, 1, 1, Cw), S1v, Cw, N * 9, 1, 1, 0, 1, wC=cp, wt_ld=cp, wc0=Cf)
    L.check(L.lib.zsg_conv_wgrad(C.byref(dwl), wed.data_ptr(), S.data_ptr(), dW.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgrad lang")
    # d(we)[b][c] = sum_{n,tap} S1[b][n*9+tap] W0[n][tap][Cf+c]: a wgrad whose 'pixels' are the N*9 weight rows
    Wrows = ops.TView(W0.view(-1), 1, Cw, cp, [ops.Level(Cf, 1, N * 9, N * 9 * cp)])
    S2v = ops.TView(S, 1, B, B, [ops.Level(B * 9 * N, 1, N * 9, N * 9 * B)])
    gwe = torch.full((B, Cw), float("nan"), device="cuda")
    dwe = ops.fwd_desc(Wrows, S2v, Cw, B, 1, 1, 0, 1, wC=Cw, wt_ld=Cw)
    L.check(L.lib.zsg_conv_wgrad(C.byref(dwe), W0.data_ptr(), S.data_ptr(), gwe.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "dwe")
    gwe_ref = x.grad[:, Cf:Cf + Cw].sum((2, 3))
    assert_close(gwe, gwe_ref, 5e-4, 5e-4 * float(gwe_ref.abs().max()), "d(we)")
    dys = torch.empty(h * w * N, device="cuda")
    L.check(L.lib.zsg_batch_sum(dyd.data_ptr(), B, h * w * N, dys.data_ptr(), st), "batch_sum")
    assert_close(dys.view(h, w, N), dy_ref.sum(0), 2e-4, 2e-4 * float(dy_ref.abs().max()), "batch sum")
    dwg = ops.fwd_desc(view_of(ops, gmd, 1, h, w, 4), view_of(ops, dys, 1, h, w, N), 4, N, 3, 1, 1, 1, wC=cp, wc0=Cf + Cw)
    L.check(L.lib.zsg_conv_wgrad(C.byref(dwg), gmd.data_ptr(), dys.data_ptr(), dW.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "wgrad grid")
    ref = ohwi(wt.grad, cp)
    assert not torch.isnan(dW).any(), "the three window launches must cover every weight column"
    assert_close(dW, ref, 5e-4, 5e-4 * float(ref.abs().max()), "dW0 (feat | lang | grid windows)")


@pytest.mark.parametrize("B,sizes", [(16, [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]), (3, [(5, 7), (1, 4), (2, 1), (1, 1)]), (2, [(4, 4)])])
def test_head_lang_map_packed(Z, B, sizes):
    """zsg_head_lang_map_packed (all pyramid levels in one launch, border-class sums in LDS) against the per-level launch
    zsg_head_lang_map (itself checked against F.conv2d in test_head_conv0_decomposition) and against the definition: fp32
    summation-order accuracy; with and without the grid map; one-row / one-column / single-pixel levels."""
    L, ops = Z
    N = 256
    g = torch.Generator().manual_seed(5 + B)
    V = dev(torch.randn(B, N * 9, generator=g))
    P = sum(h * w for h, w in sizes)
    G = dev(torch.randn(P, N, generator=g))
    hw = torch.tensor([v for s_ in sizes for v in s_], dtype=torch.int32)
    st = L.stream_ptr()
    for useG in (True, False):
        out = torch.full((B * P * N,), float("nan"), device="cuda")
        ref = torch.full((B * P * N,), float("nan"), device="cuda")
        L.check(L.lib.zsg_head_lang_map_packed(V.data_ptr(), G.data_ptr() if useG else None, B, len(sizes), hw.data_ptr(), N, out.data_ptr(), st), "packed")
        p0 = 0
        for h, w in sizes:
            L.check(L.lib.zsg_head_lang_map(V.data_ptr(), G[p0:].data_ptr() if useG else None, B, h, w, N, ref[B * p0 * N:].data_ptr(), st), "per level")
            # definition: out[b][y][x][n] = G + sum over taps inside the image of V[b][n*9 + tap]
            Vc = V.cpu().double().view(B, N, 3, 3)
            want = torch.zeros(B, h, w, N, dtype=torch.float64)
            for r in range(3):
                for q in range(3):
                    ys = [y for y in range(h) if 0 <= y + r - 1 < h]
                    xs = [x for x in range(w) if 0 <= x + q - 1 < w]
                    if ys and xs:
                        want[:, ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] += Vc[:, None, None, :, r, q]
            if useG:
                want += G[p0:p0 + h * w].cpu().double().view(1, h, w, N)
            got = out[B * p0 * N:B * (p0 + h * w) * N].view(B, h, w, N)
            assert_close(got, want, 1e-5, 1e-5, f"packed lang map level {h}x{w}")
            p0 += h * w
        assert not torch.isnan(out).any()
        assert_close(out, ref, 1e-5, 1e-5, "packed vs per-level")
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows,Cc", [(2 * 19 * 19, 64), (3 * 7 * 5, 256), (1000, 2048), (5000, 12)])
def test_batchnorm(Z, rows, Cc):
    L, _ = Z
    g = torch.Generator().manual_seed(rows + Cc)
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.7
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    res = torch.randn(rows, Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    xr, gr, br, rr = x.clone().requires_grad_(), gam.clone().requires_grad_(), bet.clone().requires_grad_(), res.clone().requires_grad_()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    x4 = xr.t().reshape(1, Cc, rows, 1)
    y = F.batch_norm(x4, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    out_ref = F.relu(y + rr.t().reshape(1, Cc, rows, 1))
    gy = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gy)
    st = L.stream_ptr()
    wsb = L.lib.zsg_bn_workspace_bytes(rows, Cc)
    ws = torch.empty(wsb // 4 + 4, device="cuda")
    xd, gd, bd, resd, rmd, rvd = dev(x), dev(gam), dev(bet), dev(res), dev(rm), dev(rv)
    mean, invstd = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    L.check(L.lib.zsg_bn_stats(xd.data_ptr(), rows, Cc, mean.data_ptr(), invstd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), 0.1, 1e-5,
                               ws.data_ptr(), wsb, st), "bn_stats")
    assert_close(mean, x.mean(0), 1e-5, 1e-5, "bn mean")
    assert_close(invstd, 1 / torch.sqrt(x.var(0, unbiased=False) + 1e-5), 1e-4, 0, "bn invstd")
    assert_close(rmd, rm_ref, 1e-5, 1e-6, "running_mean")
    assert_close(rvd, rv_ref, 1e-4, 1e-6, "running_var")
    out = torch.empty(rows, Cc, device="cuda")
    rmask = torch.zeros(rows * Cc // 4, dtype=torch.uint8, device="cuda")
    L.check(L.lib.zsg_bn_apply(xd.data_ptr(), rows, Cc, mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), resd.data_ptr(), 1,
                               out.data_ptr(), rmask.data_ptr(), st), "bn_apply")
    assert_close(out, out_ref.reshape(Cc, rows).t(), 1e-4, 1e-5, "bn out")
    bits = (out.view(-1, 4) > 0).to(torch.uint8)
    assert torch.equal(rmask, bits[:, 0] | (bits[:, 1] << 1) | (bits[:, 2] << 2) | (bits[:, 3] << 3)), "packed ReLU mask (exact)"
    dout = dev(gy.reshape(Cc, rows).t())
    dx, gout = torch.empty(rows, Cc, device="cuda"), torch.empty(rows, Cc, device="cuda")
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    sc = float(xr.grad.abs().max())
    for how, y_ptr, m_ptr in (("mask from output", out.data_ptr(), None), ("packed mask", None, rmask.data_ptr())):
        dg.zero_(), db.zero_()
        L.check(L.lib.zsg_bn_backward(dout.data_ptr(), y_ptr, m_ptr, xd.data_ptr(), rows, Cc, mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(),
                                      dx.data_ptr(), gout.data_ptr(), dg.data_ptr(), db.data_ptr(), 1, ws.data_ptr(), wsb, st), "bn_backward")
        assert_close(dx, xr.grad, 1e-3, 1e-4 * sc, f"bn dx ({how})")
        assert_close(gout, rr.grad, 1e-5, 1e-6, f"bn residual grad ({how})")
        assert_close(dg, gr.grad, 1e-3, 1e-4 * float(gr.grad.abs().max()), f"bn dgamma ({how})")
        assert_close(db, br.grad, 1e-3, 1e-4 * float(br.grad.abs().max()), f"bn dbeta ({how})")
    # eval-mode statistics
    L.check(L.lib.zsg_bn_eval_stats(rmd.data_ptr(), rvd.data_ptr(), Cc, 1e-5, mean.data_ptr(), invstd.data_ptr(), st), "bn_eval_stats")
    assert_close(invstd, 1 / torch.sqrt(rvd.cpu() + 1e-5), 1e-6, 0, "eval invstd")


def test_bn_fold_eval(Z):
    """eval-mode conv + BatchNorm == conv with W*s, bias beta - mean*s (s = gamma / sqrt(var + eps)); rel 1e-6 on the
    folded parameters (one multiply each), conv output vs F.batch_norm(F.conv2d) rel 2e-4."""
    import struct
    L, ops = Z
    g = torch.Generator().manual_seed(3)
    specs = [(8, 3, 12), (16, 1, 64)]                      # (cout, k, cin padded to 4)
    flat, jobs, row0, arena_used = [], [], 0, 0
    nb = sum(c for c, _, _ in specs)
    rm, rv = torch.randn(nb, generator=g), torch.rand(nb, generator=g) + 0.5
    bn_i, off = 0, 0
    layers = []
    for co, k, ci in specs:
        w = torch.randn(co, k, k, ci, generator=g)
        gam, bet = torch.randn(co, generator=g), torch.randn(co, generator=g)
        w_off, g_off, b_off = off, off + w.numel(), off + w.numel() + co
        flat += [w.reshape(-1), gam, bet]
        off = b_off + co
        dst, bias_dst = arena_used, arena_used + w.numel()
        arena_used = bias_dst + co
        jobs.append(struct.pack("<qqqqqiiii", w_off, dst, g_off, b_off, bias_dst, row0, co, k * k * ci, bn_i))
        layers.append((w, gam, bet, bn_i, dst, bias_dst, co, k, ci))
        row0 += co
        bn_i += co
    flatd = dev(torch.cat(flat))
    jd = torch.frombuffer(bytearray(b"".join(jobs)), dtype=torch.uint8).cuda()
    arena = torch.full((arena_used,), float("nan"), device="cuda")
    rmd, rvd = dev(rm), dev(rv)
    L.check(L.lib.zsg_bn_fold(flatd.data_ptr(), rmd.data_ptr(), rvd.data_ptr(), 1e-5, jd.data_ptr(), len(jobs), row0, arena.data_ptr(), L.stream_ptr()),
            "bn_fold")
    a = arena.cpu()
    assert not torch.isnan(a).any()
    for w, gam, bet, bi, dst, bias_dst, co, k, ci in layers:
        sc = gam / torch.sqrt(rv[bi:bi + co] + 1e-5)
        assert_close(a[dst:dst + w.numel()].view_as(w), w * sc.view(-1, 1, 1, 1), 1e-6, 1e-7, "folded weights")
        assert_close(a[bias_dst:bias_dst + co], bet - rm[bi:bi + co] * sc, 1e-6, 1e-6, "folded bias")
    w, gam, bet, bi, dst, bias_dst, co, k, ci = layers[0]
    x = torch.randn(2, ci, 9, 9, generator=g)
    ref = F.batch_norm(F.conv2d(x, w.permute(0, 3, 1, 2), None, 1, 1), rm[bi:bi + co], rv[bi:bi + co], gam, bet, False, 0.1, 1e-5)
    got = F.conv2d(x, a[dst:dst + w.numel()].view_as(w).permute(0, 3, 1, 2), a[bias_dst:bias_dst + co], 1, 1)
    assert_close(got, ref, 2e-4, 2e-4, "folded conv == conv + eval BatchNorm")


@pytest.mark.parametrize("k,s,p,ceil,H,W", [(3, 2, 1, False, 37, 40), (2, 2, 0, False, 30, 30), (2, 2, 0, True, 15, 19), (3, 1, 1, False, 9, 9)])
def test_maxpool(Z, k, s, p, ceil, H, W):
    L, _ = Z
    g = torch.Generator().manual_seed(k * 10 + s)
    B, Cc = 2, 8
    x = torch.relu(torch.randn(B, Cc, H, W, generator=g))          # many exact ties at 0, like post-ReLU maps
    xr = x.clone().requires_grad_()
    y = F.max_pool2d(xr, k, s, p, ceil_mode=ceil)
    Ho, Wo = y.shape[2:]
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = dev(x.permute(0, 2, 3, 1))
    out = torch.empty(B, Ho, Wo, Cc, device="cuda")
    idx = torch.empty(B * Ho * Wo * Cc, dtype=torch.uint8, device="cuda")
    st = L.stream_ptr()
    L.check(L.lib.zsg_maxpool_fwd(xd.data_ptr(), B, H, W, Cc, k, s, p, Ho, Wo, out.data_ptr(), idx.data_ptr(), st), "maxpool")
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), y.detach())
    dx = torch.empty(B, H, W, Cc, device="cuda")
    dyd = dev(gy.permute(0, 2, 3, 1))
    L.check(L.lib.zsg_maxpool_bwd(dyd.data_ptr(), idx.data_ptr(), B, H, W, Cc, k, s, p, Ho, Wo, dx.data_ptr(), st), "maxpool_bwd")
    assert_close(dx.permute(0, 3, 1, 2), xr.grad, 1e-6, 1e-6, "maxpool bwd")


@pytest.mark.parametrize("Hs,Ws,Hd,Wd", [(10, 10, 19, 19), (19, 19, 38, 38), (4, 5, 8, 9), (3, 3, 5, 5)])
def test_upsample_add(Z, Hs, Ws, Hd, Wd):
    L, _ = Z
    g = torch.Generator().manual_seed(Hs * Wd)
    B, Cc = 2, 8
    a, p = torch.randn(B, Cc, Hd, Wd, generator=g), torch.randn(B, Cc, Hs, Ws, generator=g)
    pr = p.clone().requires_grad_()
    out_ref = a + F.interpolate(pr, size=(Hd, Wd))
    gy = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gy)
    ad, pd = dev(a.permute(0, 2, 3, 1)), dev(p.permute(0, 2, 3, 1))
    out = torch.empty(B, Hd, Wd, Cc, device="cuda")
    st = L.stream_ptr()
    L.check(L.lib.zsg_upsample_add_fwd(ad.data_ptr(), pd.data_ptr(), B, Hs, Ws, Hd, Wd, Cc, out.data_ptr(), st), "upsample")
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), out_ref.detach())
    dp = torch.empty(B, Hs, Ws, Cc, device="cuda")
    gyd = dev(gy.permute(0, 2, 3, 1))
    L.check(L.lib.zsg_upsample_add_bwd(gyd.data_ptr(), B, Hs, Ws, Hd, Wd, Cc, dp.data_ptr(), 0, st), "upsample_bwd")
    assert_close(dp.permute(0, 3, 1, 2), pr.grad, 1e-6, 1e-6, "upsample bwd")


def test_small_ops(Z):
    L, _ = Z
    st = L.stream_ptr()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 3, 256, generator=g)
    xd = dev(x)
    o = torch.empty_like(xd)
    L.check(L.lib.zsg_relu_fwd(xd.data_ptr(), x.numel(), o.data_ptr(), st), "relu")
    assert torch.equal(o.cpu(), torch.relu(x))
    gy = torch.randn(x.shape, generator=g)
    dx = dev(torch.ones_like(x))
    gyd_ = dev(gy)
    L.check(L.lib.zsg_relu_bwd(gyd_.data_ptr(), xd.data_ptr(), x.numel(), dx.data_ptr(), 1, st), "relu_bwd")
    assert_close(dx, 1 + gy * (x > 0), 1e-6, 1e-6)
    avg = torch.empty(2, 256, device="cuda")
    L.check(L.lib.zsg_avgpool_fwd(xd.data_ptr(), 2, 9, 256, avg.data_ptr(), st), "avgpool")
    assert_close(avg, x.reshape(2, 9, 256).mean(1), 1e-6, 1e-6)
    ga = torch.randn(2, 256, generator=g)
    dxa = torch.empty(2, 9, 256, device="cuda")
    gad = dev(ga)
    L.check(L.lib.zsg_avgpool_bwd(gad.data_ptr(), 2, 9, 256, dxa.data_ptr(), 0, st), "avgpool_bwd")
    assert_close(dxa, (ga / 9)[:, None, :].expand(2, 9, 256), 1e-6, 1e-7)
    img = torch.rand(2, 3, 13, 11, generator=g)
    n4 = torch.empty(2, 13, 11, 4, device="cuda")
    imgd = dev(img)
    L.check(L.lib.zsg_nchw_to_nhwc4(imgd.data_ptr(), 2, 3, 13, 11, n4.data_ptr(), st), "nhwc4")
    assert torch.equal(n4.cpu()[..., :3], img.permute(0, 2, 3, 1)) and float(n4[..., 3].abs().max()) == 0
    # pad_rows / colsum per group
    src = torch.randn(7, 45, generator=g)
    dst = torch.full((7, 48), float("nan"), device="cuda")
    srcd = dev(src)
    L.check(L.lib.zsg_pad_rows(srcd.data_ptr(), 7, 45, 45, dst.data_ptr(), 48, st), "pad_rows")
    assert torch.equal(dst.cpu()[:, :45], src) and float(dst[:, 45:].abs().max()) == 0
    big = torch.randn(3, 700, 40, generator=g)
    cs = torch.zeros(3, 16, device="cuda")
    bigd = dev(big)
    L.check(L.lib.zsg_colsum(bigd.data_ptr(), 3, 700 * 40, 700, 40, 8, 16, cs.data_ptr(), 0, st), "colsum")
    assert_close(cs, big[:, :, 8:24].sum(1), 1e-4, 1e-4)


def test_l2norm(Z):
    L, _ = Z
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50, 512, generator=g)
    xr = x.clone().requires_grad_()
    y = xr / xr.norm(dim=1, keepdim=True)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = dev(x)
    out, nrm, dx = torch.empty_like(xd), torch.empty(50, device="cuda"), torch.empty_like(xd)
    st = L.stream_ptr()
    L.check(L.lib.zsg_l2norm_fwd(xd.data_ptr(), 50, 512, out.data_ptr(), nrm.data_ptr(), st), "l2norm")
    assert_close(out, y, 1e-5, 1e-6)
    gyd = dev(gy)
    L.check(L.lib.zsg_l2norm_bwd(gyd.data_ptr(), out.data_ptr(), nrm.data_ptr(), 50, 512, dx.data_ptr(), st), "l2norm_bwd")
    assert_close(dx, xr.grad, 1e-4, 1e-5)


def test_adam_matches_torch(Z):
    L, _ = Z
    g = torch.Generator().manual_seed(1)
    n = 4099
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pr], lr=1e-2, betas=(0.9, 0.99))
    pd, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    step = torch.zeros(2, dtype=torch.int32, device="cuda")          # [steps taken, completion ticket]
    for it in range(5):
        gr = torch.randn(n, generator=g)
        pr.grad = gr.clone()
        opt.step()
        grd = dev(gr)
        L.check(L.lib.zsg_adam_step(pd.data_ptr(), grd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-2, 0.9, 0.99, 1e-8, 0.0, 1.0,
                                    step.data_ptr(), L.stream_ptr()), "adam")
    assert step.tolist() == [5, 0]
    assert_close(pd, pr.detach(), 1e-5, 1e-6, "adam params")


def test_lstm_against_golden_and_oracle(Z, gold):
    """BiLSTM forward + BPTT (csrc/lstm.hip + MFMA input GEMMs) vs the reference golden (g7) — rtol 1e-4."""
    L, ops = Z
    gz = gold("g7_lstm")
    sd = {k: v for k, v in O.seeded_state_dict("resnet50", int(gz["seed"][0])).items() if k.startswith("lstm.")}
    qvec, qlens = torch.from_numpy(gz["qvec"]), torch.from_numpy(gz["qlens"])
    h0, c0 = torch.from_numpy(gz["h0"]), torch.from_numpy(gz["c0"])
    B, T, E, H = qvec.shape[0], qvec.shape[1], 300, 128
    st = L.stream_ptr()
    qd, ld = dev(qvec), dev(qlens)
    we = torch.zeros(B, 256, device="cuda")
    saved = {}
    for di, suf in enumerate(["", "_reverse"]):
        Tn = T if di == 0 else 1
        if di == 0:
            xin = qd
        else:
            xin = torch.empty(B, 1, E, device="cuda")
            L.check(L.lib.zsg_lstm_gather_last(qd.data_ptr(), ld.data_ptr(), B, T, E, xin.data_ptr(), st), "gather")
        gin = torch.empty(B, Tn, 4 * H, device="cuda")
        src = ops.TView(xin.view(-1), B, E, E, [ops.Level(0, 1, Tn, Tn * E)])
        gv = ops.TView(gin.view(-1), B, 4 * H, 4 * H, [ops.Level(0, 1, Tn, Tn * 4 * H)])
        d = ops.fwd_desc(src, gv, E, 4 * H, 1, 1, 0, 1, wC=E)
        wih, bih = dev(sd["lstm.weight_ih_l0" + suf]), dev(sd["lstm.bias_ih_l0" + suf])
        whh, bhh = dev(sd["lstm.weight_hh_l0" + suf]), dev(sd["lstm.bias_hh_l0" + suf])
        L.check(L.lib.zsg_conv_igemm(C.byref(d), xin.data_ptr(), wih.data_ptr(), gin.data_ptr(), bih.data_ptr(), None, None, None, st), "lstm_in")
        gates, cst, hprev = (torch.zeros(B, Tn, 4 * H, device="cuda"), torch.zeros(B, Tn, H, device="cuda"), torch.zeros(B, Tn, H, device="cuda"))
        h0d, c0d = dev(h0[di]), dev(c0[di])
        L.check(L.lib.zsg_lstm_fwd(gin.data_ptr(), whh.data_ptr(), bhh.data_ptr(), h0d.data_ptr(), c0d.data_ptr(), ld.data_ptr(),
                                   ld.data_ptr() if di == 0 else None, B, Tn, H, gates.data_ptr(), cst.data_ptr(), hprev.data_ptr(),
                                   we.data_ptr(), 256, di * H, st), "lstm_fwd")
        saved[suf] = (xin, src, gates, cst, hprev, whh, c0d, Tn)
    assert_close(we, torch.from_numpy(gz["we"]), 1e-4, 1e-5, "lstm output")
    gwd = dev(torch.from_numpy(gz["gw"]))
    for di, suf in enumerate(["", "_reverse"]):
        xin, src, gates, cst, hprev, whh, c0d, Tn = saved[suf]
        dg = torch.full((B, Tn, 4 * H), float("nan"), device="cuda")
        L.check(L.lib.zsg_lstm_bwd(gwd.data_ptr(), 256, di * H, whh.data_ptr(), gates.data_ptr(), cst.data_ptr(), c0d.data_ptr(), ld.data_ptr(),
                                   ld.data_ptr() if di == 0 else None, B, Tn, H, dg.data_ptr(), st), "lstm_bwd")
        dgv = ops.TView(dg.view(-1), B, 4 * H, 4 * H, [ops.Level(0, 1, Tn, Tn * 4 * H)])
        dwih = torch.zeros(4 * H, E, device="cuda")
        L.check(L.lib.zsg_conv_wgrad(C.byref(ops.fwd_desc(src, dgv, E, 4 * H, 1, 1, 0, 1, wC=E)), xin.data_ptr(), dg.data_ptr(), dwih.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "w_ih")
        hp = ops.TView(hprev.view(-1), B, H, H, [ops.Level(0, 1, Tn, Tn * H)])
        dwhh = torch.zeros(4 * H, H, device="cuda")
        L.check(L.lib.zsg_conv_wgrad(C.byref(ops.fwd_desc(hp, dgv, H, 4 * H, 1, 1, 0, 1, wC=H)), hprev.data_ptr(), dg.data_ptr(), dwhh.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, st), "w_hh")
        db = torch.zeros(4 * H, device="cuda")
        L.check(L.lib.zsg_colsum(dg.data_ptr(), 1, 0, B * Tn, 4 * H, 0, 4 * H, db.data_ptr(), 0, st), "bias")
        tag = "l0" + suf
        assert_close(dwih[::4, ::3], torch.from_numpy(gz["grad_weight_ih_" + tag]), 1e-3, 1e-5, "d w_ih" + suf)
        assert_close(dwhh[::4, ::3], torch.from_numpy(gz["grad_weight_hh_" + tag]), 1e-3, 1e-5, "d w_hh" + suf)
        assert_close(db, torch.from_numpy(gz["grad_bias_ih_" + tag]), 1e-3, 1e-5, "d b_ih" + suf)
        assert_close(db, torch.from_numpy(gz["grad_bias_hh_" + tag]), 1e-3, 1e-5, "d b_hh" + suf)


@pytest.mark.parametrize("shape", [(2, 64, 37, 41), (3, 16, 12, 12), (1, 64, 150, 150)], ids=["odd", "small", "stem"])
def test_stem_bn_relu_maxpool_fused(Z, shape):
    """zsg_bn_relu_maxpool_fwd / _bwd == nn.BatchNorm2d(train) -> ReLU -> MaxPool2d(3, 2, 1) (mdl.py:149-152) and its autograd
    backward: pooled values rel 5e-4, window indices exact wherever the window's top two values differ by more than rounding,
    dx / dgamma / dbeta rel 1e-3 of their scale (fp32 sums)."""
    L, ops = Z
    B, Cc, H, W = shape
    g = torch.Generator().manual_seed(31 + H)
    x = (torch.randn(B, Cc, H, W, generator=g) * 1.5 + 0.4).requires_grad_()
    gam = (torch.rand(Cc, generator=g) + 0.5).requires_grad_()
    bet = (0.3 * torch.randn(Cc, generator=g)).requires_grad_()
    a = F.relu(F.batch_norm(x, None, None, gam, bet, True, 0.1, 1e-5))
    out_ref, idx_ref = F.max_pool2d(a, 3, 2, 1, return_indices=True)
    Ho, Wo = out_ref.shape[2:]
    gy = torch.randn(out_ref.shape, generator=g)
    out_ref.backward(gy)
    st = L.stream_ptr()
    xd = dev(nhwc(x.detach()))
    xf = x.detach().permute(0, 2, 3, 1).reshape(-1, Cc).double()
    mean, invstd = dev(xf.mean(0).float()), dev((1 / torch.sqrt(xf.var(0, unbiased=False) + 1e-5)).float())
    gd, bd = dev(gam.detach()), dev(bet.detach())
    out = torch.full((B, Ho, Wo, Cc), float("nan"), device="cuda")
    idx = torch.zeros(B * Ho * Wo * Cc, dtype=torch.uint8, device="cuda")
    L.check(L.lib.zsg_bn_relu_maxpool_fwd(xd.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 3, 2, 1, Ho, Wo,
                                          out.data_ptr(), idx.data_ptr(), st), "bn_relu_maxpool_fwd")
    assert_close(out.permute(0, 3, 1, 2), out_ref, 5e-4, 5e-4, "fused stem forward")
    # window code -> flat input index, compared with torch's where the maximum is unique beyond rounding
    code = idx.view(B, Ho, Wo, Cc).permute(0, 3, 1, 2).cpu().long()
    ho = torch.arange(Ho).view(1, 1, Ho, 1)
    wo = torch.arange(Wo).view(1, 1, 1, Wo)
    flat = (ho * 2 - 1 + code // 3) * W + (wo * 2 - 1 + code % 3)
    ap = F.pad(a.detach(), (1, 1, 1, 1), value=-1.0)
    win = ap.unfold(2, 3, 2).unfold(3, 3, 2).reshape(B, Cc, Ho, Wo, 9)
    top2 = win.topk(2, dim=-1).values
    sure = (top2[..., 0] - top2[..., 1]) > 1e-4
    assert torch.equal(flat[sure], idx_ref[sure]), "arg-max window positions"
    # backward
    gyd = dev(nhwc(gy))
    dx = torch.full((B, H, W, Cc), float("nan"), device="cuda")
    dgam, dbet = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    wsb = L.lib.zsg_bn_workspace_bytes(B * Ho * Wo, Cc)
    ws = torch.empty(wsb // 4 + 16, device="cuda")
    L.check(L.lib.zsg_bn_relu_maxpool_bwd(gyd.data_ptr(), idx.data_ptr(), xd.data_ptr(), B, H, W, Cc, mean.data_ptr(), invstd.data_ptr(), gd.data_ptr(),
                                          bd.data_ptr(), 3, 2, 1, Ho, Wo, dx.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), 0, ws.data_ptr(), wsb, st),
            "bn_relu_maxpool_bwd")
    assert_close(dx.permute(0, 3, 1, 2), x.grad, 1e-3, 1e-3 * float(x.grad.abs().max()), "fused stem dx")
    assert_close(dgam, gam.grad, 1e-3, 1e-3 * float(gam.grad.abs().max()), "fused stem dgamma")
    assert_close(dbet, bet.grad, 1e-3, 1e-3 * float(bet.grad.abs().max()), "fused stem dbeta")


@pytest.mark.parametrize("case", [(2, 3, 64, 45, 37, 7, 2, 3, 40), (2, 256, 64, 20, 17, 1, 1, 0, 9), (1, 20, 48, 13, 13, 3, 1, 1, 1)],
                         ids=["stem7x7", "l1conv1", "ragged"])
def test_wgrad_256_column_tile(Z, case):
    """zsg_conv_wgrad with the 256-column tile (tile_hint BN field 255): all weight columns of a <= 64-output-channel layer from one
    block — the 7x7x4 stem (196 columns), layer1's 256 -> 64 1x1, a ragged 48 x 180 case — vs torch's weight gradient."""
    L, ops = Z
    B, Ci, Co, H, W, k, s, p, splits = case
    g = torch.Generator().manual_seed(3 + Ci + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), gy, stride=s, padding=p)
    cp, Cop = pad4(Ci), pad4(Co)
    xd, dyd = dev(nhwc(x)), dev(nhwc(gy, Cop))
    src, dyv = view_of(ops, xd, B, H, W, cp), view_of(ops, dyd, B, Ho, Wo, Cop)
    d = ops.fwd_desc(src, dyv, cp, Co, k, s, p, 1, wC=cp, tile_hint=ops.tile_hint(64, 255, splits))
    dw = torch.zeros(Co, k, k, cp, device="cuda")
    L.check(L.lib.zsg_conv_wgrad(C.byref(d), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, L.stream_ptr()), "wgrad 64x256")
    assert_close(dw[..., :Ci].permute(0, 3, 1, 2), ref, 5e-4, 5e-4 * float(ref.abs().max()), "wgrad, 256-column tile")


@pytest.mark.parametrize("B,Ci,Co,sizes", [(2, 64, 96, [(19, 19)]), (3, 256, 64, [(7, 5), (3, 3), (1, 1)]), (16, 128, 512, [(5, 5)]), (1, 36, 20, [(1, 1)]),
                                           (2, 64, 256, [(38, 38), (19, 19), (10, 10)])], ids=["one", "levels", "wide", "tiny", "pyramid"])
def test_wgrad_dense_1x1_loader(Z, B, Ci, Co, sizes):
    """1x1 / stride-1 weight gradient over batch-dense tensors = the DENSE loader of wgrad_kernel (descriptor-base addressing, the level's
    ragged last K tile and the dead prefetch cut by num_records): every tile / wave / K-depth variant and split count, ragged row counts
    (361 * B, 35 * B, 1), ragged channel counts (Co = 20 in a 4-padded row), several levels in one launch, accumulate on / off."""
    L, ops = Z
    g = torch.Generator().manual_seed(7 + Ci + Co)
    xs = [torch.randn(B, Ci, h, w, generator=g) for h, w in sizes]
    gys = [torch.randn(B, Co, h, w, generator=g) for h, w in sizes]
    ref = sum(torch.einsum("bohw,bihw->oi", gy.double(), x.double()) for gy, x in zip(gys, xs)).float()
    Cop = pad4(Co)
    packed_x = torch.cat([nhwc(x).reshape(-1) for x in xs]).cuda()
    packed_g = torch.cat([nhwc(gy, Cop).reshape(-1) for gy in gys]).cuda()
    lv_x, lv_g, ox, og = [], [], 0, 0
    for h, w in sizes:
        lv_x.append(ops.Level(ox, h, w, h * w * Ci))
        lv_g.append(ops.Level(og, h, w, h * w * Cop))
        ox += B * h * w * Ci
        og += B * h * w * Cop
    src, dyv = ops.TView(packed_x, B, Ci, Ci, lv_x), ops.TView(packed_g, B, Cop, Cop, lv_g)
    st = L.stream_ptr()
    hints = [0] + [ops.tile_hint(bm, bn, sp, w8, k32) for bm, bn in ((64, 64), (128, 64), (64, 128), (128, 128)) for sp in (1, 3, 16)
                   for w8 in ((0, 1) if (bm, bn) == (128, 128) else (0,)) for k32 in (0, 1)]
    for i, hint in enumerate(hints):
        acc = i & 1
        dw = torch.full((Co, 1, 1, Ci), 1.0, device="cuda")
        d = ops.fwd_desc(src, dyv, Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
        L.check(L.lib.zsg_conv_wgrad(C.byref(d), packed_x.data_ptr(), packed_g.data_ptr(), dw.data_ptr(), acc, WS.data_ptr(), WS.numel() * 4, st), "wgrad dense")
        assert_close(dw.view(Co, Ci).cpu() - (1.0 if acc else 0.0), ref, 5e-4, 5e-4 * float(ref.abs().max()), f"dense 1x1 wgrad hint {hint:x} acc={acc}")


@pytest.mark.parametrize("case", [(2, 32, 16, 520, 520, 3, 1, 1, (128, 128, 24)), (2, 8, 32, 520, 520, 1, 1, 0, (64, 64, 64)), (2, 32, 16, 520, 520, 3, 1, 1, (0, 0, 0))],
                         ids=["src_wide", "dy_wide", "heuristic"])
def test_wgrad_image_stride_beyond_2p23(Z, case):
    """zsg_conv_wgrad on activations of more than 2^23 elements PER IMAGE (the stem / layer1 maps of inputs beyond ~724x724; here 32 x
    520 x 520): the loader's 24-bit multiplies cannot form batch-index x image-stride, the library falls back to its 32-bit-multiply
    variant of the 64x64 tile whatever tile the hint names (round 4 refused the launch; the reference accepts any resize_img,
    dat_loader.py:121) — vs torch's weight gradient."""
    L, ops = Z
    B, Ci, Co, H, W, k, s, p, (bm, bn, splits) = case
    g = torch.Generator().manual_seed(11 + Ci)
    x = torch.randn(B, Ci, H, W, generator=g)
    Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x, (Co, Ci, k, k), gy, stride=s, padding=p)
    cp, Cop = pad4(Ci), pad4(Co)
    assert max(cp * H * W, Cop * Ho * Wo) >= (1 << 23)
    xd, dyd = dev(nhwc(x)), dev(nhwc(gy, Cop))
    src, dyv = view_of(ops, xd, B, H, W, cp), view_of(ops, dyd, B, Ho, Wo, Cop)
    d = ops.fwd_desc(src, dyv, cp, Co, k, s, p, 1, wC=cp, tile_hint=ops.tile_hint(bm, bn, splits) if bm else 0)
    dw = torch.zeros(Co, k, k, cp, device="cuda")
    L.check(L.lib.zsg_conv_wgrad(C.byref(d), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, WS.data_ptr(), WS.numel() * 4, L.stream_ptr()), "wgrad wide")
    assert_close(dw[..., :Ci].permute(0, 3, 1, 2), ref, 1e-3, 1e-3 * float(ref.abs().max()), "wgrad, image stride >= 2^23")


def test_completion_event_orders_another_stream(Z):
    """zsg_set_completion_event / zsg_stream_wait_event (include/zsg.h): a launch on stream A carries the armed event as its completion
    signal and stream B, made to wait for it, sees everything that launch wrote — without a marker in A's queue.  Also: the disarming
    call reports how many launches carried the event (0 for a call that launches nothing: the caller then records a marker)."""
    L, _ = Z
    lib = L.lib
    n = 64 << 20                                        # 256 MB: the fill takes ~50 us, a racing reader would see stale zeros
    a = torch.zeros(n, device="cuda")
    b = torch.full((n,), -1.0, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ev = lib.zsg_event_create()
    assert ev
    ev = C.c_void_p(ev)
    torch.cuda.synchronize()
    for rep in range(3):
        val = float(rep + 1)
        assert lib.zsg_set_completion_event(ev) >= 0
        L.check(lib.zsg_memset_f32(a.data_ptr(), n, val, C.c_void_p(sa.cuda_stream)), "memset")
        assert lib.zsg_set_completion_event(None) == 1  # one launch carried the event
        L.check(lib.zsg_stream_wait_event(C.c_void_p(sb.cuda_stream), ev), "wait")
        L.check(lib.zsg_relu_fwd(a.data_ptr(), n, b.data_ptr(), C.c_void_p(sb.cuda_stream)), "relu")     # b = max(a, 0) on stream B
        sb.synchronize()
        assert float(b.min()) == val and float(b.max()) == val, (rep, float(b.min()), float(b.max()))
    # a call that launches nothing reports 0 uses; the marker fallback orders the streams just the same
    lib.zsg_set_completion_event(ev)
    L.check(lib.zsg_memset_f32(a.data_ptr(), 0, 0.0, C.c_void_p(sa.cuda_stream)), "empty memset")
    assert lib.zsg_set_completion_event(None) == 0
    L.check(lib.zsg_memset_f32(a.data_ptr(), n, 7.0, C.c_void_p(sa.cuda_stream)), "memset")
    L.check(lib.zsg_event_record(ev, C.c_void_p(sa.cuda_stream)), "record")
    L.check(lib.zsg_stream_wait_event(C.c_void_p(sb.cuda_stream), ev), "wait")
    L.check(lib.zsg_relu_fwd(a.data_ptr(), n, b.data_ptr(), C.c_void_p(sb.cuda_stream)), "relu")
    sb.synchronize()
    assert float(b.min()) == 7.0 and float(b.max()) == 7.0
    torch.cuda.synchronize()
    L.check(lib.zsg_event_destroy(ev), "destroy")
