"""Stream-K implicit GEMM (zsg_conv_igemm with tile_hint bits 28-29, csrc/igemm.hip template flag SK; reference: nn.Conv2d forward /
autograd's data gradient of the layer3 / layer4 bottleneck 1x1 convolutions, fpn_resnet.py:86-100).

A launch of 256 x (1..3) workgroups shares the (tile, K step) units evenly; a tile that is cut between workgroups is completed by the
one holding its first K step, which adds the others' partial accumulator tiles in workgroup order.  Checked here, through the C ABI:
  * values against torch-CPU fp32 F.conv2d at the tolerance of the plain tiles (the K sum is only regrouped), for every stream-K tile
    variant, at the bench shapes (layer3 conv1, layer4 conv1 / conv3, a strided 3x3) and on ragged shapes with row / column tails;
  * the complete epilogue behind the hand-off: bias + ReLU, accumulate + mask, fused BatchNorm statistics (partial rows = the fp64
    column sums of the STORED output) and the BatchNorm-backward sums of zsg_conv_igemm_bnb;
  * determinism: repeated launches — alone and under a noisy neighbour stream — are bit-identical; the hand-off flags read zero after
    every launch;
  * refusals: no registered scratch (-1), scratch too small (-2), more tiles than workgroups, several segments, split-K — all loud.
In-kernel BatchNorm finalize and the bnb-tail variant on stream-K launches: tests/test_gpu_bntail.py / test_gpu_bnb.py cases.
The Winograd kernel's stream-K form (zsg_conv_wino, 32 tiles x 64 channels, four position groups; csrc/wino.hip) exchanges partial
OUTPUT tiles: test_wino_stream_k below, same checks."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import Z, assert_close, dev, nhwc, ohwi, pad4, stream_scratch, view_of  # noqa: F401
from test_gpu_wino import make_u  # noqa: F401

pytestmark = pytest.mark.gpu

SK = 28
VARIANTS = [
    # bm, bn, w8, k64
    (64, 64, 0, 0), (64, 64, 1, 0), (64, 64, 1, 1), (128, 64, 1, 0), (128, 128, 1, 0), (128, 128, 0, 0),
]
SHAPES = [
    # B, Cin, Cout, H, W, k, stride
    (16, 1024, 256, 19, 19, 1, 1),       # layer3 conv1 at the bench shape: 364 tiles of 64x64 — every tile is cut
    (16, 2048, 512, 10, 10, 1, 1),       # layer4 conv1: K = 2048, 200 tiles
    (16, 512, 2048, 10, 10, 1, 1),       # layer4 conv3: short K (16 steps), 800 tiles of 64x64 -> only the 128-row tiles qualify
    (16, 256, 256, 38, 38, 3, 2),        # layer3.0 conv2: strided 3x3, K = 9 x 256
    (3, 96, 80, 13, 11, 1, 1),           # ragged: row tail, column tail (N = 80), channel tail (C = 96: the 64-deep tile's last step is half empty)
    (2, 64, 64, 7, 5, 3, 1),             # fewer units than workgroups: most of the grid idles
]


def hint_of(ops, v, bpc):
    bm, bn, w8, k64 = v
    return ops.tile_hint(bm, bn, 1, w8) | (k64 << 27) | (bpc << SK)


def flags_zero(L):
    ws = stream_scratch(L)
    return int(ws[:4096].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("shape", SHAPES, ids=[f"s{i}" for i in range(len(SHAPES))])
def test_stream_k_forward_matches_torch_and_is_deterministic(Z, shape):
    L, ops = Z
    B, Ci, Co, H, W, k, s = shape
    g = torch.Generator().manual_seed(11 + Ci + Co + H)
    p = k // 2
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    y_lin = F.conv2d(x, w, None, s, p)
    Ho, Wo = y_lin.shape[2:]
    rows = B * Ho * Wo
    y_ref = F.relu(y_lin + b.view(1, -1, 1, 1))
    cp = pad4(Ci)
    xd, wd, bd = dev(nhwc(x)), dev(ohwi(w)), dev(b)
    src = view_of(ops, xd, B, H, W, cp)
    st = L.stream_ptr()
    side = torch.cuda.Stream()
    noise = torch.empty(32 << 20, device="cuda")
    ran = 0
    for v in VARIANTS:
        bm, bn, w8, k64 = v
        tiles = ((rows + bm - 1) // bm) * ((Co + bn - 1) // bn)
        for bpc in (1, 2, 3):
            out = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
            ov = view_of(ops, out, B, Ho, Wo, Co)
            hint = hint_of(ops, v, bpc)
            d = ops.fwd_desc(src, ov, cp, Co, k, s, p, 1, wC=cp, relu=True, tile_hint=hint)
            rc = L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), bd.data_ptr(), None, None, None, st)
            if bpc == 3 and bm != 64:
                continue                                  # (three 128-row workgroups per CU: more scratch than the test registers)
            applicable = tiles <= 256 * bpc
            if not applicable:
                assert rc != 0, f"stream-K hint {hint:x} must be refused ({tiles} tiles)"
                continue
            L.check(rc, f"stream-K {v} x{bpc}")
            ran += 1
            assert_close(out.permute(0, 3, 1, 2), y_ref, 2e-4, 2e-4, f"stream-K fwd bias+relu {v} x{bpc}")
            assert flags_zero(L), "hand-off flags must read zero after the launch"
            # fused BatchNorm statistics of the finished tiles + bit-identical reruns under an uneven load
            chunks = (rows + bm - 1) // bm
            d2 = ops.fwd_desc(src, ov, cp, Co, k, s, p, 1, wC=cp, tile_hint=hint)
            assert ops.igemm_partial_rows(d2) == chunks
            first = None
            for rep in range(3):
                part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
                out.fill_(float("nan"))
                torch.cuda.synchronize()
                if rep:
                    L.check(L.lib.zsg_memset_f32(noise.data_ptr(), noise.numel(), float(rep), C.c_void_p(side.cuda_stream)), "noise")
                L.check(L.lib.zsg_conv_igemm(C.byref(d2), xd.data_ptr(), wd.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "stream-K + stats")
                torch.cuda.synchronize()
                if first is None:
                    first = (out.clone(), part.clone())
                    assert_close(out.permute(0, 3, 1, 2), y_lin, 2e-4, 2e-4, f"stream-K fwd {v} x{bpc}")
                    yo = out.reshape(-1, Co).double()
                    assert_close(part[:, 0].double().sum(0), yo.sum(0), 1e-5, 1e-5 * float(yo.abs().sum(0).max()), "partial sums = column sums of the stored output")
                    assert_close(part[:, 1].double().sum(0), (yo * yo).sum(0), 1e-5, 1e-7, "partial sums of squares")
                else:
                    assert torch.equal(out, first[0]) and torch.equal(part, first[1]), f"stream-K {v} x{bpc}: rerun {rep} differs"
            assert flags_zero(L)
    assert ran > 0
    side.synchronize()


def test_stream_k_data_gradient_accumulate_mask_and_bn_backward_sums(Z):
    """The data gradient of layer3's conv3 (1024 -> 256 over 5776 pixels, K = 1024) as a stream-K launch: plain, accumulate + float
    mask, and with the BatchNorm-backward sums in its epilogue (zsg_conv_igemm_bnb) — stored dout bit-identical to the plain stream-K
    launch, sums against fp64."""
    L, ops = Z
    B, Ci, Co, H, W = 16, 256, 1024, 19, 19           # forward conv Ci -> Co; the data gradient reduces over Co
    g = torch.Generator().manual_seed(77)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Co ** 0.5
    dy = torch.randn(B, Co, H, W, generator=g)
    dxr = torch.nn.grad.conv2d_input((B, Ci, H, W), w, dy).permute(0, 2, 3, 1)
    st = L.stream_ptr()
    dyd, wd = dev(nhwc(dy)), dev(ohwi(w))
    wt = torch.empty((Ci, 1, 1, Co), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, 1, Ci, Co, st), "transpose_w")
    dyv = view_of(ops, dyd, B, H, W, Co)
    rows = B * H * W
    prev, mask = torch.randn(B, H, W, Ci, generator=g), torch.randn(B, H, W, Ci, generator=g)
    xbn = torch.randn(B, H, W, Ci, generator=g) * 2 + 0.5
    mean, invstd = torch.randn(Ci, generator=g) * 0.3, torch.rand(Ci, generator=g) + 0.5
    bits = torch.rand(B, H, W, Ci, generator=g) > 0.4
    bb = bits.reshape(-1, 4).to(torch.uint8)
    maskb = dev((bb[:, 0] | (bb[:, 1] << 1) | (bb[:, 2] << 2) | (bb[:, 3] << 3)).contiguous())
    xd, md, isd = dev(xbn), dev(mean), dev(invstd)
    for v, bpc in (((64, 64, 1, 1), 2), ((128, 64, 1, 0), 1), ((128, 128, 1, 0), 2), ((64, 64, 0, 0), 2)):
        hint = hint_of(ops, v, bpc)
        dx = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        dd = ops.dgrad_desc(dyv, view_of(ops, dx, B, H, W, Ci), Co, Ci, 1, 1, 0, 1, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm(C.byref(dd), dyd.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st), "stream-K dgrad")
        assert_close(dx, dxr, 5e-4, 5e-4 * float(dxr.abs().max()), f"stream-K dgrad {v}")
        dx2, maskd = dev(prev), dev(mask)
        L.check(L.lib.zsg_conv_igemm(C.byref(dd), dyd.data_ptr(), wt.data_ptr(), dx2.data_ptr(), None, dx2.data_ptr(), maskd.data_ptr(), None, st), "stream-K dgrad+")
        ref2 = (prev + dxr) * (mask > 0)
        assert_close(dx2, ref2, 5e-4, 5e-4 * float(ref2.abs().max()), f"stream-K dgrad accumulate+mask {v}")
        # BatchNorm-backward sums in the epilogue
        chunks = (rows + v[0] - 1) // v[0]
        part = torch.full((chunks, 2, Ci), float("nan"), device="cuda")
        dx3 = torch.full((B, H, W, Ci), float("nan"), device="cuda")
        d3 = ops.dgrad_desc(dyv, view_of(ops, dx3, B, H, W, Ci), Co, Ci, 1, 1, 0, 1, tile_hint=hint)
        L.check(L.lib.zsg_conv_igemm_bnb(C.byref(d3), dyd.data_ptr(), wt.data_ptr(), dx3.data_ptr(), None, xd.data_ptr(), md.data_ptr(), isd.data_ptr(),
                                         maskb.data_ptr(), part.data_ptr(), st), "stream-K dgrad + bnb")
        torch.cuda.synchronize()
        assert torch.equal(dx3, dx), "the stored dout of the bnb launch = the plain stream-K launch"
        gd = dx.double().cpu() * bits.double()
        xhat = (xbn.double() - mean.double()) * invstd.double()
        s1, s2 = gd.reshape(-1, Ci).sum(0), (gd * xhat).reshape(-1, Ci).sum(0)
        assert_close(part[:, 0].double().sum(0), s1, 1e-4, 1e-4 * float(gd.abs().reshape(-1, Ci).sum(0).max()), "sum g")
        assert_close(part[:, 1].double().sum(0), s2, 1e-4, 1e-4 * float((gd * xhat).abs().reshape(-1, Ci).sum(0).max()), "sum g xhat")
        assert flags_zero(L)


def test_stream_k_refusals(Z):
    """No scratch registered for the stream -> -1; scratch too small -> -2; several segments / split-K / the streaming 1x1 hint with
    the stream-K bits -> -1.  Nothing falls back silently."""
    L, ops = Z
    B, Ci, Co, H, W = 2, 64, 64, 9, 9
    xd = torch.zeros(B, H, W, Ci, device="cuda")
    y = torch.zeros(B, H, W, Co, device="cuda")
    wd = torch.zeros(Co, 1, 1, Ci, device="cuda")
    src, out = view_of(ops, xd, B, H, W, Ci), view_of(ops, y, B, H, W, Co)
    hint = ops.tile_hint(64, 64, 1) | (1 << SK)
    d = ops.fwd_desc(src, out, Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
    other = torch.cuda.Stream()
    with torch.cuda.stream(other):
        rc = L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), y.data_ptr(), None, None, None, None, C.c_void_p(other.cuda_stream))
        assert rc == -1 and b"zsg_set_stream_workspace" in L.lib.zsg_last_error()
        small = torch.zeros((16 << 10) // 4 + 1024, device="cuda")
        L.check(L.lib.zsg_set_stream_workspace(C.c_void_p(other.cuda_stream), small.data_ptr(), small.numel() * 4), "register")
        rc = L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), y.data_ptr(), None, None, None, None, C.c_void_p(other.cuda_stream))
        assert rc == -2, L.lib.zsg_last_error()
        L.check(L.lib.zsg_set_stream_workspace(C.c_void_p(other.cuda_stream), None, 0), "unregister")
        rc = L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), y.data_ptr(), None, None, None, None, C.c_void_p(other.cuda_stream))
        assert rc == -1
    other.synchronize()
    st = L.stream_ptr()
    for bad in (ops.tile_hint(64, 64, 2) | (1 << SK), ops.tile_hint(32, 64, 1) | (1 << SK), ops.tile_hint(128, 64, 1) | (1 << SK)):
        d.tile_hint = bad             # split-K + stream-K; the streaming 1x1 kernel; a tile without a stream-K variant (128x64, 4 waves)
        assert L.lib.zsg_conv_igemm(C.byref(d), xd.data_ptr(), wd.data_ptr(), y.data_ptr(), None, None, None, None, st) == -1, hex(bad)
    # a strided data gradient has one segment per parity class: not a stream-K geometry
    dy = torch.zeros(B, 5, 5, Co, device="cuda")
    dx = torch.zeros(B, H, W, Ci, device="cuda")
    wt = torch.zeros(Ci, 3, 3, Co, device="cuda")
    dd = ops.dgrad_desc(view_of(ops, dy, B, 5, 5, Co), view_of(ops, dx, B, H, W, Ci), Co, Ci, 3, 2, 1, 1, tile_hint=hint)
    assert dd.nseg > 1
    assert L.lib.zsg_conv_igemm(C.byref(dd), dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), None, None, None, None, st) == -1
    torch.cuda.synchronize()


WINO_SK = 32 | (64 << 8) | (1 << 16) | (1 << 24) | (1 << SK)
WINO_SHAPES = [
    # B, Cin, Cout, H, W, bias, relu
    (16, 256, 256, 19, 19, False, False),     # layer3 conv2 at the bench shape: 184 tiles x 32 chunks over 256 workgroups
    (16, 512, 512, 10, 10, False, False),     # layer4 conv2: 104 tiles x 64 chunks
    (16, 256, 256, 19, 19, True, True),       # P4_2-like: bias + ReLU behind the hand-off
    (3, 40, 96, 7, 10, True, False),          # ragged: tile tail, column tail (N = 96), C % 8 == 4 (last chunk half dead), 5 chunks
    (2, 8, 64, 9, 9, False, False),           # one chunk per tile: nothing is cut, most workgroups idle
]


@pytest.mark.parametrize("shape", WINO_SHAPES, ids=[f"w{i}" for i in range(len(WINO_SHAPES))])
def test_wino_stream_k(Z, shape):
    """Values vs torch-CPU fp32 (tolerance of test_gpu_wino.py: 3e-4 of the output scale), fused BatchNorm partial rows = column sums of
    the stored output, bit-identical reruns under a noisy neighbour, flags zero after every launch; the data-gradient form (rotated
    filters) with accumulate + mask."""
    L, ops = Z
    B, Ci, Co, H, W, bias, relu = shape
    g = torch.Generator().manual_seed(23 + Ci + Co + H)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    b = torch.randn(Co, generator=g) if bias else None
    y_ref = F.conv2d(x, w, b, 1, 1)
    if relu:
        y_ref = F.relu(y_ref)
    cp = pad4(Ci)
    st = L.stream_ptr()
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    U = make_u(L, ops, wd, Co, cp, 9 * cp, cp, False)
    out = torch.full((B, H, W, Co), float("nan"), device="cuda")
    src, ov = view_of(ops, xd, B, H, W, cp), view_of(ops, out, B, H, W, Co)
    d = ops.fwd_desc(src, ov, cp, Co, 3, 1, 1, 1, wC=cp, relu=relu, tile_hint=WINO_SK)
    bd = dev(b) if bias else None
    L.check(L.lib.zsg_conv_wino(C.byref(d), xd.data_ptr(), U.data_ptr(), out.data_ptr(), bd.data_ptr() if bias else None, None, None, None, st), "wino stream-K")
    sc = float(y_ref.abs().max())
    assert_close(out.permute(0, 3, 1, 2), y_ref, 3e-4, 3e-4 * sc, "wino stream-K fwd")
    assert flags_zero(L)
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    chunks = (tiles + 31) // 32
    if not bias and not relu:
        side = torch.cuda.Stream()
        noise = torch.empty(32 << 20, device="cuda")
        first = None
        for rep in range(4):
            part = torch.full((chunks, 2, Co), float("nan"), device="cuda")
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            if rep >= 2:
                L.check(L.lib.zsg_memset_f32(noise.data_ptr(), noise.numel(), float(rep), C.c_void_p(side.cuda_stream)), "noise")
            L.check(L.lib.zsg_conv_wino(C.byref(d), xd.data_ptr(), U.data_ptr(), out.data_ptr(), None, None, None, part.data_ptr(), st), "wino stream-K + stats")
            torch.cuda.synchronize()
            if first is None:
                first = (out.clone(), part.clone())
                assert_close(out.permute(0, 3, 1, 2), y_ref, 3e-4, 3e-4 * sc, "wino stream-K fwd + stats")
                yo = out.reshape(-1, Co).double()
                assert_close(part[:, 0].double().sum(0), yo.sum(0), 1e-5, 1e-5 * float(yo.abs().sum(0).max()), "partial sums = column sums of the stored output")
                assert_close(part[:, 1].double().sum(0), (yo * yo).sum(0), 1e-5, 1e-7, "partial sums of squares")
            else:
                assert torch.equal(out, first[0]) and torch.equal(part, first[1]), f"wino stream-K: rerun {rep} differs"
            assert flags_zero(L)
        side.synchronize()
    # data gradient: dy [B, Co] -> dx [B, Ci] with the rotated, transposed filters; accumulate + mask epilogue
    gy = torch.randn(B, Co, H, W, generator=g)
    dxr = torch.nn.grad.conv2d_input((B, Ci, H, W), w, gy, 1, 1).permute(0, 2, 3, 1)
    Cop = pad4(Co)
    dyd = dev(nhwc(gy, Cop))
    wt = torch.empty((cp, 3, 3, Cop), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, 9, cp, Cop, st), "transpose_w")
    Ut = make_u(L, ops, wt, cp, Cop, 9 * Cop, Cop, True)
    prev, mask = torch.randn(B, H, W, cp, generator=g), torch.randn(B, H, W, cp, generator=g)
    dx, maskd = dev(prev), dev(mask)
    dd = ops.dgrad_desc(view_of(ops, dyd, B, H, W, Cop), view_of(ops, dx, B, H, W, cp), Cop, cp, 3, 1, 1, 1, tile_hint=WINO_SK)
    L.check(L.lib.zsg_conv_wino(C.byref(dd), dyd.data_ptr(), Ut.data_ptr(), dx.data_ptr(), None, dx.data_ptr(), maskd.data_ptr(), None, st), "wino stream-K dgrad")
    ref = (prev[..., :Ci] + dxr) * (mask[..., :Ci] > 0)
    assert_close(dx[..., :Ci], ref, 5e-4, 5e-4 * float(ref.abs().max()), "wino stream-K dgrad accumulate+mask")
    assert flags_zero(L)
    # refusals: a tile without a stream-K variant, split-K + stream-K
    for bad in (64 | (64 << 8) | (1 << 16) | (1 << SK), 32 | (64 << 8) | (2 << 16) | (1 << 24) | (1 << SK)):
        if ((bad >> 16) & 0xff) > (cp + 7) // 8:
            continue                    # (the library clamps a split count beyond the chunk count to it: one chunk -> no split left to refuse)
        d.tile_hint = bad
        assert L.lib.zsg_conv_wino(C.byref(d), xd.data_ptr(), U.data_ptr(), out.data_ptr(), None, None, None, None, st) == -1, hex(bad)
    torch.cuda.synchronize()
