"""Forward BatchNorm statistics finalised inside the producing convolution (zsg_conv_igemm_bnstat / zsg_conv_wino_bnstat, csrc/bn_tail.h;
reference: nn.BatchNorm2d in training mode behind the convolutions of fpn_resnet.py:80-100).  The last-arriving tile of each column block
reduces the block's partial rows in fp64 in a fixed order: the stored convolution output and the partial rows must be bit-identical to
the plain fused launch, mean / invstd / running statistics must equal the fp64 reduction of those partial rows rounded once (hence
bit-identical from run to run, whatever the arrival order), and agree with torch's batch statistics of the output to fp32 accuracy.
Repeated under an uneven load (another stream streams HBM) with the same ticket words: they must come back zero every time."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import Z, dev, nhwc, ohwi, pad4, view_of  # noqa: F401
from test_gpu_wino import make_u  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin, Cout, H, W, k, stride, kernel, hint (bm, bn, w8 [, k64]) | (tb, bn, ps4)
    (16, 1024, 256, 19, 19, 1, 1, "igemm", (64, 64, 0)),        # layer3 conv1 at the bench shape: 91 rows x 4 column blocks
    (16, 256, 1024, 19, 19, 1, 1, "igemm", (128, 64, 1)),       # layer3 conv3: 46 rows x 16 column blocks
    (16, 512, 2048, 10, 10, 1, 1, "igemm", (128, 128, 1)),      # layer4 conv3
    (16, 2048, 512, 10, 10, 1, 1, "igemm", (64, 64, 1)),        # layer4 conv1: 8-wave 64x64 (two K groups)
    (16, 1024, 512, 19, 19, 1, 2, "igemm", (64, 64, 0)),        # a strided projection shortcut
    (2, 64, 96, 9, 7, 1, 1, "igemm", (64, 64, 0)),              # ragged: N = 96 (column tail), 2 rows
    (3, 32, 64, 13, 11, 3, 1, "igemm", (128, 64, 0)),           # direct 3x3
    (16, 256, 256, 19, 19, 3, 1, "wino", (32, 64, 1)),          # layer3 conv2: wino_kernel<1,2,4>, 50 x 4
    (16, 256, 256, 19, 19, 3, 1, "wino", (64, 64, 0)),          # wino_kernel<2,2,2>
    (16, 512, 512, 10, 10, 3, 1, "wino", (32, 32, 1)),          # layer4 conv2
    (2, 64, 64, 7, 9, 3, 1, "wino", (64, 32, 0)),
    # stream-K launches (hint: bm, bn, w8, k64, workgroups per CU): the finishing workgroup of a cut tile writes the partial row and arrives
    (16, 1024, 256, 19, 19, 1, 1, "igemm", (64, 64, 1, 1, 2)),   # layer3 conv1: 364 tiles over 512 workgroups
    (16, 1024, 256, 19, 19, 1, 1, "igemm", (128, 128, 1, 0, 1)), # the same as 92 tiles of 128x128 over 256 workgroups
    (16, 2048, 512, 10, 10, 1, 1, "igemm", (128, 64, 1, 0, 1)),  # layer4 conv1
    (16, 2048, 512, 10, 10, 1, 1, "igemm", (64, 64, 0, 0, 1)),   # 200 tiles over 256 workgroups
    (16, 256, 256, 19, 19, 3, 1, "wino", (32, 64, 1, 1)),        # layer3 conv2 as a Winograd stream-K launch (tb, bn, ps4, stream-K)
    (16, 512, 512, 10, 10, 3, 1, "wino", (32, 64, 1, 1)),        # layer4 conv2
]


@pytest.mark.parametrize("case", CASES, ids=[f"t{i}" for i in range(len(CASES))])
def test_conv_with_in_kernel_bn_statistics(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, k, s, kern, hint3 = case
    g = torch.Generator().manual_seed(5 + Ci + Co + H)
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5 * 1.5
    w[:, :, :, :] += 0.02                                        # a non-zero mean per channel
    cp = pad4(Ci)
    xd, wd = dev(nhwc(x)), dev(ohwi(w))
    src = view_of(ops, xd, B, H, W, cp)
    st = L.stream_ptr()
    rows = B * Ho * Wo
    if kern == "wino":
        tb, bn, ps4, skw = (tuple(hint3) + (0,))[:4]
        hint = tb | (bn << 8) | (1 << 16) | (ps4 << 24) | (skw << 28)
        wop = make_u(L, ops, wd, Co, cp, k * k * cp, cp, False)
        fn_plain, fn_tail = L.lib.zsg_conv_wino, L.lib.zsg_conv_wino_bnstat
        chunks = (B * ((H + 1) // 2) * ((W + 1) // 2) + tb - 1) // tb
    else:
        bm, bn, w8, k64, bpc = (tuple(hint3) + (0, 0))[:5]
        hint = ops.tile_hint(bm, bn, 1, w8) | (k64 << 27) | (bpc << 28)
        wop = wd
        fn_plain, fn_tail = L.lib.zsg_conv_igemm, L.lib.zsg_conv_igemm_bnstat
        chunks = (rows + bm - 1) // bm
    y0 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
    d0 = ops.fwd_desc(src, view_of(ops, y0, B, Ho, Wo, Co), cp, Co, k, s, p, 1, wC=cp, tile_hint=hint)
    part0 = torch.full((chunks, 2, Co), float("nan"), device="cuda")
    L.check(fn_plain(C.byref(d0), xd.data_ptr(), wop.data_ptr(), y0.data_ptr(), None, None, None, part0.data_ptr(), st), "conv + partial rows")
    nt = int(L.lib.zsg_conv_bn_tail_tickets(C.byref(d0), 1 if kern == "wino" else 0))
    assert nt == (Co + bn - 1) // bn and chunks <= 128
    # torch reference
    yr = F.conv2d(x.double(), w.double(), None, s, p).permute(0, 2, 3, 1).reshape(-1, Co)
    m_ref, v_ref = yr.mean(0), yr.var(0, unbiased=False)
    tickets = torch.zeros(nt, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, device="cuda")
    first = None
    for rep in range(5):
        y1 = torch.full((B, Ho, Wo, Co), float("nan"), device="cuda")
        d1 = ops.fwd_desc(src, view_of(ops, y1, B, Ho, Wo, Co), cp, Co, k, s, p, 1, wC=cp, tile_hint=hint)
        part1 = torch.full((chunks, 2, Co), float("nan"), device="cuda")
        mean, invstd = torch.full((Co,), float("nan"), device="cuda"), torch.full((Co,), float("nan"), device="cuda")
        rm, rv = torch.full((Co,), 0.25, device="cuda"), torch.full((Co,), 2.0, device="cuda")
        torch.cuda.synchronize()
        if rep >= 2:
            L.check(L.lib.zsg_memset_f32(noise.data_ptr(), noise.numel(), float(rep), C.c_void_p(side.cuda_stream)), "noise")
        L.check(fn_tail(C.byref(d1), xd.data_ptr(), wop.data_ptr(), y1.data_ptr(), part1.data_ptr(), tickets.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                        rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, st), "conv + in-kernel BatchNorm statistics")
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0, "tickets must be zero again after the launch"
        assert torch.equal(y1, y0) and torch.equal(part1, part0), "same stored values and partial rows as the plain fused launch"
        se, sse = part0[:, 0].double().sum(0), part0[:, 1].double().sum(0)
        m = se / rows
        var = (sse / rows - m * m).clamp_min(0)
        # (= the fp64 reduction of the partial rows rounded once; the host sums in another order, hence 1 ulp of slack)
        assert torch.allclose(mean.cpu(), m.float().cpu(), rtol=3e-7, atol=1e-9), "mean = fp64 reduction of the partial rows"
        assert torch.allclose(invstd.cpu(), (1.0 / torch.sqrt(var + 1e-5)).float().cpu(), rtol=1e-6)
        assert torch.allclose(rm.cpu(), (0.9 * 0.25 + 0.1 * m.float()).cpu(), rtol=1e-6, atol=1e-7)
        assert torch.allclose(rv.cpu(), (0.9 * 2.0 + 0.1 * (var * rows / (rows - 1)).float()).cpu(), rtol=1e-6, atol=1e-7)
        assert float((mean.double().cpu() - m_ref).abs().max()) <= 2e-5 * (float(yr.abs().max()) + 1)
        assert float((invstd.double().cpu() - 1 / torch.sqrt(v_ref + 1e-5)).abs().max()) <= 2e-4 * float((1 / torch.sqrt(v_ref + 1e-5)).max())
        cur = (mean.cpu(), invstd.cpu(), rm.cpu(), rv.cpu())
        if first is None:
            first = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, first)), "run-to-run bit-identical (fixed reduction order)"
    side.synchronize()


def test_tail_refused_where_it_cannot_apply(Z):
    """zsg_conv_bn_tail_tickets says -1 for the streaming 1x1 kernel, split-K, a missing hint and > 128 partial rows; the bnstat entry
    point then fails loudly (-1) instead of launching."""
    L, ops = Z
    B, Ci, Co, H, W = 4, 64, 64, 40, 40
    xd = torch.zeros(B, H, W, Ci, device="cuda")
    y = torch.zeros(B, H, W, Co, device="cuda")
    wd = torch.zeros(Co, 1, 1, Ci, device="cuda")
    src, out = view_of(ops, xd, B, H, W, Ci), view_of(ops, y, B, H, W, Co)
    for hint in (0, ops.tile_hint(32, 64, 1), ops.tile_hint(64, 64, 2), ops.tile_hint(64, 64, 1)):      # heuristic, pw, split-K, 100 rows ok
        d = ops.fwd_desc(src, out, Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=hint)
        n = int(L.lib.zsg_conv_bn_tail_tickets(C.byref(d), 0))
        assert (n == 1) == (hint == ops.tile_hint(64, 64, 1)), (hint, n)
    d = ops.fwd_desc(view_of(ops, torch.zeros(16, 38, 38, Ci, device="cuda"), 16, 38, 38, Ci), view_of(ops, torch.zeros(16, 38, 38, Co, device="cuda"), 16, 38, 38, Co),
                     Ci, Co, 1, 1, 0, 1, wC=Ci, tile_hint=ops.tile_hint(64, 64, 1))
    assert int(L.lib.zsg_conv_bn_tail_tickets(C.byref(d), 0)) == -1                  # 361 partial rows
    part = torch.zeros(361, 2, Co, device="cuda")
    tk = torch.zeros(4, dtype=torch.int32, device="cuda")
    m = torch.zeros(Co, device="cuda")
    x16, y16 = torch.zeros(16, 38, 38, Ci, device="cuda"), torch.zeros(16, 38, 38, Co, device="cuda")
    rc = L.lib.zsg_conv_igemm_bnstat(C.byref(d), x16.data_ptr(), wd.data_ptr(), y16.data_ptr(), part.data_ptr(), tk.data_ptr(), m.data_ptr(), m.data_ptr(),
                                     None, None, 0.1, 1e-5, L.stream_ptr())
    assert rc == -1 and b"128" in L.lib.zsg_last_error()
