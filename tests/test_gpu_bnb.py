"""BatchNorm-backward sums fused into the epilogue of the data gradient that completes dout (zsg_conv_igemm_bnb /
zsg_conv_wino_bnb + zsg_bn_backward_from_partials) against (a) fp64 sums computed on the host and (b) the unfused
zsg_conv_igemm + zsg_bn_backward pair on the same inputs.  Tolerances: the convolution output must be bit-identical to the
unfused launch (same kernel, same tile); sums rel 1e-4 of their scale (fp32 partial rows, different grouping); dx / dgamma /
dbeta rel 2e-4."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import Z, dev, nhwc, ohwi, pad4, view_of  # noqa: F401
from test_gpu_wino import make_u  # noqa: F401

pytestmark = pytest.mark.gpu

CASES = [
    # B, Cin (of the forward conv = channels of the BatchNorm), Cout, H, W, k, stride, accumulate, relu-mask, kernel, hint(bm,bn,w8) | (tb,bn,ps4)
    (2, 64, 256, 19, 19, 1, 1, False, True, "igemm", (64, 64, 0)),
    (2, 64, 256, 19, 19, 1, 1, True, True, "igemm", (128, 64, 0)),
    (3, 128, 64, 10, 13, 1, 1, True, False, "igemm", (128, 128, 1)),
    (2, 256, 128, 9, 9, 1, 1, False, True, "igemm", (64, 64, 1)),
    (2, 64, 64, 21, 21, 3, 2, False, True, "igemm", (64, 64, 0)),        # 3x3 stride 2: four parity classes, all with taps
    (2, 64, 64, 19, 19, 3, 1, False, True, "igemm", (128, 64, 1)),
    (2, 64, 64, 19, 19, 3, 1, False, True, "wino", (64, 64, 0)),
    (2, 128, 96, 10, 7, 3, 1, True, True, "wino", (32, 64, 1)),
    (3, 64, 128, 5, 5, 3, 1, False, False, "wino", (32, 32, 1)),
    # filter-resident streaming kernel (csrc/pw.hip, tile_hint BM = 32, BN = unit width): one partial row per workgroup
    (2, 256, 64, 19, 19, 1, 1, False, True, "igemm", (32, 128, 0)),
    (2, 256, 64, 19, 19, 1, 1, True, True, "igemm", (32, 64, 0)),
    (3, 64, 256, 10, 13, 1, 1, True, False, "igemm", (32, 32, 0)),
    (2, 64, 64, 21, 21, 1, 1, False, True, "igemm", (32, 64, 0)),
    (2, 512, 128, 19, 19, 1, 1, True, True, "igemm", (32, 128, 0)),        # four filter panels (column blocks)
    (16, 512, 128, 38, 38, 1, 1, False, True, "igemm", (32, 64, 0)),       # layer2 conv1's data gradient at the bench shape
    (16, 256, 64, 75, 75, 1, 1, True, True, "igemm", (32, 128, 0)),       # layer1 conv1's data gradient at the bench shape
    (16, 64, 256, 75, 75, 1, 1, False, True, "igemm", (32, 32, 0)),       # layer1 conv3's
    # many partial rows (100 / 200)
    (4, 64, 64, 40, 40, 1, 1, False, True, "igemm", (64, 64, 0)),
    (4, 64, 32, 40, 40, 3, 1, True, True, "wino", (32, 64, 1)),
    # stream-K launches (hint: bm, bn, w8, k64, workgroups per CU): sums and in-kernel finalize by the workgroups that finish the tiles
    (16, 256, 1024, 19, 19, 1, 1, True, True, "igemm", (64, 64, 1, 1, 2)),     # layer3 conv3's data gradient (K = 1024) at the bench shape
    (16, 512, 2048, 10, 10, 1, 1, False, True, "igemm", (128, 64, 1, 0, 1)),   # layer4 conv3's
    (2, 64, 256, 19, 19, 1, 1, True, False, "igemm", (128, 128, 1, 0, 1)),
    (16, 256, 256, 19, 19, 3, 1, False, True, "wino", (32, 64, 1, 1)),          # layer3 conv2's data gradient as a Winograd stream-K launch
    (4, 64, 96, 20, 20, 3, 1, True, True, "wino", (32, 64, 1, 1)),
]


@pytest.mark.parametrize("case", CASES, ids=[f"b{i}" for i in range(len(CASES))])
def test_dgrad_with_bn_backward_sums(Z, case):
    L, ops = Z
    B, Ci, Co, H, W, k, s, acc, use_mask, kern, hint3 = case
    g = torch.Generator().manual_seed(31 + Ci + Co + H)
    p = k // 2
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    dy = torch.randn(B, Co, Ho, Wo, generator=g)                       # gradient w.r.t. the convolution's output
    prev = torch.randn(B, H, W, Ci, generator=g) if acc else None      # what earlier consumers left in dout
    xbn = torch.randn(B, H, W, Ci, generator=g) * 2 + 0.5               # BatchNorm input
    mean, invstd = torch.randn(Ci, generator=g) * 0.3, torch.rand(Ci, generator=g) + 0.5
    gamma = torch.rand(Ci, generator=g) + 0.5
    bits = (torch.rand(B, H, W, Ci, generator=g) > 0.4) if use_mask else None
    st = L.stream_ptr()
    # reference dout in fp64
    dxr = torch.nn.grad.conv2d_input((B, Ci, H, W), w.double(), dy.double(), s, p).permute(0, 2, 3, 1)
    if acc:
        dxr = dxr + prev.double()
    gref = dxr * bits.double() if use_mask else dxr
    xhat = (xbn.double() - mean.double()) * invstd.double()
    s1_ref, s2_ref = gref.reshape(-1, Ci).sum(0), (gref * xhat).reshape(-1, Ci).sum(0)

    Cop = pad4(Co)
    dyd = dev(nhwc(dy, Cop))
    dyv = view_of(ops, dyd, B, Ho, Wo, Cop)
    wd = dev(ohwi(w))
    wt = torch.empty((Ci, k, k, Cop), device="cuda")
    L.check(L.lib.zsg_transpose_w(wd.data_ptr(), wt.data_ptr(), Co, k * k, Ci, Cop, st), "transpose_w")
    rows = B * H * W
    maskb = None
    if use_mask:
        bb = bits.reshape(-1, 4).to(torch.uint8)
        maskb = dev((bb[:, 0] | (bb[:, 1] << 1) | (bb[:, 2] << 2) | (bb[:, 3] << 3)).contiguous())
    xd, md, isd, gd = dev(xbn), dev(mean), dev(invstd), dev(gamma)
    if kern == "wino":
        tb, bn, ps4, skw = (tuple(hint3) + (0,))[:4]
        hint = tb | (bn << 8) | (1 << 16) | (ps4 << 24) | (skw << 28)
        wop = make_u(L, ops, wt, Ci, Cop, k * k * Cop, Cop, True)
        fn_plain, fn_bnb = L.lib.zsg_conv_wino, L.lib.zsg_conv_wino_bnb
        chunks = (B * ((H + 1) // 2) * ((W + 1) // 2) + tb - 1) // tb
    else:
        bm, bn, w8, k64, bpc = (tuple(hint3) + (0, 0))[:5]
        hint = ops.tile_hint(bm, bn, 1, w8) | (k64 << 27) | (bpc << 28)
        wop = wt
        fn_plain, fn_bnb = L.lib.zsg_conv_igemm, L.lib.zsg_conv_igemm_bnb

    def fresh():
        return dev(prev.clone()) if acc else torch.full((B, H, W, Ci), float("nan"), device="cuda")

    dx0 = fresh()
    d0 = ops.dgrad_desc(dyv, view_of(ops, dx0, B, H, W, Ci), Cop, Ci, k, s, p, 1, tile_hint=hint)
    assert not d0.zero_fill
    if kern != "wino":
        chunks = int(L.lib.zsg_conv_igemm_partial_rows(C.byref(d0)))
        if bm != 32:
            assert chunks == sum((B * d0.seg[i].rows_y * d0.seg[i].rows_x + bm - 1) // bm for i in range(d0.nseg))
        else:
            assert 0 < chunks <= 256 and (Ci < 512 or chunks <= 64)
    L.check(fn_plain(C.byref(d0), dyd.data_ptr(), wop.data_ptr(), dx0.data_ptr(), None, dx0.data_ptr() if acc else None, None, None, st), "plain dgrad")
    dx1 = fresh()
    d1 = ops.dgrad_desc(dyv, view_of(ops, dx1, B, H, W, Ci), Cop, Ci, k, s, p, 1, tile_hint=hint)
    part = torch.full((chunks, 2, Ci), float("nan"), device="cuda")
    L.check(fn_bnb(C.byref(d1), dyd.data_ptr(), wop.data_ptr(), dx1.data_ptr(), dx1.data_ptr() if acc else None, xd.data_ptr(), md.data_ptr(),
                   isd.data_ptr(), maskb.data_ptr() if use_mask else None, part.data_ptr(), st), "dgrad + bn-backward sums")
    assert torch.equal(dx0, dx1), "the fused launch must store exactly what the plain launch stores"
    # round 6: epi_flags bit 0 — the same launch STORES the ReLU-masked gradient (the residual branch's gradient); same partial rows
    if use_mask:
        dxm = fresh()
        dm = ops.dgrad_desc(dyv, view_of(ops, dxm, B, H, W, Ci), Cop, Ci, k, s, p, 1, tile_hint=hint)
        dm.epi_flags = 1
        partm = torch.full((chunks, 2, Ci), float("nan"), device="cuda")
        L.check(fn_bnb(C.byref(dm), dyd.data_ptr(), wop.data_ptr(), dxm.data_ptr(), dxm.data_ptr() if acc else None, xd.data_ptr(), md.data_ptr(),
                       isd.data_ptr(), maskb.data_ptr(), partm.data_ptr(), st), "dgrad + bn-backward sums, masked store")
        assert torch.equal(dxm.cpu(), dx1.cpu() * bits.float()), "epi_flags bit 0: stored dout = dout x ReLU bit, bit for bit"
        assert torch.equal(partm, part), "masked store: the partial rows are those of the unmasked launch"
    assert float((dx1.double().cpu() - dxr).abs().max()) < 2e-4 * float(dxr.abs().max())
    s1, s2 = part[:, 0].double().sum(0).cpu(), part[:, 1].double().sum(0).cpu()
    assert not torch.isnan(part).any()
    sc1 = float(gref.abs().reshape(-1, Ci).sum(0).max())
    sc2 = float((gref * xhat).abs().reshape(-1, Ci).sum(0).max())
    assert float((s1 - s1_ref).abs().max()) < 1e-4 * sc1, (float((s1 - s1_ref).abs().max()), sc1)
    assert float((s2 - s2_ref).abs().max()) < 1e-4 * sc2, (float((s2 - s2_ref).abs().max()), sc2)

    # finalize + apply from the partial rows  ==  the unfused zsg_bn_backward on the same dout
    ws = torch.empty(int(L.lib.zsg_bn_workspace_bytes(rows, Ci)) // 4 + 2 * Ci, device="cuda")
    outs = []
    for fused in (False, True):
        dxo, go = torch.empty(B, H, W, Ci, device="cuda"), torch.empty(B, H, W, Ci, device="cuda")
        dga, dbe = torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda")          # accumulate into existing values
        if fused:
            L.check(L.lib.zsg_bn_backward_from_partials(dx1.data_ptr(), maskb.data_ptr() if use_mask else None, xd.data_ptr(), rows, Ci, md.data_ptr(),
                                                        isd.data_ptr(), gd.data_ptr(), dxo.data_ptr(), go.data_ptr(), dga.data_ptr(), dbe.data_ptr(), 1,
                                                        part.data_ptr(), chunks, ws.data_ptr(), ws.numel() * 4, st), "bn_backward_from_partials")
        else:
            L.check(L.lib.zsg_bn_backward(dx0.data_ptr(), None, maskb.data_ptr() if use_mask else None, xd.data_ptr(), rows, Ci, md.data_ptr(),
                                          isd.data_ptr(), gd.data_ptr(), dxo.data_ptr(), go.data_ptr(), dga.data_ptr(), dbe.data_ptr(), 1,
                                          ws.data_ptr(), ws.numel() * 4, st), "bn_backward")
        outs.append((dxo.cpu(), go.cpu(), dga.cpu(), dbe.cpu()))
    for a, b_, what in zip(outs[1], outs[0], ("dx", "g_out", "dgamma", "dbeta")):
        assert torch.allclose(a, b_, rtol=2e-4, atol=2e-4 * float(b_.abs().max())), what
    assert torch.equal(outs[1][1], outs[0][1])
    # and against the closed form in fp64
    n = float(rows)
    dbeta_ref, dgamma_ref = s1_ref, s2_ref
    dx_ref = gamma.double() * invstd.double() * (gref - dbeta_ref / n - xhat * dgamma_ref / n)
    assert float((outs[1][0].double() - dx_ref).abs().max()) < 3e-4 * float(dx_ref.abs().max())
    assert float((outs[1][2].double() - 1 - dgamma_ref).abs().max()) < 1e-4 * sc2 + 1e-5

    # round 5: the same with the sums FINALISED inside the data-gradient launch by the last-arriving tile of each column block
    # (zsg_conv_*_bnb_tail + zsg_bn_bwd_apply, csrc/bn_tail.h) — where the launch has <= 128 partial rows per column block
    nt = int(L.lib.zsg_conv_bn_tail_tickets(C.byref(d1), 1 if kern == "wino" else 0))
    if kern != "wino" and hint3[0] == 32:
        assert nt == -1, "the streaming 1x1 kernel has no in-kernel finalize"
    if nt <= 0:
        assert chunks > 128 or (kern != "wino" and hint3[0] == 32)
        return
    fn_tail = L.lib.zsg_conv_wino_bnb_tail if kern == "wino" else L.lib.zsg_conv_igemm_bnb_tail
    tickets = torch.zeros(nt, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, device="cuda")
    for rep in range(4):
        dx2 = fresh()
        d2 = ops.dgrad_desc(dyv, view_of(ops, dx2, B, H, W, Ci), Cop, Ci, k, s, p, 1, tile_hint=hint)
        part2 = torch.full((chunks, 2, Ci), float("nan"), device="cuda")
        coef = torch.full((2, Ci), float("nan"), device="cuda")
        dga, dbe = torch.ones(Ci, device="cuda"), torch.ones(Ci, device="cuda")
        torch.cuda.synchronize()
        if rep >= 2:                       # uneven load: an HBM-bound kernel on another stream while the tiles arrive
            L.check(L.lib.zsg_memset_f32(noise.data_ptr(), noise.numel(), float(rep), C.c_void_p(side.cuda_stream)), "noise")
        L.check(fn_tail(C.byref(d2), dyd.data_ptr(), wop.data_ptr(), dx2.data_ptr(), dx2.data_ptr() if acc else None, xd.data_ptr(), md.data_ptr(),
                        isd.data_ptr(), maskb.data_ptr() if use_mask else None, part2.data_ptr(), tickets.data_ptr(), coef.data_ptr(),
                        dga.data_ptr(), dbe.data_ptr(), 1, st), "dgrad + bn-backward sums + in-kernel finalize")
        dxo, go = torch.empty(B, H, W, Ci, device="cuda"), torch.empty(B, H, W, Ci, device="cuda")
        L.check(L.lib.zsg_bn_bwd_apply(dx2.data_ptr(), maskb.data_ptr() if use_mask else None, xd.data_ptr(), rows, Ci, md.data_ptr(), isd.data_ptr(),
                                       gd.data_ptr(), coef.data_ptr(), dxo.data_ptr(), go.data_ptr(), st), "bn_bwd_apply")
        torch.cuda.synchronize()
        assert int(tickets.abs().sum()) == 0, "tickets must be zero again after the launch"
        assert torch.equal(dx2, dx1) and torch.equal(part2, part), "same stored values and partial rows as the launch without the finalize"
        c_ref = torch.stack([part[:, 0].double().sum(0), part[:, 1].double().sum(0)]) / n
        sc = float(c_ref.abs().max()) + 1e-30
        assert float((coef.double() - c_ref).abs().max()) <= 3e-7 * sc, "coefficients = fp64 sum of the partial rows, rounded once"
        assert torch.allclose(dga.cpu(), (1 + part[:, 1].double().sum(0).float()).cpu(), rtol=1e-6, atol=1e-6 * sc * n)
        assert torch.allclose(dbe.cpu(), (1 + part[:, 0].double().sum(0).float()).cpu(), rtol=1e-6, atol=1e-6 * sc * n)
        if rep == 0:
            first = (coef.clone(), dga.clone(), dbe.clone(), dxo.clone())
        else:
            assert all(torch.equal(a_, b_) for a_, b_ in zip((coef, dga, dbe, dxo), first)), "run-to-run bit-identical (fixed reduction order)"
        for a, b_, what in zip((dxo.cpu(), go.cpu()), outs[1][:2], ("dx", "g_out")):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5 * float(b_.abs().max())), what
    side.synchronize()
    assert float((outs[1][3].double() - 1 - dbeta_ref).abs().max()) < 1e-4 * sc1 + 1e-5


def test_batched_weight_gradients_in_the_plan(monkeypatch):
    """_Plan._batch_wgrads (round 6): with ZSG_WG_BATCH the Winograd weight gradients of a stage's identical bottlenecks are ONE launch
    at the last one's position — every gradient of the network as without batching (fp32 split-K order only), fewer launches, and the
    launch index each parameter's gradient is complete at (DDP buckets, Adam split) points at the batch.  Reference: autograd through
    fpn_resnet.py:86-100."""
    from oracle import zsg_oracle as O
    from zsgnet_pytorch_amd import config, loss, mdl
    from zsgnet_pytorch_amd._lib import lib
    monkeypatch.setenv("ZSG_WINO", "force")      # (every 3x3 / stride-1 convolution on the Winograd kernels whatever the tuner would time at this small size)
    # deterministic mode for both runs (no fp32-atomic split-K, fixed-order column sums): without it two runs of ONE plan already differ
    # by ~2e-3 on every gradient at this size (atomics reorder sums, ReLU / max-pool ties flip) and would hide what the batch changes
    monkeypatch.setenv("ZSG_DETERMINISTIC", "1")
    lib.zsg_set_deterministic(1)
    try:
        _batched_vs_not(monkeypatch, mdl, config, loss, O, lib)
    finally:
        lib.zsg_set_deterministic(0)


def _batched_vs_not(monkeypatch, mdl, config, loss, O, lib):
    grads, plans = [], []
    for on in (False, True):
        monkeypatch.setattr(mdl, "WG_BATCH", on)
        cfg = config.get_cfg(resnet_arch="resnet50", resize_img=[160, 160])
        net = mdl.get_default_net(9, cfg)
        net.load_state_dict(O.seeded_state_dict("resnet50", 5))
        net.to("cuda").train()
        r, s = config.ratios_scales(cfg)
        lf = loss.get_default_loss(r, s, cfg)
        bt = {k: v.cuda() for k, v in O.synthetic_batch(4, 160, 160, seed=9).items()}
        bt["h0"], bt["c0"] = torch.zeros(2, 4, 128), torch.zeros(2, 4, 128)
        lf(net(bt), bt)["loss"].mean().backward()
        torch.cuda.synchronize()
        grads.append(net.store.grad.clone())
        plans.append(next(iter(net._plans.values())))
    p0, p1 = plans
    nb = getattr(p1, "n_wgrad_batches", 0)
    n_w = sum(1 for c in p1.bwd.calls if c[0] is lib.zsg_conv_wgrad_wino_batched)
    print(f"batched plan: {nb} batches, {len(p0.bwd.calls)} -> {len(p1.bwd.calls)} backward launches")
    assert n_w >= 3 and n_w == nb and len(p1.bwd.calls) < len(p0.bwd.calls)
    ents = p1.net.store.entries
    worst = 0.0
    for n in p1.net._param_names:
        o, sz = ents[n].offset, ents[n].size
        a, b = grads[1][o:o + sz].double(), grads[0][o:o + sz].double()
        e_ = float((a - b).norm() / (b.norm() + 1e-30))
        if e_ > 2e-5:
            print(f"  {n}: rel {e_:.2e}")
        worst = max(worst, e_)
        i = p1.grad_ready.get(n, -1)
        assert -1 <= i <= len(p1.bwd.calls), (n, i)
    print(f"worst relative gradient difference batched vs not: {worst:.2e}")
    assert worst < 2e-5
    for n in ("backbone.encoder.layer3.1.conv2.weight", "backbone.encoder.layer3.5.conv2.weight"):
        assert p1.bwd.calls[p1.grad_ready[n]][0] is lib.zsg_conv_wgrad_wino_batched, n
