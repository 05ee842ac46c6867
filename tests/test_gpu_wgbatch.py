"""Batched split-K slab reduction of the weight gradients (zsg_conv_wgrad_partial / zsg_conv_wgrad_wino_partial +
zsg_wgrad_reduce_job / zsg_wgrad_reduce_batched): several layers' partial tiles summed by ONE launch must give, bit for bit,
what each layer's own zsg_conv_wgrad / zsg_conv_wgrad_wino (kernel + its private reduce launch) gives — same kernels, same
fixed summation order — including accumulation into a pre-filled gradient and a channel window of a wider weight."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_ops import dev, nhwc, pad4, view_of  # noqa: E402


@pytest.fixture(scope="module")
def Z():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import zsgnet_pytorch_amd._lib as L
    import zsgnet_pytorch_amd.ops as ops
    return L, ops


LAYERS = [
    # B, Ci, Co, H, W, k, s, p, splits, wino
    (4, 64, 64, 30, 30, 1, 1, 0, 24, False),
    (4, 64, 64, 30, 30, 3, 1, 1, 12, True),
    (2, 128, 256, 19, 19, 1, 1, 0, 6, False),
    (2, 256, 128, 10, 10, 3, 1, 1, 3, True),
    (2, 64, 128, 21, 21, 3, 2, 1, 5, False),
    (2, 256, 64, 12, 12, 1, 1, 0, 1, False),       # unsplit: written directly, no job
]


def test_batched_reduce_equals_per_layer_reduce(Z):
    L, ops = Z
    lib = L.lib
    st = L.stream_ptr()
    g = torch.Generator().manual_seed(5)
    jb = lib.zsg_wgrad_reduce_job_bytes()
    host = (C.c_char * (jb * len(LAYERS)))()
    keep, refs, outs = [], [], []
    blk, njobs = 0, 0
    for (B, Ci, Co, H, W, k, s, p, splits, wino) in LAYERS:
        Ho, Wo = ops.conv_out(H, k, s, p), ops.conv_out(W, k, s, p)
        x = dev(nhwc(torch.randn(B, Ci, H, W, generator=g)))
        dy = dev(nhwc(torch.randn(B, Co, Ho, Wo, generator=g)))
        cp = pad4(Ci)
        src, dyv = view_of(ops, x, B, H, W, cp), view_of(ops, dy, B, Ho, Wo, Co)
        d = ops.fwd_desc(src, dyv, cp, Co, k, s, p, 1, wC=cp, tile_hint=ops.tile_hint(64, 64, splits))
        pre = torch.randn(Co, k, k, cp, generator=g).cuda()              # accumulate into a pre-filled gradient
        fn_full = lib.zsg_conv_wgrad_wino if wino else lib.zsg_conv_wgrad
        fn_part = lib.zsg_conv_wgrad_wino_partial if wino else lib.zsg_conv_wgrad_partial
        need = int((lib.zsg_conv_wgrad_wino_workspace_bytes if wino else lib.zsg_conv_wgrad_workspace_bytes)(C.byref(d)))
        ws1, ws2 = torch.empty(max(need // 4, 4), device="cuda"), torch.empty(max(need // 4, 4), device="cuda")
        ref = pre.clone()
        L.check(fn_full(C.byref(d), x.data_ptr(), dy.data_ptr(), ref.data_ptr(), 1, ws1.data_ptr(), need, st), "full")
        out = pre.clone()
        ns = C.c_int32(0)
        L.check(fn_part(C.byref(d), x.data_ptr(), dy.data_ptr(), out.data_ptr(), 1, ws2.data_ptr(), need, C.addressof(ns), st), "partial")
        assert (ns.value > 1) == (splits > 1), (ns.value, splits)
        if ns.value > 1:
            torch.cuda.synchronize()
            assert torch.equal(out, pre), "a split partial launch must not touch dw"
            nb = lib.zsg_wgrad_reduce_job(C.byref(d), ws2.data_ptr(), out.data_ptr(), 1, ns.value, blk, C.addressof(host) + njobs * jb)
            assert nb > 0
            blk += nb
            njobs += 1
        keep += [x, dy, ws1, ws2, d]
        refs.append(ref)
        outs.append(out)
    jobs_dev = torch.frombuffer(bytearray(bytes(host)[:njobs * jb]), dtype=torch.uint8).cuda()
    L.check(lib.zsg_wgrad_reduce_batched(jobs_dev.data_ptr(), njobs, blk, 0.0, st), "reduce_batched")
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(outs, refs)):
        assert torch.equal(a, b), f"layer {i}: batched reduction differs from the per-layer reduction (max {float((a - b).abs().max())})"
    assert lib.zsg_wgrad_reduce_job(C.byref(keep[4]), keep[3].data_ptr(), outs[0].data_ptr(), 1, 1, 0, C.addressof(host)) == -1       # nothing to reduce


def test_workspace_query_covers_the_heuristic_launch(Z):
    """tile_hint = 0: the launch picks its own split count; a workspace sized by the query must always be enough"""
    L, ops = Z
    lib = L.lib
    st = L.stream_ptr()
    for (B, Ci, Co, H) in ((2, 64, 64, 30), (8, 32, 32, 40), (2, 256, 256, 10)):
        x = torch.randn(B, H, H, Ci, device="cuda")
        dy = torch.randn(B, H, H, Co, device="cuda")
        d = ops.fwd_desc(view_of(ops, x, B, H, H, Ci), view_of(ops, dy, B, H, H, Co), Ci, Co, 3, 1, 1, 1, wC=Ci)
        dw = torch.zeros(Co, 3, 3, Ci, device="cuda")
        need = int(lib.zsg_conv_wgrad_wino_workspace_bytes(C.byref(d)))
        ws = torch.empty(max(need // 4, 4), device="cuda")
        L.check(lib.zsg_conv_wgrad_wino(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), need, st), "wgrad_wino, heuristic splits")
        ref = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2).cpu(), (Co, Ci, 3, 3), dy.permute(0, 3, 1, 2).cpu(), padding=1)
        err = float((dw.permute(0, 3, 1, 2).cpu() - ref).abs().max())
        assert err <= 1e-3 * float(ref.abs().max()), err
