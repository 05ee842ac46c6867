"""Fixtures made by the ORACLE (oracle/zsg_oracle.py), not by the reference — for the one configuration the reference cannot construct:
ResNet-101 + FPN (BASELINE.json configs[4]; the reference hard-codes resnet50, mdl.py:411).  The oracle's blocks are pinned against the
reference by the g8_* / g10_* goldens (tests/test_oracle_golden.py); this script only moves the oracle's CPU work — one fp32 and one fp64
forward + backward of ResNet-101 at 600x600, B=4: minutes on the host — out of the GPU suite, which then reads the numbers.

    python tests/golden/make_oracle_fixtures.py [o1] [o2]  # writes tests/golden/o1_r101_600_b4.npz, o2_r50_300_b16.npz

o1_r101_600_b4: seeds; the fp64 outputs (sampled) and loss; per parameter the fp64 gradient norm, the CPU-fp32 oracle's distance from it
(the yard-stick: how far a correct fp32 implementation is from fp64 at this depth), and SAMPLED gradient entries (<= 512 per parameter,
a fixed stride) in fp64 and in CPU fp32, so that the GPU test measures the HIP gradients' distance from fp64 on the same entries.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import zsg_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NS = 512          # sampled gradient entries per parameter


def sample_idx(n: int) -> np.ndarray:
    return np.arange(0, n, max(1, n // NS))[:NS]


def r101_600(B=4, seed=13, batch_seed=8, hc_seed=4):
    arch, hw = "resnet101", 600
    sd = O.seeded_state_dict(arch, seed)
    bt = O.synthetic_batch(B, hw, hw, seed=batch_seed)
    gq = torch.Generator().manual_seed(hc_seed)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    r, s = O.default_ratios_scales()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch=arch, six_hundred=True)
    fs = [tuple(x) for x in ref["feat_sizes"].tolist()]
    anc = torch.from_numpy(O.create_anchors(fs, r, s).astype(np.float32))
    l32 = O.torch_loss(ref, bt["annot"], anc)
    l32["loss"].backward()
    print(f"fp32 oracle: loss {float(l32['loss']):.6f}", flush=True)
    sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    bt64 = {k: v.double() for k, v in bt.items()}
    ref64 = O.zsgnet_forward(sd64, bt64, h0.double(), c0.double(), arch=arch, rank=O.sort_rank(bt["qlens"]), six_hundred=True)
    l64 = O.torch_loss(ref64, bt["annot"], anc)
    l64["loss"].backward()
    print(f"fp64 oracle: loss {float(l64['loss']):.9f}", flush=True)
    o64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
    o32 = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach()
    arrs = dict(seed=np.array([seed]), batch_seed=np.array([batch_seed]), hc_seed=np.array([hc_seed]), B=np.array([B]), hw=np.array([hw]),
                feat_sizes=np.array(fs), loss64=np.array([float(l64["loss"])]), loss32=np.array([float(l32["loss"])]),
                out64_s=o64[:, ::53].numpy(), out_stride=np.array([53]), fwd_err_cpu=np.array([float((o32.double() - o64).abs().max())]))
    names, n64, e32 = [], [], []
    g64s, g32s = [], []
    for n, v in sd64.items():
        if not (v.is_floating_point() and v.grad is not None):
            continue
        g64, g32 = v.grad.reshape(-1), sd[n].grad.reshape(-1).double()
        idx = sample_idx(g64.numel())
        names.append(n)
        n64.append(float(g64.norm()))
        e32.append(float((g32 - g64).norm()))
        a = np.zeros(NS)
        b = np.zeros(NS)
        a[:len(idx)] = g64.numpy()[idx]
        b[:len(idx)] = g32.numpy()[idx]
        g64s.append(a)
        g32s.append(b)
    arrs.update(names=np.array(names), norm64=np.array(n64), err32=np.array(e32), g64_s=np.stack(g64s), g32_s=np.stack(g32s).astype(np.float32))
    path = os.path.join(OUT, "o1_r101_600_b4.npz")
    np.savez_compressed(path, **arrs)
    print(f"o1_r101_600_b4.npz  {os.path.getsize(path) / 1024:.1f} KB, {len(names)} parameters", flush=True)


def r50_300_b16():
    """o2_r50_300_b16: the oracle side of tests/test_gpu_fullshape.py::test_configs1_b16_vs_reference_golden_and_oracle — the fp32 CPU
    oracle and its fp64 twin at the benchmark shape on the inputs of the REFERENCE golden g10_e2e_300_b16 (same seeds, same h0 / c0):
    the fp32 oracle's attention logits (all anchors: the arg-max tie analysis needs them), the fp64 outputs on every 7th anchor, the
    CPU oracle's forward distance from fp64, and per parameter the fp64 gradient norm, the CPU-fp32 distance from it and sampled entries
    of both gradients.  ~40 s of host work that used to run inside the GPU suite."""
    g = np.load(os.path.join(OUT, "g10_e2e_300_b16.npz"))
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    bt = O.synthetic_batch(16, 300, 300, seed=int(g["batch_seed"][0]))
    h0, c0 = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    for k, v in sd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_()
    r, s = O.default_ratios_scales()
    ref = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet50")
    anc = torch.from_numpy(O.create_anchors([tuple(x) for x in ref["feat_sizes"].tolist()], r, s).astype(np.float32))
    l32 = O.torch_loss(ref, bt["annot"], anc)
    l32["loss"].backward()
    sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    bt64 = {k: v.double() for k, v in bt.items()}
    ref64 = O.zsgnet_forward(sd64, bt64, h0.double(), c0.double(), arch="resnet50", rank=O.sort_rank(bt["qlens"]))
    l64 = O.torch_loss(ref64, bt["annot"], anc)
    l64["loss"].backward()
    o64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
    o32 = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach()
    arrs = dict(att32=ref["att_out"].detach().squeeze(-1).numpy(), out64_s=o64[:, ::7].numpy(), out_stride=np.array([7]),
                fwd_err_cpu=np.array([float((o32.double() - o64).abs().max())]), loss32=np.array([float(l32["loss"])]), loss64=np.array([float(l64["loss"])]))
    names, n64, e32, n32, g64s, g32s = [], [], [], [], [], []
    for n, v in sd64.items():
        if not (v.is_floating_point() and v.grad is not None):
            continue
        g64_, g32_ = v.grad.reshape(-1), sd[n].grad.reshape(-1).double()
        idx = sample_idx(g64_.numel())
        names.append(n)
        n64.append(float(g64_.norm()))
        n32.append(float(g32_.norm()))
        e32.append(float((g32_ - g64_).norm()))
        a, b = np.zeros(NS), np.zeros(NS)
        a[:len(idx)] = g64_.numpy()[idx]
        b[:len(idx)] = g32_.numpy()[idx]
        g64s.append(a)
        g32s.append(b)
    arrs.update(names=np.array(names), norm64=np.array(n64), norm32=np.array(n32), err32=np.array(e32), g64_s=np.stack(g64s), g32_s=np.stack(g32s).astype(np.float32))
    path = os.path.join(OUT, "o2_r50_300_b16.npz")
    np.savez_compressed(path, **arrs)
    print(f"o2_r50_300_b16.npz  {os.path.getsize(path) / 1024:.1f} KB, {len(names)} parameters; fp32 loss {float(l32['loss']):.6f} fp64 {float(l64['loss']):.9f}", flush=True)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = [a for a in sys.argv[1:]]
    if not which or "o1" in which:
        r101_600()
    if not which or "o2" in which:
        r50_300_b16()
