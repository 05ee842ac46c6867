"""The autotuner's candidates must all compute the SAME thing (round 6): a tile / split-K / stream-K / Winograd variant that is fast and
slightly wrong would be picked on timing alone, and only on the boxes where it happens to win.  ZSG_TUNE_VERIFY=1 (ops._verify_candidates)
runs every candidate of every launch shape once from the same state of the output buffer and compares the stored outputs with the first
candidate's (fp32 summation-order tolerance).  Reference counterpart: none — cuDNN's algorithm choice behind nn.Conv2d (mdl.py:211-219,
fpn_resnet.py:86-100) is trusted the same way; this test is what makes that trust checkable here."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import zsg_oracle as O  # noqa: E402
from test_gpu_net import Z, build, to_dev  # noqa: E402,F401


@pytest.mark.parametrize("arch,hw,bs", [("resnet18", (112, 80), 3), ("resnet50", (144, 176), 5)])
def test_every_tuner_candidate_agrees_with_its_siblings(Z, monkeypatch, arch, hw, bs):
    """an odd geometry (no other test lowers it: every launch shape is tuned here, with the cross-check on): forward + backward, i.e. the
    forward / data-gradient candidates (implicit GEMM tiles, 8-wave and 64-deep-K forms, the streaming 1x1 kernel, stream-K, atomic
    split-K, Winograd tiles) and the weight-gradient candidates (direct tiles x split-K, Winograd F(3x3,2x2) x split-K x block order)"""
    from zsgnet_pytorch_amd import ops
    monkeypatch.setenv("ZSG_TUNE_VERIFY", "1")
    n0, b0, t0 = ops.TUNE_INFO.get("verify_n", 0), ops.TUNE_INFO.get("verify_bad", 0), ops.TUNE_INFO["tuned_now"]
    cfg, net, sd, lf, ev = build(Z, arch=arch, seed=11, resize_img=list(hw))
    net.train()
    bt = O.synthetic_batch(bs, hw[0], hw[1], seed=31)
    inp = to_dev(bt)
    out = net(inp)
    lf(out, inp)["loss"].mean().backward()
    torch.cuda.synchronize()
    tuned, n, bad = ops.TUNE_INFO["tuned_now"] - t0, ops.TUNE_INFO.get("verify_n", 0) - n0, ops.TUNE_INFO.get("verify_bad", 0) - b0
    print(f"{arch} {hw} B={bs}: {tuned} launch shapes tuned, {n} candidate outputs compared with their first sibling's, {bad} outliers")
    assert tuned >= 30 and n >= 300, (tuned, n)
    assert bad == 0, f"{bad} tuner candidates disagree with their siblings (see the [zsg tune-verify] lines above)"


def test_the_cross_check_is_live(Z, monkeypatch):
    """negative control: with the tolerance at 1e-9 the check must FIND the candidates' summation-order differences (different tiles,
    split-K, Winograd round differently) — a comparison that could not fail would prove nothing"""
    from zsgnet_pytorch_amd import ops
    monkeypatch.setenv("ZSG_TUNE_VERIFY", "1")
    monkeypatch.setenv("ZSG_TUNE_VERIFY_TOL", "1e-9")
    n0, b0 = ops.TUNE_INFO.get("verify_n", 0), ops.TUNE_INFO.get("verify_bad", 0)
    try:
        cfg, net, sd, lf, ev = build(Z, arch="resnet18", seed=12, resize_img=[80, 112])
        net.train()
        inp = to_dev(O.synthetic_batch(3, 80, 112, seed=32))
        lf(net(inp), inp)["loss"].mean().backward()
        torch.cuda.synchronize()
        n, bad = ops.TUNE_INFO.get("verify_n", 0) - n0, ops.TUNE_INFO.get("verify_bad", 0) - b0
        print(f"tolerance 1e-9: {bad} of {n} candidate outputs differ from their first sibling's in the last bits")
        assert n >= 300 and bad >= 20, (n, bad)
    finally:
        ops.TUNE_INFO["verify_bad"] = b0          # (the negative control's findings are not outliers)
