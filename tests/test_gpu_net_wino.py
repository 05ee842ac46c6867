"""The network-level parity tests of test_gpu_net.py with every eligible convolution FORCED onto the Winograd
F(2x2,3x3) kernel (ZSG_WINO=force; by default the autotuner picks per layer, which on the tiny test shapes may or may
not select it).  Same goldens, same tolerances: the Winograd path must meet the direct kernel's bounds."""
import pytest

pytestmark = pytest.mark.gpu

import test_gpu_net as T  # noqa: E402
from test_gpu_net import Z  # noqa: E402,F401


@pytest.fixture(autouse=True)
def _force_wino(monkeypatch):
    monkeypatch.setenv("ZSG_WINO", "force")


def _count_wino(net):
    from zsgnet_pytorch_amd._lib import lib
    n = 0
    for plan in net._plans.values():
        for prog in (plan.fwd, plan.bwd):
            n += sum(1 for fn, _, _ in prog.calls if fn is lib.zsg_conv_wino)
    return n


@pytest.mark.parametrize("tag,hw", [("e2e_128", 128), ("e2e_300", 300)])
def test_e2e_golden_winograd(Z, gold, tag, hw):
    T.test_forward_backward_vs_reference_golden(Z, gold, tag, hw)


def test_winograd_is_really_used(Z):
    import torch
    cfg, net, sd, lf, ev = T.build(Z, arch="resnet18", seed=3)
    net.train()
    bt = T.O.synthetic_batch(2, 96, 96, seed=3)
    inp = T.to_dev(bt)
    out = net(inp)
    lf(out, inp)["loss"].backward()
    torch.cuda.synchronize()
    # resnet18 @96: 7 stride-1 3x3 BasicBlock convs... + FPN P3_2/P4_2/P5_2 + 6 head convs, forward and data gradient
    assert _count_wino(net) >= 30, _count_wino(net)


@pytest.mark.parametrize("arch,B,hw", [("resnet18", 2, 96), ("resnet50", 3, 160)])
def test_vs_fp64_oracle_winograd(Z, arch, B, hw):
    T.test_forward_backward_vs_oracle(Z, arch, B, hw)


def test_ssd_vgg_winograd(Z, gold):
    T.test_ssd_vgg_backbone_vs_golden_and_fp64(Z, gold)


def test_eval_mode_winograd(Z):
    T.test_eval_mode_and_state_dict_roundtrip(Z)


def test_train_steps_winograd(Z):
    T.test_train_steps_match_oracle_and_reduce_loss(Z)
