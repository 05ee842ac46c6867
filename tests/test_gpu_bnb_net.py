"""One training step of ResNet-50 FPN with the BatchNorm-backward sums fused into the data-gradient epilogues (default) and with
the separate partial pass (ZSG_BNB_FUSE=0): the same gradients (fp32 summation order aside), and the fused plan really contains
the fused launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_gpu_net as T  # noqa: E402
from test_gpu_net import Z  # noqa: E402,F401


def test_network_gradients_with_and_without_the_fusion(Z, monkeypatch):
    import zsgnet_pytorch_amd.mdl as M
    res = {}
    for fuse in (True, False):
        monkeypatch.setattr(M, "BNB_FUSE", fuse)
        cfg, net, sd, lf, ev = T.build(Z, arch="resnet50", seed=5)
        net.train()
        bt = T.O.synthetic_batch(4, 160, 160, seed=5)
        inp = T.to_dev(bt)
        torch.manual_seed(7)
        out = net(inp)
        lf(out, inp)["loss"].backward()
        torch.cuda.synchronize()
        nf = sum(1 for plan in net._plans.values() for fn, _, _ in plan.bwd.calls
                 if fn is M.lib.zsg_bn_backward_from_partials or fn is M.lib.zsg_bn_bwd_apply)      # (+ finalised in-kernel, round 5)
        res[fuse] = ({n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None}, nf)
    assert res[True][1] >= 25 and res[False][1] == 0, (res[True][1], res[False][1])
    worst, wname, wcos = 0.0, "", 1.0
    for n, ga in res[True][0].items():
        gb = res[False][0][n]
        e = float((ga - gb).norm() / (gb.norm() + 1e-12))
        wcos = min(wcos, float((ga * gb).sum() / (ga.norm() * gb.norm() + 1e-30)))
        if e > worst:
            worst, wname = e, n
    print("worst", wname, worst, "min cosine", wcos, "fused launches", res[True][1])
    # Same mathematics, different fp32 summation grouping AND different tile choices (the tuner sees a split-K penalty with the
    # fusion on, and re-times its candidates): at B=4 / 160^2 the 50 train-mode BatchNorm layers amplify such reorderings to
    # 1-3 % on individual tensors (ReLU decisions of near-zero activations flip), the same size as HIP-vs-oracle at this shape.
    # This is a consistency check — every gradient tensor points the same way and has the same size; exactness of the fused
    # launches is test_gpu_bnb.py's job, parity with the oracle test_gpu_net.py's / test_gpu_fullshape.py's (fusion on).
    assert wcos > 0.995, (wname, worst, wcos)
    assert worst < 1e-1, (wname, worst)
