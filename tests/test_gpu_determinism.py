"""ZSG_DETERMINISTIC=1: two PROCESSES that share the tuner's tile choices (ZSG_TUNE_CACHE) produce bit-identical
outputs, losses and gradients — no launch combines partial sums with fp32 atomics in this mode (split-K candidates are
not offered, bias / border column sums use one block per element group, the weight gradient's slabs are summed in a
fixed order)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from oracle import zsg_oracle as O
from zsgnet_pytorch_amd import config, loss, mdl, optim
cfg = config.get_cfg(resnet_arch="resnet50")
net = mdl.get_default_net(9, cfg)
net.load_state_dict(O.seeded_state_dict("resnet50", 3))
net.to("cuda").train()
r, s = config.ratios_scales(cfg)
lf = loss.get_default_loss(r, s, cfg)
opt = optim.FusedAdam(net, lr=1e-4)
bt = {{k: v.cuda() for k, v in O.synthetic_batch(3, 160, 128, seed=9).items()}}
bt["h0"], bt["c0"] = torch.zeros(2, 3, 128), torch.zeros(2, 3, 128)
res = {{}}
for step in range(4):
    opt.zero_grad()
    out = net(bt)
    ls = lf(out, bt)
    ls["loss"].backward()
    if step < 2:                  # (the last two steps run without any host synchronisation between backward and the optimizer:
        torch.cuda.synchronize()  #  with ZSG_ADAM_OVERLAP=1 FusedAdam updates most parameters UNDER the backward's last weight gradients)
        res[f"out{{step}}"] = out["att_bbx_out"].detach().cpu()
        res[f"loss{{step}}"] = ls["loss"].detach().cpu()
        res[f"grad{{step}}"] = net.store.grad.clone().cpu()
    opt.step()
res["w"] = net.store.flat.clone().cpu()
torch.save(res, sys.argv[1])
"""


def test_two_processes_bit_identical(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "run.py"
    script.write_text(SCRIPT.format(root=ROOT))
    env = dict(os.environ, ZSG_DETERMINISTIC="1", ZSG_TUNE_CACHE=str(tmp_path / "tune.json"))
    outs = []
    for i in range(3):            # the third process: the optimizer update split around the backward's tail (ZSG_ADAM_OVERLAP=1) — same bits
        out = tmp_path / f"r{i}.pt"
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=dict(env, ZSG_ADAM_OVERLAP="1") if i == 2 else env, timeout=900)
        outs.append(torch.load(out))
    assert (tmp_path / "tune.json").exists(), "the first process must persist its tile choices"
    a, b, c = outs
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k} differs between two deterministic runs (max |d| {float((a[k] - b[k]).abs().max()):.3g})"
        assert torch.equal(a[k], c[k]), f"{k}: overlapped optimizer update differs from the joined one (max |d| {float((a[k] - c[k]).abs().max()):.3g})"
    assert torch.isfinite(a["grad1"]).all() and float(a["grad1"].abs().sum()) > 0
