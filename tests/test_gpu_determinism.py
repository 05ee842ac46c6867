"""ZSG_DETERMINISTIC=1: two PROCESSES that share the tuner's tile choices (ZSG_TUNE_CACHE) produce bit-identical
outputs, losses and gradients — no launch combines partial sums with fp32 atomics in this mode (split-K candidates are
not offered, bias / border column sums use one block per element group, the weight gradient's slabs are summed in a
fixed order) — and so do the same step on one stream and under the other scheduling options (round 3)."""
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
    # process 2: the optimizer update split around the backward's tail (ZSG_ADAM_OVERLAP=1); 3: every launch on ONE stream (no
    # cross-stream edge can be missing there); 4: round-2 scheduling (event-record markers instead of completion signals, the
    # backward's weight images / the query encoder released at the head of the forward, P3 before the P6 chain).  Scheduling decides
    # WHEN a launch runs, never what it computes: all five must agree to the bit.
    variants = [{}, {}, {"ZSG_ADAM_OVERLAP": "1"}, {"ZSG_SIDE_STREAM": "0"},
                {"ZSG_COMPLETION_EVENTS": "0", "ZSG_PREP_AT": "top", "ZSG_LANG_AT": "head", "ZSG_FPN_P6_FIRST": "0", "ZSG_PREP_RELEASE_TOP": "0"}]
    outs = []
    for i, extra in enumerate(variants):
        out = tmp_path / f"r{i}.pt"
        subprocess.run([sys.executable, str(script), str(out)], check=True, env=dict(env, **extra), timeout=900)
        outs.append(torch.load(out))
    assert (tmp_path / "tune.json").exists(), "the first process must persist its tile choices"
    a = outs[0]
    what = ["a second deterministic run", "the overlapped optimizer update", "the single-stream run", "the round-2 scheduling"]
    for o, w in zip(outs[1:], what):
        for k in a:
            assert torch.equal(a[k], o[k]), f"{k}: {w} differs (max |d| {float((a[k] - o[k]).abs().max()):.3g})"
    assert torch.isfinite(a["grad1"]).all() and float(a["grad1"].abs().sum()) > 0
