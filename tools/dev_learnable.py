"""Developer tool (GPU box): the learnable-task training of tests/test_gpu_fullshape.py::test_learnable_task_reaches_the_same_accuracy as
a script that KEEPS the loss curve — for hunting an intermittent bad step.  Prints the smoothed end loss, the held-out hits, the first
step whose loss deviates from the reference's curve by more than 3x (after step 30), and with --check every step's output / gradient
NaN-Inf state.  usage: python tools/dev_learnable.py [--steps N] [--noeval]   (environment switches ZSG_* pass through)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import zsg_oracle as O
from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim

RATIOS, SCALES = O.default_ratios_scales()
g = np.load(os.path.join(ROOT, "tests", "golden", "g15_learnable.npz"), allow_pickle=False)
S, B, steps, lr_ = int(g["S"][0]), int(g["B"][0]), int(g["steps"][0]), float(g["lr"][0])
if "--steps" in sys.argv:
    steps = int(sys.argv[sys.argv.index("--steps") + 1])
if "--lr" in sys.argv:
    lr_ = float(sys.argv[sys.argv.index("--lr") + 1])
decay_at = int(sys.argv[sys.argv.index("--decay-at") + 1]) if "--decay-at" in sys.argv else int(g["decay_at"][0])
ref_losses = np.concatenate([g["losses"], np.full(max(0, steps - len(g["losses"])), g["losses"][-1])])
cfg = config.get_cfg(resnet_arch="resnet50", resize_img=[S, S])
net = mdl.get_default_net(9, cfg)
net.load_state_dict(O.seeded_state_dict("resnet50", int(g["seed"][0])))
net.to("cuda").train()
lf, ev = loss.get_default_loss(RATIOS, SCALES, cfg), evaluator.get_default_eval(RATIOS, SCALES, cfg)
opt = optim.FusedAdam(net, lr=lr_, betas=(0.9, 0.99))
gq = torch.Generator().manual_seed(8)
hip = []
gn = []
for it in range(steps):
    if it == decay_at:
        for grp in opt.param_groups:
            grp["lr"] = lr_ * 0.1
    bt = O.learnable_batch(B, S, seed=100 + it)
    h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
    inp = {k: v.cuda() for k, v in bt.items()}
    inp["h0"], inp["c0"] = h0, c0
    opt.zero_grad()
    out = net(inp)
    ls = lf(out, inp)
    ls["loss"].mean().backward()
    if "--gnorm" in sys.argv:
        gn.append(float(net.store.grad.double().norm()))
    opt.step()
    hip.append(float(ls["loss"].detach()))
last = None
for cur in zip(hip, ref_losses):
    last = cur if last is None else (0.8 * last[0] + 0.2 * cur[0], 0.8 * last[1] + 0.2 * cur[1])
dev = [i for i in range(30, steps) if hip[i] > 3 * max(ref_losses[max(0, i - 5):i + 6]) or not np.isfinite(hip[i])]
hits = -1
if "--noeval" not in sys.argv:
    net.eval()
    hits = 0.0
    with torch.no_grad():
        for bi in range(16):
            bt = O.learnable_batch(16, S, seed=9000 + bi)
            h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
            inp = {k: v.cuda() for k, v in bt.items()}
            inp["h0"], inp["c0"] = h0, c0
            hits += float(ev(net(inp), inp)["Acc"]) * 16
tail = hip[60:]
print(f"lr {lr_:g} mean(last 10) {np.mean(hip[-10:]):.3f} max(step>=60) {max(tail) if tail else 0:.2f} median(60..) {np.median(tail) if tail else 0:.3f} | "
      f"end {last[0]:.3f} (ref {last[1]:.3f}) hits {hits:.0f} first-deviating-steps {dev[:6]} "
      + (" ".join(f"{i}:{hip[i]:.2f}/{ref_losses[i]:.2f}" for i in dev[:3])) + (f" max gnorm {max(gn):.3e} at {int(np.argmax(gn))}" if gn else ""), flush=True)
if "--save" in sys.argv:
    np.save(sys.argv[sys.argv.index("--save") + 1], np.array(hip, dtype=np.float64))
if "--curve" in sys.argv:
    print(" ".join(f"{v:.2f}" for v in hip))
