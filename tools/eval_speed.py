"""Developer tool: eval-mode (validation) forward + loss + evaluator throughput, ResNet-50 FPN 300x300, B=16."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, evaluator, loss, mdl
from zsgnet_pytorch_amd.synth import synthetic_batch

cfg = config.get_cfg()
net = mdl.get_default_net(9, cfg).to("cuda").eval()
bt = {k: v.cuda() for k, v in synthetic_batch(16, 300, 300, seed=1).items()}
r, s = config.ratios_scales(cfg)
lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
with torch.no_grad():
    for _ in range(5):
        out = net(bt); lf(out, bt); ev(out, bt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        out = net(bt); lf(out, bt); ev(out, bt)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
print(f"eval: {1e3 * dt:.3f} ms per batch of 16 -> {16 / dt:.0f} img/s")
