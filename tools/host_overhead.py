"""Developer tool: how far ahead of the GPU is the host?  Times the enqueue of one training step (no synchronisation)
against the step's wall clock.  usage (GPU box): python tools/host_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd.synth import synthetic_batch
from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim


def main():
    cfg = config.get_cfg()
    net = mdl.get_default_net(9, cfg).to("cuda").train()
    bt = {k: v.cuda() for k, v in synthetic_batch(16, 300, 300, seed=1).items()}
    r, s = config.ratios_scales(cfg)
    lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
    opt = optim.FusedAdam(net, lr=1e-4, betas=(0.9, 0.99))

    def step(parts=None):
        t = [time.perf_counter()]
        opt.zero_grad(); t.append(time.perf_counter())
        out = net(bt); t.append(time.perf_counter())
        ls = lf(out, bt); t.append(time.perf_counter())
        ls["loss"].mean().backward(); t.append(time.perf_counter())
        opt.step(); t.append(time.perf_counter())
        ev(out, bt); t.append(time.perf_counter())
        if parts is not None:
            parts.append([b - a for a, b in zip(t, t[1:])])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    parts = []
    t0 = time.perf_counter()
    for _ in range(20):
        step(parts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    names = ["zero_grad", "forward", "loss", "backward", "adam", "eval"]
    avg = [sum(p[i] for p in parts) / len(parts) * 1e3 for i in range(len(names))]
    print(f"host enqueue {1e3 * (t1 - t0) / 20:.2f} ms/step, wall {1e3 * (t2 - t0) / 20:.2f} ms/step")
    print("host ms per part: " + ", ".join(f"{n} {a:.2f}" for n, a in zip(names, avg)))


if __name__ == "__main__":
    main()
