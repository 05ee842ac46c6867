"""Developer tool: GPU wall time of the phases of one training step (events on the main stream)."""
import os
import sys

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
    names = ["zero_grad", "forward", "loss", "backward", "adam", "eval"]
    acc = [0.0] * len(names)
    n = 0
    for it in range(25):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        e[0].record(); opt.zero_grad()
        e[1].record(); out = net(bt)
        e[2].record(); ls = lf(out, bt)
        e[3].record(); ls["loss"].mean().backward()
        e[4].record(); opt.step()
        e[5].record(); ev(out, bt)
        e[6].record()
        torch.cuda.synchronize()
        if it >= 5:
            n += 1
            for i in range(len(names)):
                acc[i] += e[i].elapsed_time(e[i + 1])
    print("GPU ms per phase: " + ", ".join(f"{k} {v / n:.3f}" for k, v in zip(names, acc)) + f" | total {sum(acc) / n:.3f}")


if __name__ == "__main__":
    main()
