"""Developer tool: per-launch forward times of the encoder with the BatchNorm applied by the consumer convolution's loader
(ZSG_BN_CONSUMER_FUSE=1) next to the separate apply launches (=0), same process, same weights.
usage (GPU box): python tools/prefuse_ab.py [arch] [B] [img]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd.synth import synthetic_batch
from zsgnet_pytorch_amd import config, mdl
from zsgnet_pytorch_amd._lib import stream_ptr


def run(fuse, arch, B, img):
    os.environ["ZSG_BN_CONSUMER_FUSE"] = fuse
    cfg = config.get_cfg(resnet_arch=arch)
    torch.manual_seed(0)
    net = mdl.get_default_net(9, cfg).to("cuda").train()
    bt = {k: v.cuda() for k, v in synthetic_batch(B, img, img, seed=1).items()}
    for _ in range(3):
        net(bt)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    evs[0].record()
    for i in range(30):
        net(bt)
        evs[i + 1].record()
    torch.cuda.synchronize()
    wall = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(30))[15]
    plan = list(net._plans.values())[0]
    acc = {}
    for _ in range(5):
        for (what, fname, ms) in plan.fwd.profile(stream_ptr()):
            acc.setdefault(what, [fname, []])[1].append(ms)
    return wall, {k: (v[0], sorted(v[1])[len(v[1]) // 2]) for k, v in acc.items()}


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    img = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    w0, p0 = run("0", arch, B, img)
    w1, p1 = run("1", arch, B, img)
    print(f"forward wall (grad on, median of 30): separate {w0:.3f} ms, fused {w1:.3f} ms")
    print(f"sum of launches: separate {sum(v[1] for v in p0.values()):.3f} ms, fused {sum(v[1] for v in p1.values()):.3f} ms")
    keys = [k for k in p0 if "layer" in k]
    tot = [0, 0]
    print(f"{'launch':60s} separate   fused")
    for k in keys:
        a = p0[k][1]
        kk = k + "+pre" if (k + "+pre") in p1 else k
        b = p1.get(kk, p1.get("apply:" + k, (None, float('nan'))))[1]
        if k in p1 or kk in p1 or ("apply:" + k) in p1:
            tag = "(pre)" if kk != k else ("(side)" if ("apply:" + k) in p1 and k not in p1 else "")
            print(f"{k:60s} {a * 1e3:7.1f}  {b * 1e3:7.1f} {tag}")
        else:
            print(f"{k:60s} {a * 1e3:7.1f}     -")
    for k in p1:
        if k.startswith("affine:"):
            print(f"{k:60s}     -    {p1[k][1] * 1e3:7.1f}")


if __name__ == "__main__":
    main()
