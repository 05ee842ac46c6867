"""Developer tool: the lowered forward (or backward) program of the bench configuration IN PROGRAM ORDER with the lane and the
event-timed duration of every launch (each launch alone: events serialise), so that the dependent chain can be read off.
usage (GPU box): python tools/fwd_listing.py [fwd|bwd|prep|prep_u ...] > gpurun_out/fwd_listing.txt"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim            # noqa: E402
from zsgnet_pytorch_amd._lib import stream_ptr                                # noqa: E402
from zsgnet_pytorch_amd.synth import synthetic_batch                         # noqa: E402

cfg = config.get_cfg(resnet_arch="resnet50", bs=16, resize_img=[300, 300], mdl_to_use="retina")
torch.manual_seed(1234)
net = mdl.get_default_net(9, cfg).to("cuda")
net.train()
r, s = config.ratios_scales(cfg)
lf = loss.get_default_loss(r, s, cfg)
opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))
batch = {k: v.cuda() for k, v in synthetic_batch(16, 300, 300, T=20, seed=1234).items()}
for _ in range(3):
    opt.zero_grad()
    lf(net(batch), batch)["loss"].mean().backward()
    opt.step()
torch.cuda.synchronize()
plan = next(iter(net._plans.values()))
for which in (sys.argv[1:] or ["fwd"]):
    prog = getattr(plan, which)
    runs = [prog.profile(stream_ptr()) for _ in range(5)]
    med = [sorted(rr[i][2] for rr in runs)[2] for i in range(len(prog.calls))]
    tot = {0: 0.0, 1: 0.0, 2: 0.0}
    print(f"== {which}: {len(prog)} launches (median of 5 event-timed replays, us; each launch alone)")
    for i, ((fn, args, what), lane) in enumerate(zip(prog.calls, prog.lanes)):
        tot[lane] += med[i]
        print(f"{i:4d} L{lane} {med[i] * 1e3:8.1f} {fn.__name__:34s} {what}")
    print(f"# sum main (L0+L2) {1e3 * (tot[0] + tot[2]):.1f} us, side (L1) {1e3 * tot[1]:.1f} us")
