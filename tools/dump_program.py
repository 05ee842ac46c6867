"""Developer tool: print the lowered backward / forward launch programs of the bench configuration (index, lane, name).
usage (GPU box): python tools/dump_program.py [fwd|bwd|prep|prep_u]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim            # noqa: E402
from zsgnet_pytorch_amd.synth import synthetic_batch                         # noqa: E402

cfg = config.get_cfg(resnet_arch="resnet50", bs=16, resize_img=[300, 300], mdl_to_use="retina")
torch.manual_seed(1234)
net = mdl.get_default_net(9, cfg).to("cuda")
net.train()
r, s = config.ratios_scales(cfg)
lf = loss.get_default_loss(r, s, cfg)
opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))
batch = {k: v.cuda() for k, v in synthetic_batch(16, 300, 300, T=20, seed=1234).items()}
for _ in range(2):
    opt.zero_grad()
    lf(net(batch), batch)["loss"].mean().backward()
    opt.step()
torch.cuda.synchronize()
plan = next(iter(net._plans.values())) if hasattr(net, "_plans") else None
if plan is None:
    plan = [v for v in vars(net).values() if isinstance(v, dict) and v and hasattr(next(iter(v.values())), "bwd")][0]
    plan = next(iter(plan.values()))
for which in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["bwd"]):
    prog = getattr(plan, which)
    print(f"== {which}: {len(prog)} launches, wait_idx={getattr(plan, '_wait_idx', None)}")
    for i, ((fn, args, what), lane) in enumerate(zip(prog.calls, prog.lanes)):
        print(f"{i:4d} L{lane} {fn.__name__:34s} {what}")
    if which == "bwd" and "--sched" in sys.argv:
        ops, nev, busy = prog._schedule(0, len(prog), False, False)
        print("== schedule ops with cross-stream edges")
        for op in ops:
            if op[0] in ("m", "s"):
                print(f"  {op[0]} {op[1]:4d} ev{op[2]:3d} {prog.calls[op[1]][2]}")
            else:
                print(f"  {op[0]} ev{op[1]}")
