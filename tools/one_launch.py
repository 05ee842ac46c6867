"""Developer tool (GPU box): replay ONE launch of the lowered bench network (ResNet-50 FPN, 300^2, B=16) — exactly what ships: the
descriptor, tile hint and fused epilogue the plan uses — `reps` times after two full steps, for rocprofv3 --pmc runs
(tools/pmc_round5.sh takes the LAST reps dispatches of the matching kernel).  Prints the entry point, the tile hint and the shapes.
usage: python tools/one_launch.py <fwd|bwd|prep> <substring of the launch's name> [reps]"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, loss, mdl, optim            # noqa: E402
from zsgnet_pytorch_amd._lib import ConvDesc, stream_ptr           # noqa: E402
from zsgnet_pytorch_amd.synth import synthetic_batch              # noqa: E402

which, pat = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
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
plan = next(iter(net._plans.values()))
prog = getattr(plan, which)
idx = [i for i, c in enumerate(prog.calls) if pat in c[2]]
assert idx, f"no launch named *{pat}* in {which}: " + ", ".join(c[2] for c in prog.calls[:400])
i = idx[0]
fn, args, what = prog.calls[i]
info = {"program": which, "index": i, "launch": what, "entry": fn.__name__}
d = getattr(args[0], "_obj", None) if args else None
if isinstance(d, ConvDesc):
    h = d.tile_hint
    info.update(hint=hex(h), BM=h & 0xff, BN=(h >> 8) & 0xff, splits=(h >> 16) & 0xff, w8_or_ps4=(h >> 24) & 1, k64=(h >> 27) & 1,
                B=d.B, C=d.C, N=d.N, nseg=d.nseg, rows=sum(d.B * d.seg[j].rows_y * d.seg[j].rows_x for j in range(d.nseg)),
                taps=f"{d.seg[0].ty.n}x{d.seg[0].tx.n}")
print("ONE_LAUNCH", info, flush=True)
st = C.c_void_p(stream_ptr())
for _ in range(reps):
    rc = fn(*args, st)
    assert rc == 0, rc
torch.cuda.synchronize()
