"""Developer tool: per-parameter gradient error of the SSD-VGG path vs the fp64 oracle (HIP and CPU-fp32)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zsg_oracle as O
from zsgnet_pytorch_amd import config, loss, mdl

cfg = config.get_cfg(mdl_to_use="ssd_vgg")
net = mdl.get_default_net(9, cfg)
sd = O.seeded_ssd_state_dict(5)
net.load_state_dict(sd)
net.to("cuda").train()
r, s = config.ratios_scales(cfg)
lf = loss.get_default_loss(r, s, cfg)
bt = O.synthetic_batch(1, 300, 300, seed=31)
g = torch.Generator().manual_seed(77)
h0, c0 = torch.randn(2, 1, 128, generator=g), torch.randn(2, 1, 128, generator=g)
inp = {k: v.cuda() for k, v in bt.items()}
inp["h0"], inp["c0"] = h0, c0
out = net(inp)
lf(out, inp)["loss"].backward()
torch.cuda.synchronize()
anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(300, 300), r, s).astype(np.float32))
for k, v in sd.items():
    v.requires_grad_(True)
ref = O.zsgnet_forward(sd, bt, h0, c0, arch="ssd_vgg")
O.torch_loss(ref, bt["annot"], anc)["loss"].backward()
sd64 = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
bt64 = {k: v.double() for k, v in bt.items()}
ref64 = O.zsgnet_forward(sd64, bt64, h0.double(), c0.double(), arch="ssd_vgg")
O.torch_loss(ref64, bt["annot"], anc)["loss"].backward()
o5 = out["att_bbx_out"].detach().cpu().double()
o64 = torch.cat([ref64["bbx_out"], ref64["att_out"]], 2).detach()
o32 = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach().double()
print("fwd err HIP", float((o5 - o64).abs().max()), "CPU", float((o32 - o64).abs().max()))
for n, p in net.named_parameters():
    if sd64[n].grad is None:
        continue
    g64 = sd64[n].grad.flatten()
    eg = float((p.grad.cpu().double().flatten() - g64).norm() / (g64.norm() + 1e-30))
    ec = float((sd[n].grad.double().flatten() - g64).norm() / (g64.norm() + 1e-30))
    print(f"{n:45s} HIP {eg:.3e} CPU {ec:.3e} ratio {eg / (ec + 1e-30):6.1f}")

# ---- activation-gradient comparison (fp64 oracle) for the head input / first head activation per level -------------
import torch.nn.functional as F
sd64b = {k: v.detach().double().requires_grad_(True) for k, v in sd.items()}
we = O.query_encoder(sd64b, bt64["qvec"], bt64["qlens"], h0.double(), c0.double())
feats = O.ssd_forward(sd64b, bt64["img"])
for f in feats:
    f.retain_grad()
h1s, outs = [], []
for f in feats:
    x = O.fuse_lang_grid(f, we)
    h1 = F.relu(F.conv2d(x, sd64b["att_reg_box.0.0.weight"], sd64b["att_reg_box.0.0.bias"], 1, 1))
    h1.retain_grad()
    h1s.append(h1)
    y = h1
    for i in range(1, 5):
        y = F.relu(F.conv2d(y, sd64b[f"att_reg_box.{i}.0.weight"], sd64b[f"att_reg_box.{i}.0.bias"], 1, 1))
    y = F.conv2d(y, sd64b["att_reg_box.5.weight"], sd64b["att_reg_box.5.bias"], 1, 1)
    outs.append(y.permute(0, 2, 3, 1).contiguous().view(1, -1, 5))
ab = torch.cat(outs, 1)
O.torch_loss(dict(att_out=ab[..., 4:5], bbx_out=ab[..., :4]), bt["annot"], anc)["loss"].backward()
plan = list(net._plans.values())[0]
for name, refs in (("head.h1", h1s), ("head.dfeat", feats)):
    a = plan.acts[name]
    ga = a.grad if name == "head.h1" else a
    for i, r in enumerate(refs):
        got = ga.tensor(i).cpu().double().permute(0, 3, 1, 2)
        ref_g = r.grad
        if name == "head.h1":
            ref_g = ref_g * (r > 0)
        err = (got - ref_g).abs()
        idx = np.unravel_index(int(err.argmax()), tuple(err.shape))
        print(f"{name} level {i}: rel err {float((got - ref_g).norm() / ref_g.norm()):.3e}  max abs {float(err.max()):.3e} at {idx} (ref {float(ref_g[idx]):.3e})  #bad {(err > 1e-5 * float(ref_g.abs().max())).sum().item()}")
