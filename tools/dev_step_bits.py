"""Developer tool (GPU box): two training steps of the bench network in deterministic mode (ZSG_DETERMINISTIC=1, a shared ZSG_TUNE_CACHE so that
both builds lower the same tiles) -> outputs, loss and the flat gradient in a file; a second build (ZSG_LIB_PATH) must reproduce them bit for bit.
usage: ZSG_DETERMINISTIC=1 ZSG_SHIPPED_TUNE=0 ZSG_TUNE_CACHE=<f> python tools/dev_step_bits.py <out.pt> [other.pt]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, loss, mdl, optim
from zsgnet_pytorch_amd.synth import synthetic_batch

res = {}
for arch, B, img in (("resnet50", 16, 300), ("resnet18", 2, 128)):
    cfg = config.get_cfg(resnet_arch=arch, bs=B, resize_img=[img, img], mdl_to_use="retina")
    torch.manual_seed(1234)
    net = mdl.get_default_net(9, cfg).to("cuda")
    net.train()
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))
    batch = {k: v.cuda() for k, v in synthetic_batch(B, img, img, T=20, seed=1234).items()}
    batch["h0"], batch["c0"] = torch.zeros(2, B, 128), torch.zeros(2, B, 128)
    for it in range(2):
        opt.zero_grad()
        out = net(batch)
        ls = lf(out, batch)
        ls["loss"].mean().backward()
        res[f"{arch}/{it}"] = (out["att_bbx_out"].detach().cpu(), ls["loss"].detach().cpu(), net.store.grad.detach().cpu().clone())
        opt.step()
    torch.cuda.synchronize()
if len(sys.argv) > 2:
    other = torch.load(sys.argv[2])
    for k in res:
        same = [torch.equal(a, b) for a, b in zip(res[k], other[k])]
        print(k, "bit-identical" if all(same) else f"DIFFERENT (out, loss, grad) = {same}; max grad diff {float((res[k][2] - other[k][2]).abs().max()):.3e}")
torch.save(res, sys.argv[1])
