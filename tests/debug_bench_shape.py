"""Developer tool: forward / loss / gradient agreement with the CPU oracle at the benchmark shape (ResNet-50 FPN 300x300,
B=8): exercises the tile variants the autotuner picks for large launches.  Head gradients agree to ~1e-6; the
difference grows with depth through the train-mode BatchNorm stack (fp32 vs fp32, see tests/test_gpu_net.py)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zsg_oracle as O
from zsgnet_pytorch_amd import config, loss, mdl
B = 8
cfg = config.get_cfg()
net = mdl.get_default_net(9, cfg)
sd = O.seeded_state_dict("resnet50", 3)
net.load_state_dict(sd); net.to("cuda").train()
r, s = config.ratios_scales(cfg)
lf = loss.get_default_loss(r, s, cfg)
bt = O.synthetic_batch(B, 300, 300, seed=77)
g = torch.Generator().manual_seed(1)
h0, c0 = torch.randn(2, B, 128, generator=g), torch.randn(2, B, 128, generator=g)
inp = {k: v.cuda() for k, v in bt.items()}; inp["h0"], inp["c0"] = h0, c0
out = net(inp); ls = lf(out, inp); ls["loss"].backward(); torch.cuda.synchronize()
for k, v in sd.items():
    if v.is_floating_point() and "running" not in k: v.requires_grad_()
ref = O.zsgnet_forward(sd, bt, h0, c0)
anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(300, 300), r, s).astype(np.float32))
lr = O.torch_loss(ref, bt["annot"], anc); lr["loss"].backward()
o_gpu = out["att_bbx_out"].detach().cpu(); o_cpu = torch.cat([ref["bbx_out"], ref["att_out"]], 2).detach()
print("forward max abs err", float((o_gpu - o_cpu).abs().max()), "loss", ls["loss"].item(), lr["loss"].item())
worst = []
for n, p in net.named_parameters():
    gg, gc = p.grad.cpu().double().flatten(), sd[n].grad.double().flatten()
    worst.append((float((gg - gc).norm() / (gc.norm() + 1e-30)), n))
worst.sort(reverse=True)
print("worst relative gradient differences (HIP vs CPU fp32 oracle):", [(round(e, 4), n) for e, n in worst[:6]])
print("median", sorted(e for e, _ in worst)[len(worst) // 2])
for n in ("att_reg_box.5.bias", "att_reg_box.5.weight", "att_reg_box.4.0.weight", "att_reg_box.0.0.weight", "backbone.fpn.P3_2.weight", "backbone.encoder.layer4.2.conv3.weight", "backbone.encoder.conv1.weight", "lstm.weight_ih_l0"):
    p = dict(net.named_parameters())[n]
    gg, gc = p.grad.cpu().double().flatten(), sd[n].grad.double().flatten()
    cos = float((gg @ gc) / (gg.norm() * gc.norm()))
    print(f"{n:45s} rel {float((gg - gc).norm() / gc.norm()):.5f}  norm ratio {float(gg.norm() / gc.norm()):.5f}  cos {cos:.6f}")
