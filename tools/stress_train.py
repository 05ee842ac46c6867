"""Developer tool: 400 training steps at the benchmark configuration — the loss must stay finite and device memory constant."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, evaluator, loss, mdl, optim
from zsgnet_pytorch_amd.synth import synthetic_batch
cfg = config.get_cfg()
net = mdl.get_default_net(9, cfg).to("cuda").train()
r, s = config.ratios_scales(cfg)
lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
opt = optim.FusedAdam(net, lr=1e-4, betas=(0.9, 0.99))
pool = [{k: v.cuda() for k, v in synthetic_batch(16, 300, 300, seed=100 + i).items()} for i in range(16)]
m0 = None
for it in range(400):
    bt = pool[it % 16]
    opt.zero_grad(); out = net(bt); ls = lf(out, bt); ls["loss"].mean().backward(); opt.step(); em = ev(out, bt)
    if it % 50 == 49:
        torch.cuda.synchronize()
        l = float(ls["loss"]); mem = torch.cuda.memory_allocated() / 2**20
        m0 = m0 or mem
        print(f"it {it+1} loss {l:.4f} acc {float(em['Acc']):.3f} mem {mem:.0f} MiB", flush=True)
        assert l == l and abs(mem - m0) < 64, "NaN loss or memory growth"
print("stress ok")
