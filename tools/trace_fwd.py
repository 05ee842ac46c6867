"""Developer tool (GPU box): the forward-only leg of bench.py (train mode, grad mode on, 12 forwards back to back) as a workload for
`rocprofv3 --kernel-trace`, and — with a trace CSV as argument — the listing of ONE steady-state forward in start order: queue, start
offset, duration, idle time of its queue before the launch (the dependent-launch boundaries of the conv -> BatchNorm chain).
usage: rocprofv3 --kernel-trace -d /tmp/kt --output-format csv -- python tools/trace_fwd.py
       python tools/trace_fwd.py /tmp/kt/.../*kernel_trace.csv [forward index from the end, default 3]"""
import csv
import os
import re
import sys

if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    heads = [i for i, r in enumerate(rows) if "nchw_to_nhwc4" in r[2]]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lo, hi = heads[-k - 1], heads[-k]
    fw = rows[lo:hi]
    t0 = fw[0][0]
    main_q = fw[0][3]
    print(f"forward: {len(fw)} launches, {(rows[hi][0] - t0) / 1e3:.1f} us from its first launch to the next forward's first launch")
    qend, qs = {}, sorted(set(r[3] for r in fw), key=lambda q: q != main_q)
    tot_gap, n_gap, busy = 0.0, 0, 0.0
    for s, e, n, q in fw:
        gap = (s - qend[q]) / 1e3 if q in qend else 0.0
        qend[q] = e
        if q == main_q:
            busy += (e - s) / 1e3
            if gap > 0:
                tot_gap += gap
                n_gap += 1
        short = re.sub(r"\(.*", "", n).replace("void ", "")[:60]
        print(f"  q{qs.index(q)} +{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {short}")
    print(f"# main queue: kernels {busy:.1f} us, gaps {tot_gap:.1f} us over {n_gap} boundaries ({tot_gap / max(n_gap, 1):.2f} us each)")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, loss, mdl, optim            # noqa: E402
from zsgnet_pytorch_amd.synth import synthetic_batch              # noqa: E402

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
for _ in range(12):
    net(batch)
torch.cuda.synchronize()
