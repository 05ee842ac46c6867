"""Developer tool: per-launch time of one forward + backward of the lowered network (HIP events around every launch),
with algorithmic GFLOP and achieved TFLOP/s for the convolution launches.
usage (GPU box): python tools/per_op_profile.py [arch] [B] [img] > gpurun_out/per_op.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd.synth import synthetic_batch
from zsgnet_pytorch_amd import config, loss, mdl
from zsgnet_pytorch_amd._lib import ConvDesc, stream_ptr


def conv_flops(args):
    import ctypes as C
    d = C.cast(args[0], C.POINTER(ConvDesc)).contents if not isinstance(args[0], ConvDesc) else args[0]
    fl = 0.0
    for i in range(d.nseg):
        s = d.seg[i]
        fl += 2.0 * d.B * s.rows_y * s.rows_x * d.N * s.ty.n * s.tx.n * d.C
    return fl, d


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    img = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    cfg = config.get_cfg(resnet_arch=arch)
    net = mdl.get_default_net(9, cfg).to("cuda").train()
    bt = {k: v.cuda() for k, v in synthetic_batch(B, img, img, seed=1).items()}
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    for _ in range(2):
        out = net(bt)
        lf(out, bt)["loss"].backward()
        for p in net.parameters():
            p.grad = None
    plan = list(net._plans.values())[0]
    st = stream_ptr()
    rows = []
    for prog in (plan.fwd, plan.prep, plan.bwd):
        keep = {id(k): k for k in prog.keep}
        res = prog.profile(st)
        for (what, fname, ms), (fn, args, _) in zip(res, prog.calls):
            gf = None
            extra = ""
            if fname in ("zsg_conv_igemm", "zsg_conv_wgrad"):
                d = args[0]._obj
                fl = sum(2.0 * d.B * d.seg[i].rows_y * d.seg[i].rows_x * d.N * d.seg[i].ty.n * d.seg[i].tx.n * d.C for i in range(d.nseg))
                gf = fl / 1e9
                M = sum(d.B * d.seg[i].rows_y * d.seg[i].rows_x for i in range(d.nseg))
                extra = f"M={M} N={d.N} C={d.C} taps={d.seg[0].ty.n}x{d.seg[0].tx.n} nseg={d.nseg}"
            rows.append((prog.name, what, fname, ms, gf, extra))
    tot = sum(r[3] for r in rows)
    print(f"# {arch} B={B} {img}x{img}: {len(rows)} launches, {tot:.3f} ms total (event-timed, serialised by events)")
    agg = {}
    for pr, what, fname, ms, gf, extra in rows:
        a = agg.setdefault((pr, fname), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += ms
        a[2] += gf or 0
    print("# per program / function")
    for (pr, fname), (n, ms, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{pr:9s} {fname:24s} n={n:4d} {ms:9.3f} ms  {gf / ms if gf else 0:8.1f} TF/s" if gf else f"{pr:9s} {fname:24s} n={n:4d} {ms:9.3f} ms")
    print("# per launch (sorted by time)")
    for pr, what, fname, ms, gf, extra in sorted(rows, key=lambda r: -r[3])[:140]:
        tf = f"{gf / ms:7.1f} TF/s {gf:8.2f} GF" if gf else " " * 27
        print(f"{ms:8.4f} ms {tf} {pr:8s} {what:52s} {extra}")


if __name__ == "__main__":
    main()
