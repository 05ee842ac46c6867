"""Developer tool: is a gap seen in a rocprofv3 trace there WITHOUT the profiler?  Brackets launches [a, b] of a lowered program with
HIP events on the main stream in the un-profiled bench loop and prints the median elapsed time.
usage (GPU box): python tools/probe_gap.py fwd:0:0 fwd:0:16 fwd:14:16 ...      (program:first:last)"""
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
lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))
batch = {k: v.cuda() for k, v in synthetic_batch(16, 300, 300, T=20, seed=1234).items()}


def step():
    opt.zero_grad()
    out = net(batch)
    lf(out, batch)["loss"].mean().backward()
    opt.step()
    ev(out, batch)


for _ in range(12):
    step()
torch.cuda.synchronize()
plan = next(iter(net._plans.values()))
for spec in sys.argv[1:]:
    if spec == "tail":
        # from the end of the backward's last main-stream launch to the point where the main stream has joined the side stream
        # (the weight-gradient tail) and the optimizer's launch may start
        prog = plan.bwd
        last = max(i for i, l in enumerate(prog.lanes) if l != 1)
        saved, marks, armed = prog.calls[last], [], [False]
        fn0 = prog.calls[last][0]

        def f(*args):
            rc = fn0(*args)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append([e, None])
            armed[0] = True
            return rc
        f.__name__ = fn0.__name__
        prog.calls[last] = (f,) + prog.calls[last][1:]
        jg = net.join_grads

        def jg2():
            jg()
            if armed[0]:
                armed[0] = False
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks[-1][1] = e
        net.join_grads = jg2
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) * 1e3 for x, y in marks[5:])
        print(f"tail           [after {prog.calls[last][2]} .. side stream joined]  median {ts[len(ts) // 2]:8.1f} us   min {ts[0]:8.1f}   max {ts[-1]:8.1f}")
        prog.calls[last] = saved
        net.join_grads = jg
        continue
    which, a, b = spec.split(":")
    a, b = int(a), int(b)
    prog = getattr(plan, which)
    saved = (prog.calls[a], prog.calls[b])
    marks = []

    def wrap_first(fn):
        def f(*args):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append([e, None])
            return fn(*args)
        f.__name__ = fn.__name__
        return f

    def wrap_last(fn):
        def f(*args):
            rc = fn(*args)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks[-1][1] = e
            return rc
        f.__name__ = fn.__name__
        return f
    if a == b:
        fn0 = prog.calls[a][0]
        prog.calls[a] = (wrap_last(wrap_first(fn0)),) + prog.calls[a][1:]
    else:
        prog.calls[a] = (wrap_first(prog.calls[a][0]),) + prog.calls[a][1:]
        prog.calls[b] = (wrap_last(prog.calls[b][0]),) + prog.calls[b][1:]
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in marks[5:])
    print(f"{spec:14s} [{prog.calls[a][2]} .. {prog.calls[b][2]}]  median {ts[len(ts) // 2]:8.1f} us   min {ts[0]:8.1f}   max {ts[-1]:8.1f}")
    prog.calls[a], prog.calls[b] = saved
