#!/usr/bin/env python
"""bench.py — ZSGNet training-step throughput on MI355X (BASELINE.json metric: train images/sec).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU)

A "step" is the reference's hot loop body, utils.py:407-414: zero_grad -> ZSGNet.forward -> ZSGLoss -> backward
(-> bucketed RCCL all-reduce) -> Adam -> Evaluator, on a synthetic batch already resident in HBM (SURVEY.md §8d:
img ~ U[0,1) 300x300, 20-token queries, per-GPU batch 16 = BASELINE configs[1]).  W warm-up steps, then exactly K timed
steps between barrier + torch.cuda.synchronize(); MAX over ranks; rank 0 prints ONE JSON line.

`median_ms_per_step` comes from HIP events recorded at the step boundaries INSIDE the timed region (no extra sync).

Extra legs (rank 0, outside the timed region):
  roofline     — per-launch HIP-event timing of every kernel (zsg_prof_*, entries named as rocprofv3 names them); the
                 kernel with the largest share is reported as achieved TFLOP/s = ALGORITHMIC 2*MAC (direct convolution) /
                 event time against the 157.3 TFLOP/s fp32-MFMA peak: with every launch alone on the GPU (primary) and as
                 timed (side stream on).  The Winograd kernels (wino_kernel, wino_wgrad_kernel) execute 4/9 of the
                 algorithmic multiply-adds on the MFMA pipe, so their algorithmic fraction can exceed 1; `mfma_executed`
                 gives the pipe's own utilisation (algorithmic x 4/9).  `traffic` (HBM bytes per launch, rocprofv3 --pmc
                 FETCH_SIZE x2 + WRITE_SIZE) is read from profiles/<round>_hbm_traffic.json ONLY when that file was
                 produced from the same kernel sources (sha256 stamp), else null.
  forward      — train-mode forward only (the north_star's ">= 70 % on the ResNet-50+FPN forward"): HIP-event median.
  cpu_baseline — the CPU oracle's same step (torch-CPU fp32, B=4) on the host cores ("port"); N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# forward conv 2*MAC per image, keyed by (arch, input size); a training step is 3x
FWD_GF = {("resnet50", 300): 32.569, ("ssd_vgg", 300): 75.003, ("resnet50", 600): 85.911, ("resnet101", 600): 140.609}
PEAK_TF = 157.3                                                         # fp32-input MFMA, MI355X_MICROARCH.md


def traffic_file(stamp: str):
    """The rocprofv3 --pmc traffic summary that belongs to the sources this process runs: the newest profiles/rNN_hbm_traffic.json whose
    sha256 source stamp matches (no hard-coded round name: a summary of another build is never picked up, a forgotten constant cannot
    silently null the traffic field).  Returns (path | None, note | None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")), reverse=True)
    other = []
    for f in files:
        try:
            st = json.load(open(f)).get("_source_stamp")
        except Exception as e:
            other.append(f"{os.path.basename(f)}: unreadable ({e})")
            continue
        if st == stamp:
            return f, None
        other.append(f"{os.path.basename(f)}: stamp {st}")
    if not files:
        return None, "no rocprofv3 --pmc summary under profiles/ (tools/rocprof_round.sh)"
    return None, f"no rocprofv3 --pmc summary was measured on these kernel sources (stamp {stamp}); found " + "; ".join(other[:3])


class ClockSampler:
    """Shader clock of this rank's GPU while the timed region runs (sysfs pp_dpm_sclk: the line marked '*'), sampled from a host thread
    every 20 ms — no device work, no synchronisation.  The driver's box and the builder's boxes differ by 1.5-3 % in step time; this
    field says whether the clock explains it."""

    def __init__(self, device_index: int):
        import glob
        self.path, self.samples, self._stop, self._thr = None, [], False, None
        try:
            bus = torch.cuda.get_device_properties(device_index).pci_bus_id
            dom = getattr(torch.cuda.get_device_properties(device_index), "pci_domain_id", 0)
            dev = getattr(torch.cuda.get_device_properties(device_index), "pci_device_id", 0)
            want = f"{dom:04x}:{bus:02x}:{dev:02x}"
        except Exception:
            want = None
        cands = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
        for c in cands:
            try:
                slot = [ln.split("=", 1)[1].strip() for ln in open(os.path.join(os.path.dirname(c), "uevent")) if ln.startswith("PCI_SLOT_NAME")]
            except Exception:
                slot = []
            if want and slot and slot[0].lower().startswith(want):
                self.path = c
        if self.path is None and len(cands) == 1:
            self.path = cands[0]

    def _read(self):
        try:
            for ln in open(self.path):
                if "*" in ln:
                    return int("".join(ch for ch in ln.split(":", 1)[1] if ch.isdigit()))
        except Exception:
            return None
        return None

    def start(self):
        if self.path is None:
            return self
        import threading

        def run():
            while not self._stop:
                v = self._read()
                if v:
                    self.samples.append(v)
                time.sleep(0.02)
        self._thr = threading.Thread(target=run, daemon=True)
        self._thr.start()
        return self

    def stop(self):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)
        if not self.samples:
            return None
        s_ = sorted(self.samples)
        return {"median": s_[len(s_) // 2], "min": s_[0], "max": s_[-1], "samples": len(s_), "source": self.path}


def config_label(arch: str, backbone: str, img: int, bs: int, world: int) -> str:
    """Which BASELINE.json configs[] entry this run has the (per-GPU) shape of"""
    if backbone == "ssd_vgg" and img == 300:
        return f"BASELINE configs[3] shape{'' if bs == 32 else f' at per-GPU bs={bs} (configs[3]: 32)'}"
    if arch == "resnet101" and img == 600:
        return f"BASELINE configs[4] per-GPU shape{'' if bs == 32 else f' at per-GPU bs={bs} (configs[4]: 32)'}"
    if arch == "resnet50" and img == 300 and bs == 16:
        return "BASELINE configs[1] shape" if world == 1 else f"BASELINE configs[2] per-GPU shape, {world} ranks"
    return "not a BASELINE configs[] shape"


def exec_ratio(kernel: str) -> float:
    """MFMA multiply-adds executed / algorithmic (direct-convolution) multiply-adds"""
    return 4.0 / 9.0 if kernel.startswith(("wino_kernel", "wino_wgrad_kernel")) else 1.0


def source_stamp() -> str:
    """sha256 over the kernel sources: ties a committed rocprof summary / the shipped tuning table to the build it was measured on"""
    from zsgnet_pytorch_amd import ops as _o
    return _o.source_stamp()


def tuning_info() -> dict:
    """which tile choices this run used: the shipped table (when its stamp matches the kernel sources) and / or fresh tunings"""
    from zsgnet_pytorch_amd import ops as _o
    t = dict(_o.TUNE_INFO)
    t["stamp_match"] = bool(t["table"]) and t["table_stamp"] == t["stamp"]
    t["tune_cache_env"] = os.environ.get("ZSG_TUNE_CACHE") or None
    t["what"] = ("tile / split-K choices: `loaded` entries from the shipped table (used only when its source stamp matches), `tuned_now` launch "
                 "shapes autotuned by this process (median of interleaved samples)")
    return t


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--bs", type=int, default=16, help="per-GPU batch (configs[1]: 16)")
    ap.add_argument("--arch", default="resnet50")
    ap.add_argument("--backbone", default="retina", choices=["retina", "ssd_vgg"], help="retina = ResNet(--arch)+FPN; ssd_vgg = config 4")
    ap.add_argument("--img", type=int, default=300)
    ap.add_argument("--tokens", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--forward-leg", action="store_true", help="time the forward-only leg also under --no-roofline (tuning tools)")
    ap.add_argument("--no-bx", action="store_true", help="(accepted and ignored: the bf16x6 leg was removed in round 4)")
    ap.add_argument("--prof-out", default="")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in the RCCL data-parallel reducer even at world size 1 (smoke test)")
    ap.add_argument("--settle", type=int, default=-1, help="untimed settle steps AFTER the requested warm-up (default: until the step time is "
                    "stable, between 25 and 60 steps; 0 = none)")
    ap.add_argument("--other-configs", default="auto", choices=["auto", "on", "off"], help="after the headline legs, time BASELINE configs[3] / [4]'s "
                    "per-GPU shapes (SSD-VGG16 300^2 B=32, ResNet-101 FPN 600^2 B=32) in this process; auto = only for the default headline "
                    "configuration at N=1 with a stamp-matched shipped tuning table (otherwise they would be autotuned for minutes)")
    ap.add_argument("--ddp-variants", default="auto", choices=["auto", "on", "off"], help="data-parallel runs only: after the headline legs, two short "
                    "legs with ONE thing changed each — the chain's wave priority off (zsg_set_main_priority(0): how do RCCL's priority-0 kernels "
                    "fare beside a priority-3 backward?) and the other transport (ZSG_COMM torch <-> native) — printed beside the default in `rccl`; "
                    "auto = whenever the run is data-parallel (N > 1 or --force-ddp)")
    ap.add_argument("--refine", default="auto", choices=["auto", "off"], help="auto: when this process autotuned any launch shape itself, re-rank the "
                    "tuner's near-ties inside the real step before the timed region (ZSGNet.refine_tuning; untimed, N = 1 only)")
    ap.add_argument("--launch-check", action="store_true", help="only bring up the N-rank process group (backend ZSG_DIST_BACKEND, default "
                    "nccl), all-reduce one tensor and print a JSON line: tests the launcher without a GPU (gloo)")
    return ap.parse_args()


def self_launch(a) -> int:
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU of this node, env://
    rendezvous on 127.0.0.1 at a free port (the reference does the same through `python -m torch.distributed.launch
    --nproc_per_node=$ngpus code/main_dist.py`, README.md:69-76 / main_dist.py:70-77).  The children's stdout is ours."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL / cross-process device memory on this driver)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(arch, img, tokens):
    """The oracle's training step on the host cores (kind 'port'): B=4, 2 warm-up + 5 timed steps (~20 s)."""
    import numpy as np
    from oracle import zsg_oracle as O
    B = 4
    sd = O.seeded_state_dict(arch, 0)
    params = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    opt = torch.optim.Adam(list(params.values()), lr=1e-4, betas=(0.9, 0.99))
    bt = O.synthetic_batch(B, img, img, T=tokens, seed=1234)
    r, s = O.default_ratios_scales()
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(img, img), r, s).astype(np.float32))
    times = []
    for it in range(7):
        h0, c0 = torch.randn(2, B, 128), torch.randn(2, B, 128)
        t0 = time.perf_counter()
        O.cpu_train_step(params, buffers, opt, bt, h0, c0, anc, arch=arch)
        times.append(time.perf_counter() - t0)
    med = sorted(times[2:])[2]
    return {"value": round(B / med, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle.cpu_train_step, {arch} {img}x{img}, B={B}, 2 warm-up + 5 timed steps, median {med:.3f} s/step"}


def time_config(arch: str, backbone: str, img: int, bs: int, tokens: int, warm: int, steps: int) -> dict:
    """One more configuration in this process (other_configs leg): the same step as the headline's — zero_grad, forward, loss, backward,
    Adam, evaluator — on a synthetic batch resident in HBM; `warm` untimed steps (plan lowering included), `steps` timed ones between
    two synchronisations.  Reference: code/ssd_vgg.py:54-102 (configs[3]), code/fpn_resnet.py with resnet101 at 600x600 (configs[4])."""
    from zsgnet_pytorch_amd import config, evaluator, loss, mdl, ops as zops, optim
    from zsgnet_pytorch_amd.synth import synthetic_batch
    tuned0 = zops.TUNE_INFO["tuned_now"]
    cfg = config.get_cfg(resnet_arch=arch, bs=bs, resize_img=[img, img], mdl_to_use=backbone)
    torch.manual_seed(1234)
    net = mdl.get_default_net(9, cfg).to("cuda")
    net.train()
    r, s = config.ratios_scales(cfg)
    lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
    opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))
    batch = {k: v.cuda() for k, v in synthetic_batch(bs, img, img, T=tokens, seed=1234).items()}

    def step():
        opt.zero_grad()
        out = net(batch)
        ls = lf(out, batch)
        ls["loss"].mean().backward()
        opt.step()
        return ls, ev(out, batch)
    for _ in range(warm):
        ls, em = step()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        ls, em = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    key = "ssd_vgg" if backbone == "ssd_vgg" else arch
    gf = FWD_GF.get((key, img))
    ips = bs * steps / dt
    res = {"workload": f"{'SSD-VGG16' if backbone == 'ssd_vgg' else arch + '+FPN'} {img}x{img}, per-GPU bs={bs}, {tokens}-token queries ({config_label(key, backbone, img, bs, 1)})",
           "images_per_s": round(ips, 1), "ms_per_step": round(1e3 * dt / steps, 3), "median_ms_per_step": round(per[len(per) // 2], 3), "steps": steps,
           "warmup": warm, "step_mfma_frac": round(ips * 3 * gf * 1e9 / (PEAK_TF * 1e12), 4) if gf else None, "final_loss": round(float(ls["loss"].detach()), 4),
           "shapes_autotuned_now": zops.TUNE_INFO["tuned_now"] - tuned0}
    del net, opt, batch
    torch.cuda.empty_cache()
    return res


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))                    # no launcher around us: become one (one rank per GPU)
    if a.launch_check:
        from zsgnet_pytorch_amd import dist as zdist
        backend = os.environ.get("ZSG_DIST_BACKEND", "nccl")
        zdist.init_process_group_from_env(backend)
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.ones(4, device=dev) * (rank + 1)
        if world > 1:
            dist.all_reduce(t)
            dist.barrier()
        ok = float(t[0]) == world * (world + 1) / 2
        if dist.is_initialized():
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"launch_check": ok, "n_gpus": world, "backend": backend}), flush=True)
        raise SystemExit(0 if ok else 1)
    local = local % max(1, torch.cuda.device_count())      # (a 1-GPU box can host a 2-rank gloo dry run of the N > 1 control flow)
    torch.cuda.set_device(local)
    from zsgnet_pytorch_amd import config, dist as zdist, evaluator, loss, mdl, optim
    from zsgnet_pytorch_amd._lib import ProfEntry, lib
    if world > 1:
        zdist.init_process_group_from_env(os.environ.get("ZSG_DIST_BACKEND", "nccl"))       # nccl = RCCL over xGMI
    elif a.force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1)

    cfg = config.get_cfg(resnet_arch=a.arch, bs=a.bs, resize_img=[a.img, a.img], mdl_to_use=a.backbone)
    if a.backbone == "ssd_vgg":
        a.arch = "ssd_vgg"
    torch.manual_seed(1234)                       # identical initial weights on every rank (and C3 broadcasts anyway)
    net = mdl.get_default_net(9, cfg).to("cuda")
    net.train()
    model = (zdist.DistributedDataParallel(net, device_ids=[local], broadcast_buffers=True, force_collectives=a.force_ddp)
             if (world > 1 or a.force_ddp) else net)       # --force-ddp: every collective also in a 1-rank group (values unchanged)

    r, s = config.ratios_scales(cfg)
    lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
    opt = optim.FusedAdam(net, lr=cfg["lr"], betas=(0.9, 0.99))

    from zsgnet_pytorch_amd.synth import synthetic_batch
    bt = synthetic_batch(a.bs, a.img, a.img, T=a.tokens, seed=1234 + rank)
    batch = {k: v.cuda() for k, v in bt.items()}
    torch.manual_seed(99 + rank)                  # LSTM initial states (mdl.py:279-294) are drawn on the host each step

    def step():
        opt.zero_grad()
        out = model(batch)
        ls = lf(out, batch)
        ls["loss"].mean().backward()
        opt.step()
        return ls, ev(out, batch)

    for _ in range(a.warmup):
        ls, em = step()
    # Shapes this process had to tune itself (no stamp-matched shipped table): the tuner's near-ties are re-ranked INSIDE the step
    # (ops.refine_in_step: both streams, real neighbours) before anything is timed — what round 5 did by hand with six fresh tunings.
    # Untimed; single-GPU runs only (under DDP rank 0's table is broadcast before the other ranks lower).
    if model is net and a.refine != "off" and tuning_info()["tuned_now"] > 0:
        net.refine_tuning(batch, log=(lambda m: print(m, file=sys.stderr, flush=True)) if os.environ.get("ZSG_REFINE_LOG") else None)
        for _ in range(3):
            ls, em = step()

    # Untimed settle phase AFTER the requested warm-up: the shader clock leaves its idle state and the launch queues
    # fill over the first few dozen steps (a 5-step warm-up is 70 ms of GPU work), which put the driver's 20-step mean 1.7 % above its
    # median in round 4.  Steps run until the last 8 event-timed steps lie within 0.5 % of their median (at least 25, at most 60 steps;
    # --settle N fixes the count).  Nothing here is timed or reported as throughput.
    settle_n = 0
    if a.settle != 0:
        lo, hi = (a.settle, a.settle) if a.settle > 0 else ((25, 60) if world == 1 else (30, 30))     # (N > 1: the same count on every rank)
        sev = [torch.cuda.Event(enable_timing=True)]
        sev[0].record()
        while settle_n < hi:
            ls, em = step()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            sev.append(e)
            settle_n += 1
            if settle_n >= lo and settle_n % 4 == 0:
                sev[-1].synchronize()
                last = [sev[i].elapsed_time(sev[i + 1]) for i in range(len(sev) - 9, len(sev) - 1)]
                med = sorted(last)[4]
                if max(abs(t - med) for t in last) <= 0.005 * med:
                    break

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    red0 = next((p.reducer for p in net._plans.values() if p.reducer is not None), None) if model is not net else None
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    clk = ClockSampler(local)
    fence()
    clk.start()
    t0 = time.perf_counter()
    marks[0].record()
    host_t = [t0]
    for i in range(a.steps):
        ls, em = step()
        marks[i + 1].record()               # (an event record is ~1 us of host time and no synchronisation)
        host_t.append(time.perf_counter())
    t_host = host_t[-1] - t0                # when the host had enqueued everything
    fence()
    dt = time.perf_counter() - t0
    gpu_clock = clk.stop()
    host = {"enqueue_ms_per_step": round(1e3 * t_host / a.steps, 3), "lead_ms_at_end": round(1e3 * (dt - t_host), 3),
            "first_steps_ms": [round(1e3 * (host_t[i + 1] - host_t[i]), 3) for i in range(min(4, a.steps))],
            "what": "host time to enqueue one step (mean; the first steps after the fence, before the launch queues fill and throttle "
                    "the host); how far ahead of the GPU the host was when it had enqueued the last timed step"}
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    median_ms = per_step[len(per_step) // 2]
    exposed_ms = None
    if red0 is not None:
        # the exposed all-reduce time is measured in its OWN short leg behind the timed region (every rank, same count): its two event
        # records per step no longer sit in the headline number
        red0.time_wait = True
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        red0.time_wait = False
        exposed_ms = red0.exposed_ms()
    # Data-parallel runs: the same step with ONE thing changed, so that the first multi-GPU run answers by itself (a) whether the chain's
    # wave priority starves or helps the collectives that run beside it and (b) which transport is faster.  Every
    # rank runs the same legs in the same order; untimed warm-up + 20 timed steps each, max over ranks; not part of `value`.
    variants = None
    if model is not net and a.ddp_variants != "off":
        def short_leg(n_warm=8, n=20):
            for _ in range(n_warm):
                step()
            fence()
            t0_ = time.perf_counter()
            for _ in range(n):
                step()
            fence()
            t_ = torch.tensor([time.perf_counter() - t0_], device="cuda", dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            red_ = next((p.reducer for p in net._plans.values() if p.reducer is not None), None)
            ex_ = None
            if red_ is not None:
                red_.time_wait = True
                for _ in range(6):
                    step()
                torch.cuda.synchronize()
                red_.time_wait = False
                ex_ = red_.exposed_ms()
            return {"ms_per_step": round(1e3 * float(t_.item()) / n, 3), "images_per_s": round(a.bs * world * n / float(t_.item()), 1),
                    "exposed_allreduce_ms": round(ex_, 3) if ex_ is not None else None}
        variants = {"what": "the headline step with one setting changed (20 timed steps each, max over ranks): wave priority of the dependent "
                            "chain's kernels off; the other all-reduce transport", "default": {"main_priority": lib.zsg_get_main_priority(),
                            "transport": "native" if model.comm is not None else "torch", "ms_per_step": round(1e3 * dt / a.steps, 3)}}
        try:
            prio0 = lib.zsg_get_main_priority()
            lib.zsg_set_main_priority(0 if prio0 else 3)
            variants["main_priority_%d" % (0 if prio0 else 3)] = short_leg()
            lib.zsg_set_main_priority(prio0)
            other = "torch" if model.comm is not None else "native"
            if other == "native" and dist.is_initialized() and dist.get_backend() != "nccl":
                raise RuntimeError("transport leg skipped: the native transport (RCCL inside libzsg.so) needs the nccl backend, one GPU per rank "
                                   f"(this run: {dist.get_backend()})")
            keep_model = model
            model = zdist.DistributedDataParallel(net, device_ids=[local], broadcast_buffers=True, force_collectives=a.force_ddp, comm=other)
            for p_ in net._plans.values():
                p_.reducer = None            # (rebuilt by the next backward on the new wrapper's transport)
            variants["transport_" + other] = short_leg()
            if model.comm is not None:
                model.close()
            model = keep_model
            object.__setattr__(net, "_ddp", keep_model)
            for p_ in net._plans.values():
                p_.reducer = None
            step()
            torch.cuda.synchronize()
        except Exception as e:               # (never lose the headline line to a secondary leg)
            variants["error"] = f"{type(e).__name__}: {e}"
    per_rank = None
    if world > 1:                            # every rank's own clock and exposed wait, for the first scaling curve
        mine = torch.tensor([1e3 * dt / a.steps, median_ms, exposed_ms if exposed_ms is not None else -1.0], device="cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"ms_per_step": [round(float(t[0]), 3) for t in allr], "median_ms_per_step": [round(float(t[1]), 3) for t in allr],
                    "exposed_allreduce_ms": [round(float(t[2]), 3) for t in allr]}
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val, acc = float(ls["loss"].detach()), float(em["Acc"])
    ips = a.bs * world * a.steps / dt

    # (the forward-only leg runs BEFORE the per-kernel profiled legs: bracketing every launch with timing events leaves the
    # process ~3 % slower afterwards)
    fwd = None
    if not a.no_roofline or a.forward_leg:          # every rank (the training forward of a DDP model broadcasts the BatchNorm buffers)
        # grad mode ON, as in a training step: the forward then also co-schedules the backward's weight preparation on the side
        # stream (mdl._Plan.run_forward), which a no_grad forward would leave out.  (The 21 forwards advance the BatchNorm running
        # statistics; nothing below depends on them.)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
        model(batch)
        torch.cuda.synchronize()
        evs[0].record()
        for i in range(20):
            model(batch)
            evs[i + 1].record()
        torch.cuda.synchronize()
        fm = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(20))[10]
        fgf = FWD_GF.get((a.arch, a.img))
        fwd = {"median_ms": round(fm, 3), "images_per_s": round(a.bs / fm * 1e3, 1),
               "mfma_frac": round(a.bs * fgf * 1e9 / (fm * 1e-3) / (PEAK_TF * 1e12), 4) if fgf else None,
               "what": "train-mode ZSGNet.forward only, grad mode on (batch-statistics BatchNorm; the backward's weight preparation "
                       "co-runs on the side stream as in a step), algorithmic conv FLOPs / fp32-MFMA peak"}
    roof, prof_rows = None, []
    if not a.no_roofline:
        # EVERY rank runs the profiled steps (they contain the data-parallel collectives); rank 0 reports its own kernels
        from zsgnet_pytorch_amd import ops as zops

        def profiled(nprof, side):
            """per-kernel HIP-event times (events recorded on the stream each kernel is launched on) over nprof steps"""
            keep, keep_g = zops.SIDE_STREAM, zops.HIP_GRAPH
            zops.SIDE_STREAM, zops.HIP_GRAPH = side, False      # per-launch events need eager launches
            step()                                    # settle into the mode
            torch.cuda.synchronize()
            lib.zsg_prof_enable(1)
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
            lib.zsg_prof_enable(0)
            zops.SIDE_STREAM, zops.HIP_GRAPH = keep, keep_g
            arr = (ProfEntry * 96)()
            n = lib.zsg_prof_collect(arr, 96)
            tot = sum(arr[i].ms for i in range(n))
            rows = []
            for i in range(n):
                e = arr[i]
                rows.append(dict(kernel=e.name.decode(), launches_per_step=e.launches / nprof, ms_per_step=e.ms / nprof,
                                 tflops=(e.flops / (e.ms * 1e9)) if e.ms > 0 and e.flops > 0 else None,
                                 gbps=(e.bytes / (e.ms * 1e6)) if e.ms > 0 and e.bytes > 0 else None,
                                 alg_bytes_per_launch=(e.bytes / e.launches) if e.launches and e.bytes > 0 else None,
                                 share=e.ms / tot if tot else 0))
            rows.sort(key=lambda x: -x["ms_per_step"])
            return rows, tot / nprof
        # isolated: every launch alone on the GPU (one stream) — the kernel's own duration, what the roofline fraction is
        # about; as timed: with the weight-gradient kernels co-running on the side stream (the timed region's mode), where
        # kernels time-share the CUs and one kernel's duration is no longer a property of that kernel
        iso_rows, iso_tot = profiled(2, False)
        prof_rows, tot = profiled(2, zops.SIDE_STREAM)
        dom = iso_rows[0]
        timed = next((r for r in prof_rows if r["kernel"] == dom["kernel"]), None)
        traffic, rp_avg, rp_ser, tnote = None, None, None, None   # HBM bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE)
        tfile, tnote = traffic_file(source_stamp())       # written by tools/rocprof_round.sh at the same sources (stamp-matched)
        if tfile:
            tj = json.load(open(tfile))
            ent = tj.get(dom["kernel"], {})
            traffic, rp_avg, rp_ser = ent.get("hbm_bytes_per_launch"), ent.get("rocprof_avg_ms"), ent.get("rocprof_avg_ms_serial")
            if traffic is None:
                tnote = f"{os.path.basename(tfile)} has no entry for {dom['kernel']}"
        flops_all = sum((r["tflops"] or 0) * r["ms_per_step"] for r in iso_rows)         # GFLOP per step over all MFMA kernels
        mfma_ms = sum(r["ms_per_step"] for r in iso_rows if r["tflops"])
        exe_all = sum((r["tflops"] or 0) * r["ms_per_step"] * exec_ratio(r["kernel"]) for r in iso_rows)
        if dom["tflops"]:
            # `achieved` / `frac` are what the matrix pipe EXECUTES: algorithmic 2*MAC of the direct convolution x the kernel's
            # executed share (Winograd F(2x2,3x3) / F(3x3,2x2): 16 multiplies per 2x2 tile and channel pair instead of 36 = 4/9),
            # so frac <= 1 by construction; the algorithmic rate (what the direct convolution would have needed) is carried beside it.
            er = exec_ratio(dom["kernel"])
            share_exec = (exe_all / flops_all) if flops_all else 1.0       # FLOP-weighted executed share over all MFMA kernels
            roof = {"bound": "mfma", "kernel": dom["kernel"],
                    "what": "the kernel with the largest share of the step's isolated kernel time (it changes from round to round as kernels get "
                            "faster: top_kernels lists the next ones with their own fractions, all_mfma_kernels is FLOP-weighted over all of them)",
                    "achieved": round(dom["tflops"] * er, 2), "peak": PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(dom["tflops"] * er / PEAK_TF, 4), "traffic": traffic, "traffic_note": tnote,
                    "algorithmic_bytes_per_launch": round(dom["alg_bytes_per_launch"]) if dom.get("alg_bytes_per_launch") else None,
                    "traffic_over_algorithmic": round(traffic / dom["alg_bytes_per_launch"], 3) if (traffic and dom.get("alg_bytes_per_launch")) else None,
                    "traffic_note2": "algorithmic bytes = every operand and output element once, from the launch descriptors (zsg_conv_alg_bytes); "
                                     "traffic / algorithmic > 1 = re-reads that reached the memory side of L2",
                    "flops": "EXECUTED MFMA FLOPs: algorithmic 2*MAC of the direct convolution" + (" x 4/9 (Winograd: 16 of 36 multiplies per 2x2 tile)" if er < 1 else ""),
                    "executed_over_algorithmic": round(er, 4),
                    "algorithmic_achieved": round(dom["tflops"], 2), "algorithmic_frac": round(dom["tflops"] / PEAK_TF, 4),
                    "ceiling_algorithmic": round(1.0 / share_exec, 4),
                    "ceiling_note": "algorithmic FLOPs / peak that the step's kernel mix permits at 100 % MFMA utilisation = 1 / sum(FLOP share x executed share)",
                    "avg_launch_ms": round(dom["ms_per_step"] / dom["launches_per_step"], 5), "launches_per_step": dom["launches_per_step"],
                    "kernel_ms_per_step": round(dom["ms_per_step"], 3), "all_kernels_ms_per_step": round(iso_tot, 3),
                    "rocprof_avg_launch_ms": rp_ser, "mode": "one stream (ZSG_SIDE_STREAM=0): every launch alone on the GPU",
                    "as_timed": {"achieved": round(timed["tflops"] * er, 2), "frac": round(timed["tflops"] * er / PEAK_TF, 4),
                                 "algorithmic_frac": round(timed["tflops"] / PEAK_TF, 4),
                                 "avg_launch_ms": round(timed["ms_per_step"] / timed["launches_per_step"], 5),
                                 "rocprof_avg_launch_ms": rp_avg, "all_kernels_ms_per_step": round(tot, 3),
                                 "mode": "weight-gradient kernels co-running on the side stream"} if timed and timed["tflops"] else None,
                    "all_mfma_kernels": {"achieved": round(exe_all / mfma_ms, 2) if mfma_ms else None,
                                         "frac": round(exe_all / mfma_ms / PEAK_TF, 4) if mfma_ms else None,
                                         "algorithmic_frac": round(flops_all / mfma_ms / PEAK_TF, 4) if mfma_ms else None,
                                         "ms_per_step": round(mfma_ms, 3)},
                    "top_kernels": [{"kernel": r["kernel"], "ms_per_step": round(r["ms_per_step"], 3), "launches_per_step": r["launches_per_step"],
                                     "tflops_executed": round(r["tflops"] * exec_ratio(r["kernel"]), 1) if r["tflops"] else None,
                                     "tflops_algorithmic": round(r["tflops"], 1) if r["tflops"] else None,
                                     "frac": round(r["tflops"] * exec_ratio(r["kernel"]) / PEAK_TF, 3) if r["tflops"] else None,
                                     "gbps": round(r["gbps"], 0) if r["gbps"] else None,
                                     "alg_mb_per_launch": round(r["alg_bytes_per_launch"] / 1e6, 2) if r.get("alg_bytes_per_launch") else None}
                                    for r in iso_rows[:10]]}
        else:
            roof = {"bound": "hbm", "kernel": dom["kernel"], "achieved": round(dom["gbps"] or 0, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round((dom["gbps"] or 0) / 8000.0, 4), "traffic": traffic}
        if a.prof_out and rank == 0:
            with open(a.prof_out, "w") as f:
                json.dump({"as_timed": prof_rows, "isolated": iso_rows}, f, indent=1)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(a.arch, a.img, a.tokens)

    others = None
    headline_default = (a.arch, a.backbone, a.img, a.bs, a.tokens) == ("resnet50", "retina", 300, 16, 20)
    if rank == 0 and world == 1 and not a.force_ddp and (a.other_configs == "on" or (a.other_configs == "auto" and headline_default and tuning_info()["stamp_match"]
                                                                                    and os.environ.get("ZSG_SHIPPED_TUNE", "1") != "0")):
        # BASELINE.json configs[3] and configs[4]'s per-GPU shape, on the shipped table, OUTSIDE the headline's timed region (VERDICT r04
        # item 4: the driver's one command now sees them); ~12 s together
        others = {"what": "the other single-GPU-sized BASELINE.json configurations, timed in this process after the headline legs (same step, "
                          "synthetic batches resident in HBM); not part of `value`"}
        for name, kw in (("configs[3]", dict(arch="resnet50", backbone="ssd_vgg", img=300, bs=32, tokens=a.tokens, warm=4, steps=12)),
                         ("configs[4]_per_gpu", dict(arch="resnet101", backbone="retina", img=600, bs=32, tokens=a.tokens, warm=3, steps=10))):
            try:
                others[name] = time_config(**kw)
            except Exception as e:      # (never lose the headline line to a secondary leg)
                others[name] = {"error": f"{type(e).__name__}: {e}"}

    rccl = None
    if model is not net:               # the data-parallel exchange of this run
        red = next((p.reducer for p in net._plans.values() if p.reducer is not None), None)
        rccl = {"nranks": dist.get_world_size() if dist.is_initialized() else 1, "backend": dist.get_backend() if dist.is_initialized() else None,
                "transport": "zsg_comm_* (RCCL inside libzsg.so)" if model.comm is not None else "torch.distributed all_reduce (ProcessGroupNCCL = RCCL)",
                "buckets_per_step": len(red.buckets) if red else None,
                "allreduce_bytes_per_step": int(sum(b.end - b.start for b in red.buckets) * 4) if red else None,
                "bn_buffer_broadcast_bytes_per_step": int(net._rmv.numel() * 4),
                "exposed_allreduce_ms": round(exposed_ms, 3) if exposed_ms is not None else None,
                "exposed_note": "time the compute stream stands behind the last bucket's collective after the backward's last launch (rank 0; mean per step)",
                "variants": variants,
                "per_rank": per_rank,
                "per_rank_min_max_ms_per_step": [min(per_rank["ms_per_step"]), max(per_rank["ms_per_step"])] if per_rank else None}
    if rank == 0:
        fwd_gf = FWD_GF.get((a.arch, a.img))
        step_frac = (ips / world) * (3 * fwd_gf) * 1e9 / (PEAK_TF * 1e12) if fwd_gf else None
        out = {
            "metric": "train images/sec", "value": round(ips, 2), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "median_ms_per_step": round(median_ms, 3), "settle_steps": settle_n,
            "gpu_clock_mhz": gpu_clock, "host": host, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (img~U[0,1), qvec~N(0,.35), random boxes; random-init weights)",
            "config": {"workload": f"ZSGNet train step, {a.arch + '+FPN' if a.backbone == 'retina' else 'SSD-VGG16'}, {a.img}x{a.img}, per-GPU bs={a.bs}, {a.tokens}-token queries "
                                   f"({config_label(a.arch, a.backbone, a.img, a.bs, world)})", "global_batch": a.bs * world,
                       "parallelism": f"dp{world}", "step": "zero_grad+fwd+loss+bwd(+allreduce)+adam+eval"},
            "step_mfma_frac": round(step_frac, 4) if step_frac else None,
            "step_mfma_frac_note": "whole step: ALGORITHMIC conv FLOPs (3 x forward 2*MAC) / time / fp32-MFMA peak; compare with roofline.ceiling_algorithmic, not with 1",
            "rccl": rccl,
            "final_loss": round(loss_val, 4), "final_acc": acc,
            "forward": fwd, "source_stamp": source_stamp(), "tuning": tuning_info(),
            "roofline": roof, "cpu_baseline": cpu, "other_configs": others,
        }
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()            # (RCCL prints its version banner here: keep the JSON line the LAST line of stdout)
    if rank == 0:
        sys.stdout.flush()
        try:                                    # RCCL's banner sits in the C library's stdout buffer until exit: push it out first
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
