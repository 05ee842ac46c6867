"""Developer tool (GPU box): build the shipped tuning table zsgnet-pytorch_amd/tuning/gfx950.json.
Lowers (= autotunes, median of ZSG_TUNE_ROUNDS interleaved samples per candidate) the training and eval plans of the BASELINE.json
configurations in a FRESH tuning state and writes every choice with the sha256 stamp of the kernel sources.
usage: ZSG_SHIPPED_TUNE=0 python tools/make_tuning_table.py [out.json] [--seed cache.json] [--no-refine] [--verbose] [configs: r50 r18 ssd r101 ...]
After the single-launch tuning of a configuration's training plan, the near-ties are re-ranked INSIDE the step (ZSGNet.refine_tuning ->
ops.refine_in_step: both streams, the launch's real neighbours), so the table does not depend on which of two equal-looking tiles the
single-launch median happened to prefer (round 5: 1.1 % of the step between fresh tunings).
--seed: start from the choices of a tuning cache (ZSG_TUNE_CACHE format) instead of an empty state (a development aid: round 5 seeded the
table by hand from the fastest of six fresh tunings; the in-step refinement replaced that procedure)."""
import json
import os
import sys

os.environ["ZSG_SHIPPED_TUNE"] = "0"
os.environ.setdefault("ZSG_TUNE_ROUNDS", "9")          # the table is made once: more interleaved samples per candidate than a run-time tuning takes
os.environ.pop("ZSG_TUNE_CACHE", None)
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import config, loss, mdl, ops
from zsgnet_pytorch_amd.synth import synthetic_batch

CONFIGS = {
    "r50": dict(arch="resnet50", B=16, img=300, backbone="retina"),     # configs[1] / [2] per-GPU shape (the headline)
    "r18": dict(arch="resnet18", B=2, img=300, backbone="retina"),      # configs[0] shape on the GPU
    "ssd": dict(arch="resnet50", B=32, img=300, backbone="ssd_vgg"),    # configs[3]
    "r101": dict(arch="resnet101", B=32, img=600, backbone="retina"),   # configs[4] per-GPU shape
}


def lower(arch, B, img, backbone):
    cfg = config.get_cfg(resnet_arch=arch, bs=B, resize_img=[img, img], mdl_to_use=backbone)      # (as bench.py builds it)
    net = mdl.get_default_net(9, cfg).to("cuda")
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    bt = {k: v.cuda() for k, v in synthetic_batch(B, img, img, seed=1).items()}
    net.train()
    for _ in range(2):
        lf(net(bt), bt)["loss"].backward()
        for p in net.parameters():
            p.grad = None
    if "--no-refine" not in sys.argv:
        # the tuner's near-ties re-ranked inside the real two-stream step (ops.refine_in_step) — what tools/best_of_tunings.sh +
        # refine_tuning.py did by hand in round 5
        res = net.refine_tuning(bt, log=(print if "--verbose" in sys.argv else None))
        print(f"  in-step refinement: {res}", flush=True)
    net.eval()
    with torch.no_grad():
        net(bt)
    torch.cuda.synchronize()
    del net, bt
    torch.cuda.empty_cache()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".json") else ops.SHIPPED_TABLE
    names = [a for a in sys.argv[1:] if a in CONFIGS] or ["r50", "r18"]
    seed = sys.argv[sys.argv.index("--seed") + 1] if "--seed" in sys.argv else None
    if seed:
        print(f"seeded with {ops.load_tune_cache(seed)} choices from {seed}", flush=True)
    for n in names:
        n0 = len(ops._TUNE_CACHE)
        lower(**CONFIGS[n])
        print(f"{n}: {len(ops._TUNE_CACHE) - n0} launch shapes tuned", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        json.dump({"source_stamp": ops.source_stamp(), "device": torch.cuda.get_device_name(0), "tune_rounds": ops.TUNE_ROUNDS,
                   "configs": names, "seeded": bool(seed), "entries": {repr(k): v for k, v in sorted(ops._TUNE_CACHE.items(), key=lambda kv: repr(kv[0]))}}, f, indent=0)
    print(f"wrote {out}: {len(ops._TUNE_CACHE)} entries, stamp {ops.source_stamp()}")


if __name__ == "__main__":
    main()
