"""CLI counterpart of the reference `code/main_dist.py`:

    python -m zsgnet_pytorch_amd.main_dist <uid> [--key=value ...]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m zsgnet_pytorch_amd.main_dist <uid> --bs=16

Same flags as configs/cfg.json + device/local_rank/do_dist/only_val/only_test (extended_config.py:13-21); an unknown
key or a type mismatch is an assertion (extended_config.py:78-88).  Accepts the legacy launcher's --local_rank as well
as torchrun's LOCAL_RANK.  Data: synthetic batches (cfg.synthetic) — the CSV datasets are not available offline."""
import os
import sys
from functools import partial

import torch

from . import dist as zdist
from .config import get_cfg, ratios_scales, update_from_dict


def parse_argv(argv):
    if not argv or argv[0].startswith("--"):
        raise SystemExit("usage: main_dist.py <uid> [--key=value ...]")
    uid, kw, i = argv[0], {}, 1
    while i < len(argv):
        a = argv[i]
        assert a.startswith("--"), f"unexpected argument {a}"
        if "=" in a:
            k, v = a[2:].split("=", 1)
        else:
            k, v = a[2:], (argv[i + 1] if i + 1 < len(argv) and not argv[i + 1].startswith("--") else "True")
            i += 0 if v == "True" and (i + 1 >= len(argv) or argv[i + 1].startswith("--")) else 1
        kw[k] = v
        i += 1
    return uid, kw


def learner_init(uid: str, cfg):
    """main_dist.py:18-54"""
    from .evaluator import get_default_eval
    from .loss import get_default_loss
    from .mdl import get_default_net
    from .optim import FusedAdam
    from .trainer import Learner
    device = torch.device(cfg["device"])
    if cfg["synthetic"]:
        from .synth import get_data
        data = get_data(cfg, zdist.get_rank())
    else:                                   # CSV / image datasets of cfg.ds_to_use (dat_loader.py:233-257)
        from .dat_loader import get_data
        data = get_data(cfg)
    ratios, scales = ratios_scales(cfg)
    num_anchors = len(ratios) * len(scales)
    mdl = get_default_net(num_anchors=num_anchors, cfg=cfg)
    mdl.to(device)
    if cfg["do_dist"]:
        mdl = zdist.DistributedDataParallel(mdl, device_ids=[cfg["local_rank"]], output_device=cfg["local_rank"],
                                            broadcast_buffers=True, find_unused_parameters=True)
    loss_fn = get_default_loss(ratios, scales, cfg)
    eval_fn = get_default_eval(ratios, scales, cfg)
    opt_fn = partial(FusedAdam, betas=(0.9, 0.99))
    return Learner(uid=uid, data=data, mdl=mdl, loss_fn=loss_fn, opt_fn=opt_fn, eval_fn=eval_fn, device=device, cfg=cfg)


def main_dist(uid: str, **kwargs):
    """main_dist.py:57-97"""
    cfg = get_cfg()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "local_rank" in kwargs or world > 1:
        local = int(kwargs.pop("local_rank", os.environ.get("LOCAL_RANK", "0")))
        cfg["local_rank"] = local
        if world > 1:
            cfg["do_dist"] = True
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
            zdist.init_process_group_from_env()
            zdist.synchronize()
    cfg["num_gpus"] = max(1, world)
    cfg = update_from_dict(cfg, kwargs)
    cfg.freeze()
    learn = learner_init(uid, cfg)
    if not (cfg["only_val"] or cfg["only_test"]):
        learn.fit(epochs=cfg["epochs"], lr=cfg["lr"])
    else:
        if cfg["only_val"]:
            learn.testing(learn.data.valid_dl)
        if cfg["only_test"]:
            learn.testing(learn.data.test_dl)
    return learn


if __name__ == "__main__":
    _uid, _kw = parse_argv(sys.argv[1:])
    main_dist(_uid, **_kw)
