"""CPU-only checks of the host side: config rules, reference parameter names / layouts, the C-ABI library exports,
the bucketed gradient reducer over gloo (world_size 2), CLI plumbing, and the fail-loudly rule (no CPU fallback)."""
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import zsg_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_exports_every_declared_symbol():
    import ctypes
    from zsgnet_pytorch_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "zsg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(zsg_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/zsg.h but not exported by libzsg.so"
        assert name in _lib.SIGNATURES, f"{name} has no Python binding"
    assert sorted(_lib.SIGNATURES) == declared
    assert _lib.lib.zsg_version() == 100 and _lib.lib.zsg_last_error() is not None


def test_config_rules():
    from zsgnet_pytorch_amd.config import get_cfg, ratios_scales, update_from_dict
    cfg = get_cfg()
    assert cfg.bs == cfg["bs"] == 16 and cfg.mdl_to_use == "retina" and cfg.resize_img == [300, 300]
    update_from_dict(cfg, {"bs": "4", "lr": "1e-3", "use_focal": "False", "resize_img": "[600, 600]", "ds_to_use": "flickr30k_c0"})
    assert cfg.bs == 4 and cfg.lr == 1e-3 and cfg.use_focal is False and cfg.resize_img == [600, 600] and cfg.ds_to_use == "flickr30k_c0"
    with pytest.raises(AssertionError):
        update_from_dict(cfg, {"no_such_key": 1})           # extended_config.py:78-85
    with pytest.raises(AssertionError):
        update_from_dict(cfg, {"bs": "abc"})                # type must be kept (extended_config.py:87)
    r, s = ratios_scales(get_cfg())
    ro, so = O.default_ratios_scales()
    assert r == ro and np.array_equal(s, so)
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.bs = 3


@pytest.mark.parametrize("arch,nparams", [("resnet18", None), ("resnet50", 35594349), ("resnet101", None)])
def test_module_names_shapes_and_storage_layout(arch, nparams):
    from zsgnet_pytorch_amd.config import get_cfg
    from zsgnet_pytorch_amd.mdl import get_default_net
    net = get_default_net(9, get_cfg(resnet_arch=arch))
    ref = O.seeded_state_dict(arch, 0)
    sd = net.state_dict()
    assert list(sd.keys()) and set(sd.keys()) == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    if nparams:
        assert sum(p.numel() for p in net.parameters()) == nparams      # 37 643 349 minus torchvision's unused fc
    net.load_state_dict(ref)
    for k, v in ref.items():
        assert torch.equal(net.state_dict()[k], v), k
    # OHWI storage, input channels padded to a multiple of 4 with zeros; the Parameter is a strided OIHW view of it
    raw = net.store.raw("backbone.encoder.conv1.weight").view(64, 7, 7, 4)
    assert torch.equal(raw[..., :3].permute(0, 3, 1, 2), ref["backbone.encoder.conv1.weight"]) and raw[..., 3].abs().max() == 0
    raw = net.store.raw("att_reg_box.0.0.weight").view(256, 3, 3, 516)
    assert torch.equal(raw[..., :514].permute(0, 3, 1, 2), ref["att_reg_box.0.0.weight"]) and raw[..., 514:].abs().max() == 0
    assert torch.equal(net.state_dict()["att_reg_box.5.bias"], torch.tensor([0, 0, 0, 0, -4.0] * 9))
    assert [n for n, _ in net.named_parameters()] == net._param_names
    # DDP-style / torchvision-style checkpoints load (utils.py:489)
    dd = {"module." + k: v for k, v in ref.items()}
    dd["module.backbone.encoder.fc.weight"] = torch.zeros(10, 10)
    net.load_state_dict(dd)
    # parameters stay views of ONE flat buffer after .to()
    net.to("cpu")
    p = dict(net.named_parameters())["backbone.fpn.P6.weight"]
    assert p.data_ptr() >= net.store.flat.data_ptr() and p.data_ptr() < net.store.flat.data_ptr() + net.store.flat.numel() * 4


def test_no_cpu_fallback():
    """The hot path must fail loudly without the MI355X (task rule ③)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from zsgnet_pytorch_amd.config import get_cfg
    from zsgnet_pytorch_amd.mdl import get_default_net
    net = get_default_net(9, get_cfg(resnet_arch="resnet18"))
    bt = O.synthetic_batch(2, 64, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(bt)
    two = get_default_net(9, get_cfg(use_same_atb=False))          # the reference's historical two-head variant (mdl.py:220-225)
    assert set(two.state_dict().keys()) == set(O.seeded_state_dict("resnet50", 0, same_atb=False).keys())
    assert float(two.state_dict()["att_box.5.bias"][0]) == -4.0 and two.state_dict()["reg_box.5.weight"].shape == (36, 256, 3, 3)
    for kw, cin in ((dict(use_lang=False), 256), (dict(use_img=False), 256), (dict(use_lang=False, use_img=False), 2), (dict(do_norm=True), 514)):
        assert get_default_net(9, get_cfg(**kw)).state_dict()["att_reg_box.0.0.weight"].shape == (256, cin, 3, 3)     # mdl.py:196-209
    ssd = get_default_net(9, get_cfg(mdl_to_use="ssd_vgg"))
    assert set(ssd.state_dict().keys()) == set(O.seeded_ssd_state_dict(0).keys())


def test_host_anchor_tables_match_oracle():
    from zsgnet_pytorch_amd import anchors
    r, s = O.default_ratios_scales()
    for fs in (O.feat_sizes_for(300, 300), O.feat_sizes_for(600, 600, True), [(7, 3), (1, 5)]):
        assert np.array_equal(anchors.create_anchors_np(fs, r, s), O.create_anchors(fs, r, s))
    assert np.array_equal(anchors.create_grid((19, 19)).numpy(), O.create_grid(19, 19))


def test_cli_argument_parsing_and_config_1_plumbing():
    """BASELINE configs[0]: refclef, ResNet-18 FPN, bs=2, CPU-only world_size=1 — CLI -> cfg -> modules (no GPU compute)."""
    from zsgnet_pytorch_amd.config import get_cfg, update_from_dict
    from zsgnet_pytorch_amd.main_dist import parse_argv
    from zsgnet_pytorch_amd.mdl import get_default_net
    from zsgnet_pytorch_amd.synth import get_data
    uid, kw = parse_argv(["exp", "--ds_to_use=refclef", "--bs=2", "--resnet_arch", "resnet18", "--only_val"])
    cfg = update_from_dict(get_cfg(), kw)
    assert uid == "exp" and cfg.bs == 2 and cfg.resnet_arch == "resnet18" and cfg.only_val is True
    net = get_default_net(9, cfg)
    assert net.block_kind == "basic" and "backbone.encoder.layer4.1.conv2.weight" in net.state_dict()
    data = get_data(cfg)
    b = next(iter(data.train_dl))
    assert b["img"].shape == (2, 3, 300, 300) and b["qvec"].shape == (2, 20, 300) and b["annot"].shape == (2, 4)
    assert all(v.dtype == torch.float32 for v in b.values())          # collater casts every field to float (dat_loader.py:193)


def test_plan_buckets_cover_flat_buffer_in_readiness_order():
    from zsgnet_pytorch_amd.dist import plan_buckets
    spans = [(0, 100, 90), (100, 60, 80), (160, 400, 70), (560, 40, 10), (600, 8, 95)]
    b = plan_buckets(spans, 150)
    assert sorted((x.start, x.end) for x in b) == [(0, 160), (160, 560), (560, 608)]
    assert [x.ready for x in b] == sorted(x.ready for x in b)
    assert {(x.start, x.end): x.ready for x in b}[(0, 160)] == 90
    # the bucket that completes last is split so that only a small all-reduce is exposed at the end of backward
    enc = [(i * 100, 100, 50 - i) for i in range(12)]            # forward order: the first parameter is written last
    b = plan_buckets(enc, 600, tail_elems=150)
    assert sorted((x.start, x.end) for x in b) == [(0, 200), (200, 600), (600, 1200)] and b[-1].start == 0 and b[-1].ready == 50
    assert sum(x.end - x.start for x in b) == 1200 and [x.ready for x in b] == sorted(x.ready for x in b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reducer_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zsgnet_pytorch_amd.dist import BucketReducer, DistributedDataParallel, plan_buckets, reduce_dict
    torch.manual_seed(rank)
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    launched = []
    spans = [(0, 300, 7), (300, 300, 4), (600, 400, 1)]
    red = BucketReducer(flat, plan_buckets(spans, 100))
    red.run(10, lambda i, j: launched.append((i, j)))
    red.wait()
    expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, expect) and launched == [(0, 2), (2, 5), (5, 8), (8, 10)]

    class Store:
        pass

    class Fake(torch.nn.Module):          # the slice of ZSGNet the wrapper touches
        def __init__(self):
            super().__init__()
            self.store = Store()
            self.store.flat = torch.full((16,), float(rank))
            self.store.grad = torch.zeros(16)
            self._rmv = torch.cat([torch.full((4,), float(rank)), torch.full((4,), float(rank) + 1)])     # means | variances, one buffer
            self._rm, self._rv = self._rmv[:4], self._rmv[4:]
            self._nbt = torch.tensor([rank])

        def forward(self, x):
            return x
    m = Fake()
    ddp = DistributedDataParallel(m)
    ok = ok and float(m.store.flat.sum()) == 0.0 and float(m._rv.sum()) == 4.0 and int(m._nbt) == 0     # rank 0's values everywhere
    m._rm.fill_(float(rank + 5))
    m.train()
    ddp(torch.zeros(1))
    ok = ok and float(m._rm[0]) == 5.0                                                                   # C2: buffers re-broadcast per forward
    # the `comm="native"` control flow of the wrapper and the reducer (zsg_comm_* needs RCCL and a GPU per rank): a communicator
    # object with NativeComm's interface, carried by gloo — parameter / buffer broadcasts and the bucket loop go through comm.*
    class GlooComm:
        def __init__(self):
            self.calls = []

        def all_reduce(self, t):
            self.calls.append("ar")
            dist.all_reduce(t)

        def broadcast(self, t, src=0):
            self.calls.append("bc")
            dist.broadcast(t, src=src)

        def wait(self):
            self.calls.append("wait")

        def close(self):
            self.calls.append("close")
    gc = GlooComm()
    m2 = Fake()
    ddp2 = DistributedDataParallel(m2, comm=gc)
    ok = ok and float(m2.store.flat.sum()) == 0.0 and gc.calls[:2] == ["bc", "bc"]                     # flat weights, BatchNorm buffers
    flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red2 = BucketReducer(flat2, plan_buckets(spans, 100), comm=gc)
    red2.run(10, lambda i, j: None)
    red2.wait()
    ok = ok and torch.equal(flat2, expect) and gc.calls.count("ar") == len(red2.buckets) and gc.calls[-1] == "wait"
    ddp2.close()
    ok = ok and gc.calls[-1] == "close"
    rd = reduce_dict({"a": torch.tensor(float(rank + 1)), "b": torch.tensor(2.0)}, average=True)
    if rank == 0:
        ok = ok and abs(float(rd["a"]) - 1.5) < 1e-6 and abs(float(rd["b"]) - 2.0) < 1e-6
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _tsync_worker(rank, world, port, ret):
    """ranks meet the query-length buckets in DIFFERENT orders (the collater cuts qvec to each rank's own longest query)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from zsgnet_pytorch_amd import ops
    from zsgnet_pytorch_amd.dist import DistributedDataParallel
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class Store:
        pass

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.store = Store()
            self.store.flat, self.store.grad = torch.zeros(8), torch.zeros(8)
            self._rmv = torch.zeros(8)
            self._nbt = torch.tensor([0])
            self._plans, self.lowered = {}, []

        def plan_geometry(self, inp):
            return (2, 96, 96, 20 if inp["qvec"].shape[1] <= 20 else 50)

        def _plan_for(self, B, H, W, T):
            self._plans[(B, H, W, T, self.training)] = True
            self.lowered.append(T)
            ops._TUNE_CACHE[("fake", B, H, W, T)] = 7          # "rank 0 tuned something"

        def forward(self, inp):
            return inp["qvec"].sum()
    m = Fake().train()
    ddp = DistributedDataParallel(m)
    # step:        0   1   2   3          (rank 1 meets bucket 50 two steps before rank 0 does)
    lens = [[12, 18, 31, 40], [33, 15, 9, 44]][rank]
    for T in lens:
        ddp({"qvec": torch.zeros(2, T, 4)})
        x = torch.ones(1)
        dist.all_reduce(x)                                     # the step's gradient collective: must pair up on both ranks
        assert float(x) == world
    m.eval()
    ddp({"qvec": torch.zeros(2, lens[0], 4)})                  # a mode switch is a new (symmetric) key
    ret[rank] = (len(ddp._tuned), dict(ops._TUNE_CACHE), list(m.lowered))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_tuning_sync_is_rank_symmetric_gloo_world2():
    """the tuning broadcast must not depend on the per-rank query-length bucket"""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tsync_worker, args=(2, _free_port(), ret), nprocs=2, join=True)          # (a collective mismatch would hang / raise)
    assert ret[0][0] == 2 and ret[1][0] == 2, "one broadcast per (B, H, W, mode), on every rank"
    assert ret[0][2] == [20, 20] and ret[1][2] == [], "only rank 0 lowers ahead of the broadcast (its own bucket)"
    assert ret[1][1] == ret[0][1] and ret[0][1], "rank 1 runs rank 0's table"


def test_bucket_reducer_and_ddp_wrapper_gloo_world2():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reducer_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_eval_script_prediction_files(tmp_path):
    """eval_script.py:19-56 — pickle format, per-rank merge, each id once, STRICT IoU > thr; box_iou against the oracle."""
    import pickle
    from zsgnet_pytorch_amd import eval_script as E
    rng = np.random.default_rng(0)
    gt = np.sort(rng.uniform(0, 300, (12, 2, 2)), axis=1).transpose(0, 2, 1).reshape(12, 4)        # x1 y1 x2 y2? -> fix below
    gt = np.stack([np.minimum(gt[:, 0], gt[:, 2]), np.minimum(gt[:, 1], gt[:, 3]), np.maximum(gt[:, 0], gt[:, 2]), np.maximum(gt[:, 1], gt[:, 3])], 1)
    pred = gt + rng.normal(0, 25, gt.shape)
    pred[3] = gt[3]                                              # identical box: IoU just below 1
    pred[4] = [0, 0, 1, 1]
    ious = [E.box_iou(p, g) for p, g in zip(pred, gt)]
    ref = [float(O.iou_values(np.float32(p)[None], np.float32(g)[None])[0, 0]) for p, g in zip(pred, gt)]
    assert ious == ref, "fp32 IoU must follow the reference's operation order bit for bit"
    with open(tmp_path / "gt.csv", "w") as f:
        f.write("img_id,bbox,query\n")
        for i, b in enumerate(gt):
            f.write(f'{i}.jpg,"{[float(v) for v in b]}",a thing\n')
    recs = [{"id": float(i), "pred_boxes": [float(v) for v in pred[i]], "pred_scores": 0.5} for i in range(12)]
    with open(tmp_path / "0_preds.pkl", "wb") as f:
        pickle.dump(recs[:7] + recs[:2], f)                    # DDP's padded sampler repeats samples
    with open(tmp_path / "1_preds.pkl", "wb") as f:
        pickle.dump(recs[7:], f)
    with pytest.raises(AssertionError):
        E.evaluate(tmp_path / "preds.pkl", tmp_path / "gt.csv")
    acc, corr, tot = E.evaluate(tmp_path / "preds.pkl", tmp_path / "gt.csv", num_gpus=2)
    assert tot == 12 and corr == sum(v > 0.5 for v in ref) and acc == corr / 12 and (tmp_path / "preds.pkl").exists()
    thr = ref[0]                                                # strict: a box exactly at the threshold is wrong
    _, corr_at, _ = E.evaluate(tmp_path / "preds.pkl", tmp_path / "gt.csv", acc_iou_thresh=thr)
    assert corr_at == sum(v > thr for v in ref)
    assert E.main([str(tmp_path / "preds.pkl"), str(tmp_path / "gt.csv"), "--acc_iou_thresh=0.5"])[2] == 12


def test_pretrained_encoder_key_mapping(tmp_path):
    """a13 (reference mdl.py:411, 415-416): torchvision-layout ResNet files and the reduced-fc VGG16 trunk are mapped onto
    backbone.encoder.* / backbone.encoder.vgg.*; a file that matches nothing raises instead of being ignored."""
    from zsgnet_pytorch_amd.config import get_cfg
    from zsgnet_pytorch_amd import mdl
    g = torch.Generator().manual_seed(0)
    # torchvision-layout resnet18 file: keys without prefix, plus the unused fc.*
    ref = O.seeded_state_dict("resnet18", 5)
    tv = {k[len("backbone.encoder."):]: torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone()
          for k, v in ref.items() if k.startswith("backbone.encoder.")}
    tv["fc.weight"], tv["fc.bias"] = torch.randn(1000, 512), torch.randn(1000)
    f = tmp_path / "resnet18_tv.pth"
    torch.save(tv, f)
    net = mdl.get_default_net(9, get_cfg(resnet_arch="resnet18", pretrained_path=str(f)))
    sd = net.state_dict()
    for k, v in tv.items():
        if not k.startswith("fc."):
            assert torch.equal(sd["backbone.encoder." + k], v), k
    assert not torch.equal(sd["att_reg_box.0.0.weight"], torch.zeros_like(sd["att_reg_box.0.0.weight"]))     # head untouched (random init)
    # DDP-wrapped full checkpoint: passes through
    full = {"model_state_dict": {"module." + k: v.clone() for k, v in ref.items()}}
    f2 = tmp_path / "full.pth"
    torch.save(full, f2)
    net2 = mdl.get_default_net(9, get_cfg(resnet_arch="resnet18", pretrained_path=str(f2)))
    assert all(torch.equal(net2.state_dict()[k], v) for k, v in ref.items())
    # vgg16_reducedfc layout: '0.weight', '0.bias', '2.weight', ...
    ssd = mdl.get_default_net(9, get_cfg(mdl_to_use="ssd_vgg"))
    own = {k: v for k, v in ssd.state_dict().items() if k.startswith("backbone.encoder.vgg.")}
    assert "backbone.encoder.vgg.0.weight" in own and "backbone.encoder.vgg.33.bias" in own
    vgg = {k[len("backbone.encoder.vgg."):]: torch.randn(v.shape, generator=g) for k, v in own.items()}
    f3 = tmp_path / "vgg16_reducedfc.pth"
    torch.save(vgg, f3)
    ssd2 = mdl.get_default_net(9, get_cfg(mdl_to_use="ssd_vgg", pretrained_path=str(f3)))
    for k, v in vgg.items():
        assert torch.equal(ssd2.state_dict()["backbone.encoder.vgg." + k], v), k
    # a resnet file offered to the SSD model matches nothing -> loud failure; so does a wrong shape
    with pytest.raises(ValueError, match="none of its"):
        mdl.get_default_net(9, get_cfg(mdl_to_use="ssd_vgg", pretrained_path=str(f)))
    tv_bad = dict(tv)
    tv_bad["conv1.weight"] = torch.zeros(64, 3, 3, 3)
    f4 = tmp_path / "bad.pth"
    torch.save(tv_bad, f4)
    with pytest.raises(ValueError, match="has shape"):
        mdl.get_default_net(9, get_cfg(resnet_arch="resnet18", pretrained_path=str(f4)))


def test_optimizer_state_roundtrip_and_resume_rules(tmp_path):
    """FusedAdam.load_state_dict restores moments, step AND param_groups; a plain torch Adam state is refused;
    Learner.load_model_dict tolerates a missing file (reference utils.py:443-457) and restores optimizer + scheduler."""
    from zsgnet_pytorch_amd.config import get_cfg
    from zsgnet_pytorch_amd import mdl, optim, trainer, loss, evaluator, config
    cfg = get_cfg(resnet_arch="resnet18", tmp_path=str(tmp_path), resume=True, resume_path=str(tmp_path / "nope.pth"), load_opt=True)
    net = mdl.get_default_net(9, cfg)
    opt = optim.FusedAdam(net, lr=1e-4)
    opt.m.normal_(); opt.v.uniform_(); opt.step_count.fill_(17)
    opt.param_groups[0]["lr"] = 2.5e-5
    sd = opt.state_dict()
    opt2 = optim.FusedAdam(net, lr=1e-4)
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.m, opt.m) and torch.equal(opt2.v, opt.v) and int(opt2.step_count) == 17
    assert opt2.param_groups[0]["lr"] == 2.5e-5
    with pytest.raises(ValueError, match="not a FusedAdam state"):
        opt2.load_state_dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(2))]).state_dict())
    r, s = config.ratios_scales(cfg)
    lf, ev = loss.get_default_loss(r, s, cfg), evaluator.get_default_eval(r, s, cfg)
    lrn = trainer.Learner("t", None, net, lf, cfg, ev, lambda m, lr: optim.FusedAdam(m, lr=lr), device=torch.device("cpu"))
    assert lrn.num_it == 0 and lrn.optimizer is None          # missing resume file: fresh start, no exception
    lrn.prepare_optimizer(1e-4)
    lrn.optimizer.m.fill_(0.5); lrn.optimizer.step_count.fill_(9); lrn.optimizer.param_groups[0]["lr"] = 1e-5
    lrn.num_it, lrn.num_epoch, lrn.best_met = 40, 2, 0.3
    lrn.lr_scheduler.step(0.3)
    lrn.save_model_dict()
    cfg2 = get_cfg(resnet_arch="resnet18", tmp_path=str(tmp_path), resume=True, resume_path=str(lrn.model_file), load_opt=True)
    lrn2 = trainer.Learner("t2", None, mdl.get_default_net(9, cfg2), lf, cfg2, ev, lambda m, lr: optim.FusedAdam(m, lr=lr), device=torch.device("cpu"))
    assert lrn2.num_it == 40 and lrn2.num_epoch == 2 and lrn2.best_met == 0.3
    assert lrn2.optimizer is not None and int(lrn2.optimizer.step_count) == 9 and lrn2.optimizer.param_groups[0]["lr"] == 1e-5
    assert float(lrn2.optimizer.m[0]) == 0.5 and lrn2.lr_scheduler.state_dict()["best"] == lrn.lr_scheduler.state_dict()["best"]


def test_bench_self_launches_n_ranks_gloo():
    """`python bench.py --gpus 2` with no launcher around it must become one (the reference is started by
    `python -m torch.distributed.launch --nproc_per_node=N code/main_dist.py`, main_dist.py:70-77): two ranks come up over
    env:// on 127.0.0.1 at a free port, all-reduce, and rank 0 prints ONE JSON line.  gloo here (no GPU); nccl on the GPU node."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, ZSG_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 2, "backend": "gloo"}


@pytest.mark.parametrize("name,defer", [("fwd", 0), ("bwd", 1)])
def test_lane_schedule_orders_every_cross_stream_edge(name, defer, monkeypatch):
    """ops.Program._schedule (the dry run behind the completion-event launcher): on random lane patterns every side-stream launch is
    ordered behind the latest main-stream launch in front of it, every join behind every side-stream launch in front of it, and an
    event is attached to a launch only if that launch is the latest of its stream (else a marker records it)."""
    from zsgnet_pytorch_amd import ops
    monkeypatch.setattr(ops, "SIDE_DEFER", defer)
    rng = np.random.default_rng(7)
    conv = ops._MAIN_CONVS[0]
    for trial in range(60):
        n = int(rng.integers(3, 60))
        prog = ops.Program(name)
        prog.side_batch = int(rng.integers(1, 4))
        lanes = [int(v) for v in rng.choice([0, 0, 0, 1, 1, 2], size=n)]
        prog.calls = [((conv if rng.random() < 0.5 else (lambda *a: 0)), (), f"c{i}") for i in range(n)]
        prog.lanes = lanes
        for join in (True, False):
            for busy in (False, True):
                sched, nev, busy_out = prog._schedule(0, n, join, busy)
                # simulate two in-order streams
                main_pos, side_pos = -1, -1          # program index of the latest launch issued on each stream (op order)
                ev_main, ev_side = {}, {}            # event -> what it covers
                side_sees_main, main_sees_side = -1, -1
                issued = []
                pre_side = busy                      # side-stream work of an earlier range is outstanding
                for t, *rest in sched:
                    if t == "m":
                        i, k = rest
                        assert lanes[i] != 1
                        if lanes[i] == 2:            # a join launch: everything issued on the side stream so far is visible
                            assert main_sees_side == side_pos and not pre_side, (trial, i)
                        main_pos = i
                        issued.append(i)
                        if k >= 0:
                            ev_main[k] = i
                    elif t == "s":
                        i, k = rest
                        assert lanes[i] == 1
                        assert side_sees_main == main_pos, (trial, i, side_sees_main, main_pos)
                        side_pos = i
                        issued.append(i)
                        if k >= 0:
                            ev_side[k] = i
                    elif t == "rm":
                        ev_main[rest[0]] = main_pos
                    elif t == "rs":
                        ev_side[rest[0]] = side_pos
                        pre_side = False             # (a marker on the side stream also covers the earlier range's work)
                    elif t == "ws":
                        side_sees_main = max(side_sees_main, ev_main[rest[0]])
                    elif t == "wm":
                        main_sees_side = max(main_sees_side, ev_side[rest[0]])
                        if ev_side[rest[0]] == side_pos and side_pos >= 0:
                            pre_side = False         # in-order stream: its latest launch finishing implies everything before it
                assert sorted(issued) == list(range(n))
                if join:
                    assert not busy_out and (side_pos < 0 or main_sees_side == side_pos) and not pre_side
                assert nev == len(set(ev_main) | set(ev_side)) or nev >= len(ev_main) + len(ev_side)


def test_partial_row_counts_and_streaming_kernel_geometry():
    """Host logic of zsg_conv_igemm_partial_rows / ops.pw_cands (no launch): regular tiles write one BatchNorm-partial row per BM
    rows per segment; the filter-resident streaming kernel (tile_hint BM = 32) one per workgroup, and only for dense 1x1 /
    stride-1 geometries whose filter fits the LDS next to the eight wave buffers."""
    import ctypes as C

    import torch
    import zsgnet_pytorch_amd.ops as ops
    from zsgnet_pytorch_amd._lib import lib

    def view(B, H, W, Cc):
        return ops.TView(torch.empty(1), B, Cc, Cc, [ops.Level(0, H, W, H * W * Cc)])

    B, H, W = 16, 75, 75
    rows = B * H * W
    d = ops.fwd_desc(view(B, H, W, 64), view(B, H, W, 256), 64, 256, 1, 1, 0, 1, wC=64)
    assert lib.zsg_conv_igemm_partial_rows(C.byref(d)) == -1                      # heuristic hint: unknown
    for bm in (64, 128):
        d.tile_hint = ops.tile_hint(bm, 64, 1)
        assert lib.zsg_conv_igemm_partial_rows(C.byref(d)) == (rows + bm - 1) // bm
    assert ops.pw_cands(d) == [ops.tile_hint(32, 32, 1), ops.tile_hint(32, 64, 1), ops.tile_hint(32, 128, 1)]
    assert d.tile_hint == ops.tile_hint(128, 64, 1)                               # (left as it was)
    d.tile_hint = ops.tile_hint(32, 128, 1)
    assert ops.igemm_partial_rows(d) == 256                                       # one row per workgroup, one workgroup per CU
    small = ops.fwd_desc(view(1, 5, 5, 64), view(1, 5, 5, 64), 64, 64, 1, 1, 0, 1, wC=64, tile_hint=ops.tile_hint(32, 64, 1))
    assert ops.igemm_partial_rows(small) == 1
    # 256 -> 64: units of 32 / 64 columns only; 3x3, strided, non-dense and too-large-filter geometries are refused
    assert ops.pw_cands(ops.fwd_desc(view(B, H, W, 256), view(B, H, W, 64), 256, 64, 1, 1, 0, 1, wC=256)) == [ops.tile_hint(32, 32, 1), ops.tile_hint(32, 64, 1)]
    assert ops.pw_cands(ops.fwd_desc(view(B, H, W, 64), view(B, H, W, 64), 64, 64, 3, 1, 1, 1, wC=64)) == []
    assert ops.pw_cands(ops.fwd_desc(view(B, 76, 76, 64), view(B, 38, 38, 64), 64, 64, 1, 2, 0, 1, wC=64)) == []
    # a filter that does not fit as a whole is cut into 2 / 4 / 8 panels (column blocks of the grid): 128 -> 512 in 4 panels of 128
    # rows (64 row groups x 4), 512 -> 128 only in 4 panels of 32; 256 -> 1024 would need 16: refused
    d512 = ops.fwd_desc(view(B, 38, 38, 128), view(B, 38, 38, 512), 128, 512, 1, 1, 0, 1, wC=128)
    assert ops.pw_cands(d512) == [ops.tile_hint(32, 32, 1), ops.tile_hint(32, 64, 1), ops.tile_hint(32, 128, 1)]
    d512.tile_hint = ops.tile_hint(32, 128, 1)
    assert ops.igemm_partial_rows(d512) == 64
    assert ops.pw_cands(ops.fwd_desc(view(B, 38, 38, 512), view(B, 38, 38, 128), 512, 128, 1, 1, 0, 1, wC=512)) == [ops.tile_hint(32, 32, 1)]
    assert ops.pw_cands(ops.fwd_desc(view(B, 19, 19, 256), view(B, 19, 19, 1024), 256, 1024, 1, 1, 0, 1, wC=256)) == []
    strided_out = ops.TView(torch.empty(1), B, 64, 128, [ops.Level(0, H, W, H * W * 128)])      # pixel stride 128: still dense rows
    assert ops.pw_cands(ops.fwd_desc(view(B, H, W, 64), strided_out, 64, 64, 1, 1, 0, 1, wC=64)) != []
    dg = ops.dgrad_desc(view(B, H, W, 256), view(B, H, W, 64), 256, 64, 1, 1, 0, 1)
    assert ops.pw_cands(dg) == [ops.tile_hint(32, 32, 1), ops.tile_hint(32, 64, 1)]
