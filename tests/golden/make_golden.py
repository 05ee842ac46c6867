"""Generate golden vectors by IMPORTING THE REFERENCE (container-only; /root/reference never travels).

Run from the repo root:  python tests/golden/make_golden.py
Writes tests/golden/*.npz (data only: inputs + the reference's outputs).  The reference tree is imported with
third-party stubs (torchvision / yacs / spacy / fire / fastprogress / tensorboard are absent here)
and with the documented overrides: CPU anchors (anchors.py:66 defaults to 'cuda'), fp32 anchor regime
(pre-seeded ``.anchs``), recorded LSTM initial state (mdl.py:279-294 draws a random one every call).
"""
import sys
sys.dont_write_bytecode = True   # never write __pycache__ into /root/reference
import functools
import hashlib
import os
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle import zsg_oracle as O   # only for the seed-only weight / batch generators


def import_reference():
    os.chdir(REF)
    sys.path.insert(0, os.path.join(REF, "code"))

    class CN(dict):
        def __init__(self, d=None, new_allowed=False):
            super().__init__()
            for k, v in (d or {}).items():
                self[k] = CN(v) if isinstance(v, dict) and not isinstance(v, CN) else v

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def freeze(self):
            pass

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    yc = mod("yacs.config", CfgNode=CN)
    mod("yacs", config=yc)
    mod("spacy", load=lambda name: None)
    mod("fire", Fire=lambda f: None)
    fp = mod("fastprogress.fastprogress", master_bar=None, progress_bar=None)
    mod("fastprogress", fastprogress=fp)
    mod("torch.utils.tensorboard", SummaryWriter=object)
    import fpn_resnet

    def resnet50(pretrained=False):
        return fpn_resnet.ResNet(1, fpn_resnet.Bottleneck, [3, 4, 6, 3])
    tvm = mod("torchvision.models", resnet50=resnet50)
    tvt_f = mod("torchvision.transforms.functional")
    tvt = mod("torchvision.transforms", functional=tvt_f)
    mod("torchvision", models=tvm, transforms=tvt)
    import anchors
    import evaluator
    import loss
    import mdl
    import ssd_vgg
    from extended_config import cfg
    os.chdir(REPO)
    cfg.device = "cpu"
    return dict(anchors=anchors, evaluator=evaluator, loss=loss, mdl=mdl, fpn_resnet=fpn_resnet, ssd_vgg=ssd_vgg, cfg=cfg)


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]     # e.g. `make_golden.py g12_` rewrites only the matching files


def save(name, **arrs):
    if ONLY and not any(name.startswith(o) for o in ONLY):
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KB")


def adversarial_boxes(rng, n, anchors32):
    """random + degenerate / full-image / anchor-identical / tie-heavy boxes (y1x1y2x2 in [-1,1])."""
    c = rng.uniform(-0.6, 0.6, (n, 2))
    s = rng.uniform(0.1, 0.9, (n, 2))
    b = np.clip(np.concatenate([c - s / 2, c + s / 2], 1), -1, 1).astype(np.float32)
    extra = [
        [-1, -1, 1, 1], [0, 0, 0, 0], [0.25, 0.25, 0.25, 0.75], [-1, -1, -0.99, -0.99],
        [-0.5, -0.5, 0.5, 0.5], [-0.001, -0.001, 0.001, 0.001], [0.0, -1.0, 1.0, 1.0],
    ]
    ids = rng.integers(0, anchors32.shape[0], 9)
    extra += [np.clip(anchors32[i], -1, 1).tolist() for i in ids]          # anchor-identical (clipped)
    extra += [anchors32[i].tolist() for i in ids[:3]]                      # anchor-identical (unclipped)
    return np.concatenate([b, np.array(extra, dtype=np.float32)], 0)


def gen_resize(cfg):
    """G14: the reference loader's image path at photo-like down-scaling ratios (dat_loader.py:98-146: PIL.Image.open -> convert("RGB")
    -> resize(resize_img), PIL's default filter -> pil2tensor / 255): raw uint8 images of several sizes and what ImgQuDataset makes
    of them, as uint8 (the float image x 255 is integral).  Pins oracle.pil_resize_u8 and the HIP kernel zsg_resize_u8."""
    import re
    import tempfile
    import PIL.Image
    rng = np.random.default_rng(14)
    sizes = {"p.png": (125, 167), "q.png": (111, 160), "r.png": (160, 213), "s.png": (67, 50), "t.png": (100, 100), "u.png": (100, 151), "v.png": (240, 90)}
    imgs = {}
    for k, (h, w) in sizes.items():
        # smooth structure + noise + saturated patches: exercises the negative bicubic lobes and the clipping at 0 / 255
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127 + 100 * np.sin(yy / 7.0)[..., None] * np.cos(xx / 5.0)[..., None] * np.array([1.0, 0.7, -0.8])
        a = np.clip(base + rng.normal(0, 30, (h, w, 3)), 0, 255)
        a[h // 4:h // 2, w // 3:w // 2] = 255
        a[h // 2:h // 2 + 9, : w // 4] = 0
        imgs[k] = a.astype(np.uint8)

    def fake_nlp(text):
        class Tok:
            def __init__(self, t):
                self.text, self.vector = t, np.zeros(300, np.float32)
        return [Tok(t) for t in re.findall(r"\w+|[^\w\s]", str(text))]
    with tempfile.TemporaryDirectory() as td:
        for k, v in imgs.items():
            PIL.Image.fromarray(v).save(os.path.join(td, k))
        with open(os.path.join(td, "d.csv"), "w") as f:
            f.write("img_id,bbox,query\n")
            for k in imgs:
                f.write(f'{k},"[1, 2, 30, 40]","a thing"\n')
        os.chdir(REF)
        import dat_loader as DL
        os.chdir(REPO)
        DL.nlp = fake_nlp
        if not hasattr(np, "float_"):
            np.float_ = np.float64
        c3 = cfg.__class__(dict(cfg))
        c3.resize_img = [100, 100]
        c3.ds_info = cfg.__class__({"refclef": cfg.__class__({"img_dir": td})})
        ds = DL.ImgQuDataset(c3, os.path.join(td, "d.csv"), "refclef")
        items = [ds[i] for i in range(len(ds))]
    d = dict(resize_img=np.array([100, 100]), names=np.array(list(imgs)))
    for i, (k, v) in enumerate(imgs.items()):
        f = items[i]["img"].numpy().astype(np.float64) * 255.0              # [3, H, W] in [0, 1]
        u8 = np.rint(f)
        assert np.abs(f - u8).max() < 1e-3
        d["raw_" + k[0]] = v
        d["out_" + k[0]] = u8.astype(np.uint8).transpose(1, 2, 0)
    save("g14_resize", **d)


def gen_learnable(R, S=128, B=16, steps=320, lr=5e-4, decay_at=260):
    """G15: the Acc@IoU0.5 proxy's REFERENCE side.  The reference network (mdl.py get_default_net), loss (loss.py) and evaluator
    (evaluator.py) trained on the CPU with torch.optim.Adam(betas=(0.9, 0.99)) (main_dist.py:50) on a task it can learn —
    O.learnable_batch: the annotated box is a bright rectangle — from a seeded start, a fresh batch and fresh LSTM start states
    every step; then eval mode on 256 held-out samples.  Recorded: every step's loss and the Acc@IoU0.5 hits per held-out batch.
    tests/test_gpu_fullshape.py trains the HIP model on the same stream of batches and must learn the same thing."""
    A, L, E, M, cfg = R["anchors"], R["loss"], R["evaluator"], R["mdl"], R["cfg"]
    ratios = eval(cfg["ratios"], {})
    scales = cfg["scale_factor"] * np.array(eval(cfg["scales"], {}))
    cpu = torch.device("cpu")
    sd = O.seeded_state_dict("resnet50", seed=3)
    net = M.get_default_net(num_anchors=9, cfg=cfg)
    net.load_state_dict(sd, strict=False)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.99))
    lf = L.get_default_loss(ratios, scales, cfg)
    ev = E.get_default_eval(ratios, scales, cfg)
    gq = torch.Generator().manual_seed(8)
    losses = []
    for it in range(steps):
        if it == decay_at:              # the last steps at lr / 10: the weights settle and the BatchNorm running statistics catch up with them
            for grp in opt.param_groups:
                grp["lr"] = lr * 0.1
        bt = O.learnable_batch(B, S, seed=100 + it)
        h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
        net.lstm_init_hidden = lambda bs, h0=h0, c0=c0: (h0, c0)
        out = net(bt)
        if it == 0:
            fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
            anc = A.create_anchors(fs, ratios, scales, device=cpu).float()
            lf.anchs = anc
            ev.anchs = anc
        ls = lf(out, bt)
        opt.zero_grad()
        ls["loss"].backward()
        opt.step()
        losses.append(float(ls["loss"].item()))
        if it % 20 == 0:
            print(f"  g15 step {it}: loss {losses[-1]:.4f}", flush=True)
    net.eval()
    hits = []
    with torch.no_grad():
        for bi in range(16):
            bt = O.learnable_batch(16, S, seed=9000 + bi)
            h0, c0 = torch.randn(2, 16, 128, generator=gq), torch.randn(2, 16, 128, generator=gq)
            net.lstm_init_hidden = lambda bs, h0=h0, c0=c0: (h0, c0)
            hits.append(float(ev(net(bt), bt)["Acc"].item()) * 16)
    print(f"  g15: loss {losses[0]:.3f} -> {np.mean(losses[-10:]):.3f}; held-out Acc@IoU0.5 hits {sum(hits):.0f}/256")
    save("g15_learnable", losses=np.array(losses, np.float64), hits=np.array(hits, np.float64), S=np.array([S]), B=np.array([B]),
         steps=np.array([steps]), lr=np.array([lr]), decay_at=np.array([decay_at]), seed=np.array([3]), feat_sizes=np.array(fs))


def gen_trajectory(R, S=300, B=16, steps=12, lr=1e-4):
    """G16 (round 6): the REFERENCE's own first 12 optimisation steps at the configs[1] shape (ResNet-50 FPN, 300x300, B=16; mdl.py /
    loss.py + torch.optim.Adam(lr 1e-4, betas (0.9, 0.99)) as main_dist.py:50) from a seeded start, a fresh synthetic batch and fresh
    LSTM start states every step.  Recorded: every step's loss and three BatchNorm running statistics after the last step.
    tests/test_gpu_fullshape.py::test_training_trajectory_and_eval_argmax_agreement steps the HIP model through the same stream of
    batches (until round 5 it stepped the CPU oracle beside it inside the GPU test: 100 s of the GPU suite)."""
    A, L, M, cfg = R["anchors"], R["loss"], R["mdl"], R["cfg"]
    ratios = eval(cfg["ratios"], {})
    scales = cfg["scale_factor"] * np.array(eval(cfg["scales"], {}))
    sd = O.seeded_state_dict("resnet50", seed=17)
    net = M.get_default_net(num_anchors=9, cfg=cfg)
    net.load_state_dict(sd, strict=False)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(0.9, 0.99))
    lf = L.get_default_loss(ratios, scales, cfg)
    gq = torch.Generator().manual_seed(8)
    losses = []
    for it in range(steps):
        bt = O.synthetic_batch(B, S, S, seed=500 + it)
        h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
        net.lstm_init_hidden = lambda bs, h0=h0, c0=c0: (h0, c0)
        out = net(bt)
        if it == 0:
            fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
            lf.anchs = A.create_anchors(fs, ratios, scales, device=torch.device("cpu")).float()
        ls = lf(out, bt)
        opt.zero_grad()
        ls["loss"].mean().backward()
        opt.step()
        losses.append(float(ls["loss"].item()))
        print(f"  g16 step {it}: loss {losses[-1]:.4f}", flush=True)
    st = net.state_dict()
    keys = ("backbone.encoder.bn1.running_mean", "backbone.encoder.layer2.3.bn3.running_var", "backbone.encoder.layer4.2.bn3.running_mean")
    save("g16_trajectory", losses=np.array(losses, np.float64), seed=np.array([17]), batch_seed0=np.array([500]), hc_seed=np.array([8]),
         lr=np.array([lr]), rm_bn1=st[keys[0]].numpy(), rv_l2=st[keys[1]].numpy(), rm_l4=st[keys[2]].numpy())


def main():
    R = import_reference()
    A, L, E, M, cfg = R["anchors"], R["loss"], R["evaluator"], R["mdl"], R["cfg"]
    if ONLY and all(o.startswith(("g14", "g15", "g16")) for o in ONLY):
        if any(o.startswith("g14") for o in ONLY):
            gen_resize(cfg)
        if any(o.startswith("g15") for o in ONLY):
            gen_learnable(R)
        if any(o.startswith("g16") for o in ONLY):
            gen_trajectory(R)
        return
    ratios = eval(cfg["ratios"], {})
    scales = cfg["scale_factor"] * np.array(eval(cfg["scales"], {}))
    cpu = torch.device("cpu")
    rng = np.random.default_rng(20260928)

    # ---- G1 create_grid ---------------------------------------------------------------------
    g1 = {}
    for hw in [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1), (1, 7), (75, 75), (4, 9)]:
        g1[f"g_{hw[0]}_{hw[1]}"] = A.create_grid(hw).numpy()
    save("g1_grid", **g1)

    # ---- G2 create_anchors ------------------------------------------------------------------
    fs300 = [(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)]
    fs600 = [(38, 38), (19, 19), (10, 10), (5, 5)]
    a300 = A.create_anchors(fs300, ratios, scales, device=cpu)
    a600 = A.create_anchors(fs600, ratios, scales, device=cpu)
    assert a300.dtype == torch.float64 and a300.shape == (17460, 4) and a600.shape == (17370, 4)
    samp = rng.integers(0, 17370, 64)
    save("g2_anchors", a300_f32=a300.float().numpy(), a300_sha256=np.frombuffer(
        hashlib.sha256(a300.numpy().tobytes()).digest(), dtype=np.uint8),
        a600_sha256=np.frombuffer(hashlib.sha256(a600.numpy().tobytes()).digest(), dtype=np.uint8),
        sample_ids=samp, a300_f64_rows=a300.numpy()[samp], a600_f64_rows=a600.numpy()[samp],
        ratios=np.array(ratios), scales=scales)
    anc32 = a300.float()

    # ---- G3 IoU / argmax / mask -------------------------------------------------------------
    boxes = adversarial_boxes(rng, 237, anc32.numpy())
    iou = A.IoU_values(torch.from_numpy(boxes), anc32)
    mx, am = iou.max(1)
    pos = (iou > 0.6)
    rows, cols = torch.nonzero(pos, as_tuple=True)
    matches = A.simple_match_anchors(anc32, torch.from_numpy(boxes), match_thr=0.6)
    assert bool(((matches >= 0) == pos).all())
    srow = rng.integers(0, boxes.shape[0], 8)
    save("g3_iou", boxes=boxes, argmax=am.numpy().astype(np.int32), maxval=mx.numpy(),
         pos_rows=rows.numpy().astype(np.int32), pos_cols=cols.numpy().astype(np.int32),
         sample_rows=srow, sample_iou=iou.numpy()[srow][:, ::7])

    # ---- G4 encode / decode -----------------------------------------------------------------
    small_fs = [(5, 5), (3, 3), (1, 1)]
    anc_s = A.create_anchors(small_fs, ratios, scales, device=cpu).float()      # A = 315
    bx = torch.from_numpy(boxes[:12].copy())
    enc = A.bbox_to_reg_params(anc_s, bx)
    regs = torch.from_numpy(rng.normal(0, 0.7, (12, anc_s.shape[0], 4)).astype(np.float32))
    dec = A.reg_params_to_bbox(anc_s, regs)
    save("g4_codec", anchors=anc_s.numpy(), boxes=bx.numpy(), enc=enc.numpy(), regs=regs.numpy(), dec=dec.numpy())

    # ---- G5/G6 loss + evaluator -------------------------------------------------------------
    def run_loss_eval(tag, anc, B, flags=None, nan_case=False, full_out=True):
        c = type(cfg)(dict(cfg))
        for k, v in (flags or {}).items():
            c[k] = v
        lf = L.get_default_loss(ratios, scales, c)
        ev = E.get_default_eval(ratios, scales, c)
        lf.anchs = anc
        ev.anchs = anc
        Aa = anc.shape[0]
        bt = O.synthetic_batch(B, 32, 32, seed=77 + B)
        annot = bt["annot"]
        if nan_case:
            annot = annot.clone()
            annot[0] = torch.tensor([0.3, 0.3, 0.3, 0.3])     # zero-area box -> log(0) -> NaN branch of loss.py:128
        att = torch.from_numpy(rng.normal(-3.0, 1.5, (B, Aa, 1)).astype(np.float32)).requires_grad_()
        bbx = torch.from_numpy(rng.normal(0, 0.6, (B, Aa, 4)).astype(np.float32)).requires_grad_()
        out = dict(att_out=att, bbx_out=bbx, feat_sizes=torch.zeros(1, 2).long(), num_f_out=torch.tensor([1]))
        inp = dict(annot=annot, idxs=bt["idxs"], img_size=bt["img_size"])
        ls = lf(out, inp)
        ls["loss"].backward()
        with torch.no_grad():
            em = ev(out, inp)
            att_s = torch.sigmoid(att).squeeze(-1)
            top2 = torch.topk(att_s, 2, dim=1)[0]
        d = dict(att=att.detach().numpy(), bbx=bbx.detach().numpy(), annot=annot.numpy(), img_size=bt["img_size"].numpy(),
                 loss=np.float64(ls["loss"].item()), cls_ls=np.float64(ls["cls_ls"].item()), box_ls=np.float64(ls["box_ls"].item()),
                 Acc=em["Acc"].numpy(), MaxPos=em["MaxPos"].numpy(), pred_boxes=em["pred_boxes"].numpy(),
                 pred_scores=em["pred_scores"].numpy(), score_gap=(top2[:, 0] - top2[:, 1]).numpy(),
                 pred_ids=att_s.max(1)[1].numpy())
        if att.grad is not None and full_out:
            d["g_att"] = att.grad.numpy()
            d["g_bbx"] = bbx.grad.numpy()
        elif att.grad is not None:
            d["g_att_s"] = att.grad.numpy()[:, ::11]
            d["g_bbx_s"] = bbx.grad.numpy()[:, ::11]
            d["g_att_abs_sum"] = np.float64(att.grad.abs().double().sum().item())
            d["g_bbx_abs_sum"] = np.float64(bbx.grad.abs().double().sum().item())
            d.pop("att"), d.pop("bbx")
            d["att_seed_note"] = np.array([0])
        return d

    g5 = {}
    for B in (1, 2, 16):
        for k, v in run_loss_eval(f"b{B}", anc_s, B).items():
            g5[f"b{B}_{k}"] = v
    for k, v in run_loss_eval("nomulti", anc_s, 4, dict(use_multi=False)).items():
        g5[f"nomulti_{k}"] = v
    for k, v in run_loss_eval("nofocal", anc_s, 4, dict(use_focal=False)).items():
        g5[f"nofocal_{k}"] = v
    for k, v in run_loss_eval("softmax", anc_s, 4, dict(use_multi=False, use_softmax=True)).items():
        g5[f"softmax_{k}"] = v
    for k, v in run_loss_eval("nan", anc_s, 2, nan_case=True).items():
        g5[f"nan_{k}"] = v
    g5["anchors"] = anc_s.numpy()
    save("g5_loss_eval_small", **g5)

    # one full-A case (inputs regenerated from a seed by the test to keep the file small)
    torch.manual_seed(5)
    B = 2
    bt = O.synthetic_batch(B, 32, 32, seed=99)
    gfull = torch.Generator().manual_seed(4242)
    att = (torch.randn(B, 17460, 1, generator=gfull) * 1.5 - 3.0).requires_grad_()
    bbx = (torch.randn(B, 17460, 4, generator=gfull) * 0.6).requires_grad_()
    lf = L.get_default_loss(ratios, scales, cfg)
    ev = E.get_default_eval(ratios, scales, cfg)
    lf.anchs = anc32
    ev.anchs = anc32
    out = dict(att_out=att, bbx_out=bbx, feat_sizes=torch.zeros(1, 2).long(), num_f_out=torch.tensor([1]))
    inp = dict(annot=bt["annot"], idxs=bt["idxs"], img_size=bt["img_size"])
    ls = lf(out, inp)
    ls["loss"].backward()
    em = ev(out, inp)
    save("g5_loss_eval_full", annot=bt["annot"].numpy(), img_size=bt["img_size"].numpy(),
         loss=np.float64(ls["loss"].item()), cls_ls=np.float64(ls["cls_ls"].item()), box_ls=np.float64(ls["box_ls"].item()),
         g_att_s=att.grad.numpy()[:, ::13], g_bbx_s=bbx.grad.numpy()[:, ::13],
         g_att_abs_sum=np.float64(att.grad.abs().double().sum().item()),
         g_bbx_abs_sum=np.float64(bbx.grad.abs().double().sum().item()),
         Acc=em["Acc"].detach().numpy(), MaxPos=em["MaxPos"].detach().numpy(), pred_boxes=em["pred_boxes"].detach().numpy(),
         pred_scores=em["pred_scores"].detach().numpy(), gen_seed=np.array([4242]))

    # ---- G7 apply_lstm ----------------------------------------------------------------------
    sd = O.seeded_state_dict("resnet50", seed=3)
    net = M.get_default_net(num_anchors=9, cfg=cfg)
    missing = net.load_state_dict(sd, strict=False)
    assert all(k.startswith("backbone.encoder.fpn.") for k in missing.missing_keys), missing.missing_keys
    assert not missing.unexpected_keys
    Bq = 5
    gq = torch.Generator().manual_seed(11)
    qvec = torch.randn(Bq, 20, 300, generator=gq) * 0.35
    qlens = torch.tensor([7.0, 20.0, 7.0, 1.0, 12.0])
    h0 = torch.randn(2, Bq, 128, generator=gq)
    c0 = torch.randn(2, Bq, 128, generator=gq)
    net.lstm_init_hidden = lambda bs: (h0, c0)
    net.train()
    we = net.apply_lstm(qvec[:, :20].contiguous(), qlens, 20)
    gw = torch.randn(Bq, 256, generator=gq)
    net.zero_grad()
    (we * gw).sum().backward()
    lg = {k.replace("lstm.", "").replace(".", "_"): (p.grad.numpy() if p.grad.dim() == 1 else p.grad.numpy()[::4, ::3])
          for k, p in net.named_parameters() if k.startswith("lstm.")}
    perm = qlens.sort(0, descending=True)[1]
    save("g7_lstm", qvec=qvec.numpy(), qlens=qlens.numpy(), h0=h0.numpy(), c0=c0.numpy(), we=we.detach().numpy(),
         gw=gw.numpy(), perm=perm.numpy(), seed=np.array([3]), **{"grad_" + k: v for k, v in lg.items()})

    # ---- G8 FPN + Bottleneck ----------------------------------------------------------------
    FR = R["fpn_resnet"]
    fpn = net.backbone.fpn
    gf = torch.Generator().manual_seed(21)
    c3 = torch.randn(1, 512, 38, 38, generator=gf)
    c4 = torch.randn(1, 1024, 19, 19, generator=gf)
    c5 = torch.randn(1, 2048, 10, 10, generator=gf)
    with torch.no_grad():
        outs = fpn([c3, c4, c5])
    save("g8_fpn", seed=np.array([3]), in_seed=np.array([21]),
         **{f"p{i}": (o.numpy() if o.numel() < 40000 else o.numpy()[:, ::8, ::3, ::3]) for i, o in enumerate(outs)})
    blk = net.backbone.encoder.layer2[0]     # stride-2 bottleneck with downsample
    blk.train()
    xb = torch.randn(2, 256, 12, 12, generator=gf).requires_grad_()
    for m_ in blk.modules():
        if isinstance(m_, torch.nn.BatchNorm2d):
            m_.running_mean.zero_()
            m_.running_var.fill_(1)
    yb = blk(xb)
    gy = torch.randn(yb.shape, generator=gf)
    blk.zero_grad()
    (yb * gy).sum().backward()
    save("g8_bottleneck", seed=np.array([3]), x=xb.detach().numpy(), gy=gy.numpy(), y=yb.detach().numpy(), gx=xb.grad.numpy(),
         g_conv2=blk.conv2.weight.grad.numpy()[::2, ::2], g_bn3_w=blk.bn3.weight.grad.numpy(), g_bn3_b=blk.bn3.bias.grad.numpy(),
         g_ds=blk.downsample[0].weight.grad.numpy()[::4, ::4], rm_bn2=blk.bn2.running_mean.numpy(), rv_bn2=blk.bn2.running_var.numpy())

    # ---- G8b FPN 600x600 branch (fpn_resnet.py:173-174: p3 is computed and dropped, four levels) -----------------------
    cfg600 = type(fpn.cfg)(dict(fpn.cfg)) if not isinstance(fpn.cfg, dict) else dict(fpn.cfg)
    cfg600["resize_img"] = [600, 600]
    keep_cfg = fpn.cfg
    fpn.cfg = cfg600
    g6 = torch.Generator().manual_seed(23)
    c3b = torch.randn(1, 512, 75, 75, generator=g6)
    c4b = torch.randn(1, 1024, 38, 38, generator=g6)
    c5b = torch.randn(1, 2048, 19, 19, generator=g6)
    with torch.no_grad():
        outs6 = fpn([c3b, c4b, c5b])
    fpn.cfg = keep_cfg
    assert len(outs6) == 4
    save("g8_fpn600", seed=np.array([3]), in_seed=np.array([23]), sizes=np.array([list(o.shape[2:]) for o in outs6]),
         **{f"p{i}": (o.numpy() if o.numel() < 40000 else o.numpy()[:, ::8, ::3, ::3]) for i, o in enumerate(outs6)})

    # ---- G8c BasicBlock encoder (fpn_resnet.py:26-58; ResNet(1, BasicBlock, [2,2,2,2]) = the resnet18 stand-in) ---------
    sd18 = O.seeded_state_dict("resnet18", seed=9)
    enc18 = FR.ResNet(1, FR.BasicBlock, [2, 2, 2, 2])
    pre18 = "backbone.encoder."
    res18 = enc18.load_state_dict({k[len(pre18):]: v for k, v in sd18.items() if k.startswith(pre18)}, strict=False)
    assert not res18.unexpected_keys and all(k.startswith("fpn.") or k.startswith("fc.") for k in res18.missing_keys), res18
    enc18.train()
    g8 = torch.Generator().manual_seed(29)
    blk18 = enc18.layer2[0]                   # stride-2 BasicBlock with downsample
    xk = torch.randn(2, 64, 14, 14, generator=g8).requires_grad_()
    yk = blk18(xk)
    gyk = torch.randn(yk.shape, generator=g8)
    enc18.zero_grad()
    (yk * gyk).sum().backward()
    d18 = dict(seed=np.array([9]), x=xk.detach().numpy(), gy=gyk.numpy(), y=yk.detach().numpy(), gx=xk.grad.numpy(),
               g_conv1=blk18.conv1.weight.grad.numpy()[::2, ::2], g_conv2=blk18.conv2.weight.grad.numpy()[::2, ::2],
               g_bn2_w=blk18.bn2.weight.grad.numpy(), g_ds=blk18.downsample[0].weight.grad.numpy(),
               rm_bn1=blk18.bn1.running_mean.numpy().copy(), rv_bn1=blk18.bn1.running_var.numpy().copy())
    # the whole BasicBlock trunk on a small image: stem -> layer1..4 (train-mode BN), the taps the FPN reads
    enc18b = FR.ResNet(1, FR.BasicBlock, [2, 2, 2, 2])
    enc18b.load_state_dict({k[len(pre18):]: v for k, v in sd18.items() if k.startswith(pre18)}, strict=False)
    enc18b.train()
    img18 = torch.rand(2, 3, 96, 80, generator=g8)
    with torch.no_grad():
        t = enc18b.maxpool(enc18b.relu(enc18b.bn1(enc18b.conv1(img18))))
        t1 = enc18b.layer1(t)
        t2 = enc18b.layer2(t1)
        t3 = enc18b.layer3(t2)
        t4 = enc18b.layer4(t3)
    d18.update(img=img18.numpy(), c3=t2.numpy()[:, ::4], c4=t3.numpy()[:, ::8], c5=t4.numpy()[:, ::16])
    save("g8_basicblock", **d18)

    # ---- G9 head ordering on a 5x5 level ----------------------------------------------------
    xh = torch.randn(2, 514, 5, 5, generator=gf)
    with torch.no_grad():
        yh = net.permute_correctly(net.att_reg_box(xh), 5)
    save("g9_head", seed=np.array([3]), x=xh.numpy(), y=yh.numpy())

    # ---- G10 end-to-end forward/backward, B=2 ------------------------------------------------
    for tag, HW in (("e2e_128", 128), ("e2e_300", 300)):
        sd = O.seeded_state_dict("resnet50", seed=7)
        net = M.get_default_net(num_anchors=9, cfg=cfg)
        net.load_state_dict(sd, strict=False)
        net.train()
        bt = O.synthetic_batch(2, HW, HW, seed=1234)
        gq = torch.Generator().manual_seed(55)
        h0 = torch.randn(2, 2, 128, generator=gq)
        c0 = torch.randn(2, 2, 128, generator=gq)
        net.lstm_init_hidden = lambda bs: (h0, c0)
        out = net(bt)
        fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
        anc = A.create_anchors(fs, ratios, scales, device=cpu).float()
        lf = L.get_default_loss(ratios, scales, cfg)
        ev = E.get_default_eval(ratios, scales, cfg)
        lf.anchs = anc
        ev.anchs = anc
        ls = lf(out, bt)
        net.zero_grad()
        ls["loss"].backward()
        em = ev(out, bt)
        grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        keep = ["backbone.encoder.conv1.weight", "backbone.encoder.bn1.weight", "backbone.encoder.bn1.bias",
                "backbone.encoder.layer1.0.conv1.weight", "backbone.encoder.layer4.2.bn3.weight",
                "backbone.fpn.P3_1.bias", "backbone.fpn.P6.bias", "att_reg_box.5.bias", "att_reg_box.5.weight",
                "att_reg_box.0.0.bias", "lstm.bias_ih_l0", "lstm.bias_hh_l0_reverse", "lstm.weight_hh_l0_reverse"]
        d = dict(feat_sizes=np.array(fs), seed=np.array([7]), batch_seed=np.array([1234]), h0=h0.numpy(), c0=c0.numpy(),
                 loss=np.float64(ls["loss"].item()), cls_ls=np.float64(ls["cls_ls"].item()), box_ls=np.float64(ls["box_ls"].item()),
                 Acc=em["Acc"].detach().numpy(), MaxPos=em["MaxPos"].detach().numpy(), pred_boxes=em["pred_boxes"].detach().numpy(),
                 pred_scores=em["pred_scores"].detach().numpy(),
                 rm_bn1=net.backbone.encoder.bn1.running_mean.numpy(), rv_bn1=net.backbone.encoder.bn1.running_var.numpy(),
                 rm_l4=net.backbone.encoder.layer4[2].bn3.running_mean.numpy(),
                 rv_l4=net.backbone.encoder.layer4[2].bn3.running_var.numpy())
        ao, bo = out["att_out"].detach().numpy(), out["bbx_out"].detach().numpy()
        if HW == 128:
            d["att_out"], d["bbx_out"] = ao, bo
        else:
            d["att_out_s"], d["bbx_out_s"] = ao[:, ::7], bo[:, ::7]
            d["att_abs_sum"] = np.float64(np.abs(ao).astype(np.float64).sum())
            d["bbx_abs_sum"] = np.float64(np.abs(bo).astype(np.float64).sum())
        names = sorted(grads)
        d["grad_names"] = np.array(names)
        d["grad_norms"] = np.array([grads[k].double().norm().item() for k in names])
        for k in keep:
            d["grad__" + k] = grads[k].numpy()
        save("g10_" + tag, **d)
    # ---- G10b end-to-end at the BENCHMARK batch (configs[1]: B=16, 300x300): outputs sub-sampled, every gradient norm -----
    if not ONLY or any("g10_e2e_300_b16".startswith(o) for o in ONLY):
        sd = O.seeded_state_dict("resnet50", seed=7)
        net = M.get_default_net(num_anchors=9, cfg=cfg)
        net.load_state_dict(sd, strict=False)
        net.train()
        bt = O.synthetic_batch(16, 300, 300, seed=4321)
        gq = torch.Generator().manual_seed(56)
        h0 = torch.randn(2, 16, 128, generator=gq)
        c0 = torch.randn(2, 16, 128, generator=gq)
        net.lstm_init_hidden = lambda bs: (h0, c0)
        out = net(bt)
        fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
        anc = A.create_anchors(fs, ratios, scales, device=cpu).float()
        lf = L.get_default_loss(ratios, scales, cfg)
        ev = E.get_default_eval(ratios, scales, cfg)
        lf.anchs = anc
        ev.anchs = anc
        ls = lf(out, bt)
        net.zero_grad()
        ls["loss"].backward()
        em = ev(out, bt)
        grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        names = sorted(grads)
        keep = ["backbone.encoder.conv1.weight", "backbone.encoder.bn1.weight", "backbone.encoder.layer1.0.conv1.weight",
                "backbone.encoder.layer2.1.conv2.weight", "backbone.encoder.layer3.3.conv2.weight", "backbone.encoder.layer4.2.bn3.weight",
                "backbone.fpn.P3_2.weight", "backbone.fpn.P6.bias", "att_reg_box.5.bias", "att_reg_box.5.weight", "att_reg_box.2.0.weight",
                "att_reg_box.0.0.bias", "lstm.bias_ih_l0", "lstm.weight_hh_l0_reverse"]
        ao, bo = out["att_out"].detach().numpy(), out["bbx_out"].detach().numpy()
        sc = torch.sigmoid(out["att_out"].detach().squeeze(-1))
        top2 = sc.topk(2, dim=1)
        d = dict(feat_sizes=np.array(fs), seed=np.array([7]), batch_seed=np.array([4321]), h0=h0.numpy(), c0=c0.numpy(),
                 loss=np.float64(ls["loss"].item()), cls_ls=np.float64(ls["cls_ls"].item()), box_ls=np.float64(ls["box_ls"].item()),
                 Acc=em["Acc"].detach().numpy(), MaxPos=em["MaxPos"].detach().numpy(), pred_boxes=em["pred_boxes"].detach().numpy(),
                 pred_scores=em["pred_scores"].detach().numpy(), top1_idx=top2.indices[:, 0].numpy(), top2_gap=(top2.values[:, 0] - top2.values[:, 1]).numpy(),
                 att_out_s=ao[:, ::37], bbx_out_s=bo[:, ::37],
                 att_abs_sum=np.float64(np.abs(ao).astype(np.float64).sum()), bbx_abs_sum=np.float64(np.abs(bo).astype(np.float64).sum()),
                 rm_bn1=net.backbone.encoder.bn1.running_mean.numpy(), rv_l4=net.backbone.encoder.layer4[2].bn3.running_var.numpy(),
                 grad_names=np.array(names), grad_norms=np.array([grads[k].double().norm().item() for k in names]))
        for k in keep:
            g_ = grads[k].numpy()
            d["grad__" + k] = g_ if g_.size <= 20000 else g_.reshape(-1)[::max(1, g_.size // 20000)]
        save("g10_e2e_300_b16", **d)
    # ---- G11 SSD-VGG16 backbone (config 4): SSD.forward + ZSGNet head/loss, B=1, seeded weights -------------------
    S = R["ssd_vgg"]
    sd = O.seeded_ssd_state_dict(seed=5)
    enc = S.build_ssd("train", cfg=cfg)
    net = M.ZSGNet(M.SSDBackBone(enc, cfg), 9, cfg=cfg)
    res = net.load_state_dict(sd, strict=True)
    net.train()
    bt = O.synthetic_batch(1, 300, 300, seed=31)
    gq = torch.Generator().manual_seed(77)
    h0 = torch.randn(2, 1, 128, generator=gq)
    c0 = torch.randn(2, 1, 128, generator=gq)
    net.lstm_init_hidden = lambda bs: (h0, c0)
    with torch.no_grad():
        feats = enc(bt["img"])
    out = net(bt)
    fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
    anc = A.create_anchors(fs, ratios, scales, device=cpu).float()
    lf = L.get_default_loss(ratios, scales, cfg)
    lf.anchs = anc
    ls = lf(out, bt)
    net.zero_grad()
    ls["loss"].backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    names = sorted(grads)
    unused = sorted(k for k, p in net.named_parameters() if p.grad is None)
    d = dict(seed=np.array([5]), batch_seed=np.array([31]), h0=h0.numpy(), c0=c0.numpy(), feat_sizes=np.array(fs),
             loss=np.float64(ls["loss"].item()), att_out_s=out["att_out"].detach().numpy()[:, ::7], bbx_out_s=out["bbx_out"].detach().numpy()[:, ::7],
             grad_names=np.array(names), grad_norms=np.array([grads[k].double().norm().item() for k in names]), unused=np.array(unused),
             keys=np.array(sorted(net.state_dict().keys())))
    for i, f in enumerate(feats):
        d[f"feat{i}_s"] = f.numpy()[:, ::8, ::3, ::3] if f.shape[2] > 5 else f.numpy()
    for k in ("backbone.encoder.vgg.0.bias", "backbone.encoder.vgg.21.bias", "backbone.encoder.fproj1.bias", "backbone.encoder.extras.7.bias"):
        d["grad__" + k] = grads[k].numpy()
    save("g11_ssd", **d)

    # ---- G12 ablation variants (mdl.py:118-130, 199-210, 363-375): blind heads and do_norm, 128x128, B=2 ------------
    for tag, kw, head_in in (("lang_blind", dict(use_lang=False), 256), ("img_blind", dict(use_img=False), 256),
                             ("both_blind", dict(use_lang=False, use_img=False), 2), ("do_norm", dict(do_norm=True), 514),
                             ("two_heads", dict(use_same_atb=False), 514)):
        c2 = cfg.__class__(dict(cfg))
        c2.device = "cpu"
        for k, v in kw.items():
            c2[k] = v
        sd = O.seeded_state_dict("resnet50", seed=11, head_in=head_in, same_atb=c2["use_same_atb"])
        net = M.get_default_net(num_anchors=9, cfg=c2)
        net.load_state_dict(sd, strict=False)
        net.train()
        bt = O.synthetic_batch(2, 128, 128, seed=4321)
        gq = torch.Generator().manual_seed(56)
        h0 = torch.randn(2, 2, 128, generator=gq)
        c0 = torch.randn(2, 2, 128, generator=gq)
        net.lstm_init_hidden = lambda bs: (h0, c0)
        out = net(bt)
        fs = [tuple(int(v) for v in r) for r in out["feat_sizes"].tolist()]
        anc = A.create_anchors(fs, ratios, scales, device=cpu).float()
        lf = L.get_default_loss(ratios, scales, c2)
        lf.anchs = anc
        ls = lf(out, bt)
        net.zero_grad()
        ls["loss"].backward()
        grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
        names = sorted(grads)
        d = dict(seed=np.array([11]), batch_seed=np.array([4321]), head_in=np.array([head_in]), h0=h0.numpy(), c0=c0.numpy(),
                 feat_sizes=np.array(fs), loss=np.float64(ls["loss"].item()), att_out=out["att_out"].detach().numpy(),
                 bbx_out=out["bbx_out"].detach().numpy(), grad_names=np.array(names),
                 grad_norms=np.array([grads[k].double().norm().item() for k in names]),
                 unused=np.array(sorted(k for k, p in net.named_parameters() if p.grad is None)),
                 rm_bn1=net.backbone.encoder.bn1.running_mean.numpy())
        hp = "att_reg_box" if c2["use_same_atb"] else "reg_box"
        for k in (hp + ".0.0.bias", hp + ".5.bias"):
            d["grad__" + k] = grads[k].numpy()
        d["grad__" + hp + ".0.0.weight_s"] = grads[hp + ".0.0.weight"].numpy()[::8, ::5]
        if not c2["use_same_atb"]:
            d["grad__att_box.5.bias"] = grads["att_box.5.bias"].numpy()
            d["keys"] = np.array(sorted(net.state_dict().keys()))
        save("g12_" + tag, **d)

    # ---- G13 batch producer: ImgQuDataset + collater (dat_loader.py:68-196) on a tiny on-disk dataset ----------------
    # spaCy is absent: `nlp` is replaced by a table tokenizer (runs of word characters | single punctuation marks;
    # unknown words -> zero vector), the same rule zsgnet_pytorch_amd.dat_loader.TableEmbedder implements.
    import re
    import tempfile
    import PIL.Image
    rng = np.random.default_rng(13)
    words = ["the", "red", "dog", "left", "of", "a", "tree", "man", "in", "blue", "shirt", ",", "sky", "PD"]
    table = rng.normal(0, 0.3, (len(words), 300)).astype(np.float32)
    table[words.index("PD")] = 0.01

    class Tok:
        def __init__(self, t):
            self.text = t
            self.vector = table[words.index(t)] if t in words else np.zeros(300, np.float32)

    def fake_nlp(text):
        return [Tok(t) for t in re.findall(r"\w+|[^\w\s]", str(text))]
    imgs = {"a.png": rng.integers(0, 256, (30, 40, 3), dtype=np.uint8), "b.png": rng.integers(0, 256, (47, 33, 3), dtype=np.uint8),
            "c.png": rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)}
    rows = [("a.png", [3, 4, 30, 25], "the red dog"), ("b.png", [0, 0, 33, 47], "man in blue shirt , left of a tree"),
            ("c.png", [10.5, 20.25, 50, 60], "sky"), ("a.png", [5, 5, 12, 9], "unknownword left_of the tree"),
            ("c.png", [1, 2, 3, 4], "a dog in the sky , a man")]
    with tempfile.TemporaryDirectory() as td:
        for k, v in imgs.items():
            PIL.Image.fromarray(v).save(os.path.join(td, k))
        with open(os.path.join(td, "d.csv"), "w") as f:
            f.write("img_id,bbox,query\n")
            for i, b, q in rows:
                f.write(f'{i},"{b}","{q}"\n')
        os.chdir(REF)
        import dat_loader as DL
        os.chdir(REPO)
        DL.nlp = fake_nlp
        if not hasattr(np, "float_"):
            np.float_ = np.float64       # alias removed in NumPy 2.0; the reference (dat_loader.py:135) was written for 1.x
        c3 = cfg.__class__(dict(cfg))
        c3.resize_img = [48, 40]
        c3.ds_info = cfg.__class__({"refclef": cfg.__class__({"img_dir": td})})
        ds = DL.ImgQuDataset(c3, os.path.join(td, "d.csv"), "refclef")
        items = [ds[i] for i in range(len(ds))]
        batch = DL.collater(items[:3])
    d = dict(words=np.array(words), table=table, csv_img=np.array([r[0] for r in rows]), csv_bbox=np.array([r[1] for r in rows], dtype=np.float64),
             csv_query=np.array([r[2] for r in rows]), resize_img=np.array([48, 40]))
    for k, v in imgs.items():
        d["png_" + k[0]] = v
    for i, it in enumerate(items):
        for k, v in it.items():
            d[f"item{i}_{k}"] = v.numpy()
    for k, v in batch.items():
        d["batch_" + k] = v.numpy()
        d["batchdtype_" + k] = np.array(str(v.dtype))
    save("g13_dataset", **d)
    gen_resize(cfg)
    gen_learnable(R)
    gen_trajectory(R)
    print("done")


if __name__ == "__main__":
    main()
