"""CPU oracle for the ZSGNet training-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker / reported baseline.  The product path (``zsgnet-pytorch_amd``) never imports this module and
fails loudly when its HIP library is missing.

This is a from-scratch CPU restatement (numpy for the box/index arithmetic, torch-CPU fp32 functional
ops for the network) of the algorithm in the reference tree.  Every function cites the reference
``file:line`` it follows.  Parity status: **pinned** against golden vectors generated in the build
container by importing the reference itself (``tests/golden/make_golden.py`` → ``tests/golden/*.npz``,
checked by ``tests/test_oracle_golden.py``).  The reference ships no tests of its own.

Numeric regime: anchors are evaluated with the reference formula in float64 (what the
reference does under torch>=1.5 dtype inference) and rounded ONCE to float32; matching / arg-max / loss
then run in float32.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32

# ----------------------------------------------------------------------------------------------
# anchors.py restatement (numpy)
# ----------------------------------------------------------------------------------------------


def linspace_f32(start: float, end: float, steps: int) -> np.ndarray:
    """float32 linspace with torch's CPU rule, as used by reference anchors.py:54-55,58-59 through
    ``torch.linspace``: step = (end-start)/(steps-1) in fp32; lower half = fma(step, i, start), upper half =
    fma(-step, steps-1-i, end) (ONE rounding each: torch's vectorised kernel fuses the multiply-add; probed
    bit-exact against torch 2.10 for every size the pyramid uses).  The fused form is emulated through float64
    (step*i is exact in double for these sizes)."""
    start = F32(start)
    end = F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    i = np.arange(steps)
    lo = (np.float64(start) + np.float64(step) * i).astype(F32)
    hi = (np.float64(end) - np.float64(step) * (steps - 1 - i)).astype(F32)
    return np.where(i < steps // 2, lo, hi).astype(F32)


def create_grid(h: int, w: int) -> np.ndarray:
    """Cell centres in (-1,1): returns float32 [h*w, 2] with (y, x) per row.
    Reference anchors.py:47-63 (n == 1 gives 0)."""
    xs = linspace_f32(-1 + 1 / w, 1 - 1 / w, w) if w > 1 else np.zeros(1, F32)
    ys = linspace_f32(-1 + 1 / h, 1 - 1 / h, h) if h > 1 else np.zeros(1, F32)
    g = np.empty((h, w, 2), dtype=F32)
    g[:, :, 0] = ys[:, None]
    g[:, :, 1] = xs[None, :]
    return g.reshape(-1, 2)


def create_anchors(feat_sizes: Sequence[Tuple[int, int]], ratios: Sequence[float],
                   scales: Sequence[float]) -> np.ndarray:
    """float64 [A,4] (y1,x1,y2,x2).  Reference anchors.py:66-87 + cthw2tlbr :11-15.
    aspect list order: ratio-major, scale-minor (anchors.py:69-70); the per-level scale
    factor (2/h, 2/w) is a float32 tensor promoted to float64 by the multiply (:77-78)."""
    aspects = np.array([[s * np.sqrt(r), s * np.sqrt(1 / r)] for r in ratios for s in scales],
                       dtype=np.float64).reshape(-1, 2)
    out = []
    for h, w in feat_sizes:
        h, w = int(h), int(w)
        lvl_scale = np.array([2 / h, 2 / w], dtype=F32).astype(np.float64)
        sized = aspects * lvl_scale                                   # [a,2] (ah, aw)
        grid = create_grid(h, w).astype(np.float64)                   # [n,2] (cy, cx)
        n, a = grid.shape[0], sized.shape[0]
        ctr = np.broadcast_to(grid[:, None, :], (n, a, 2))
        sz = np.broadcast_to(sized[None, :, :], (n, a, 2))
        tl = ctr - sz / 2
        br = ctr + sz / 2
        out.append(np.concatenate([tl, br], axis=2).reshape(-1, 4))
    return np.concatenate(out, axis=0)


def default_ratios_scales(scale_factor: float = 4.0):
    """configs/cfg.json:23-25 evaluated as main_dist.py:24-31 does."""
    ratios = [1 / 2, 1, 2]
    scales = scale_factor * np.array([1, 2 ** (1 / 3), 2 ** (2 / 3)])
    return ratios, scales


def iou_values(boxes: np.ndarray, anchors: np.ndarray) -> np.ndarray:
    """float32 IoU [B,A] of boxes [B,4] vs anchors [A,4], both tlbr.
    Op order follows reference anchors.py:90-116 as called ``IoU_values(annot, anchs)``
    (loss.py:76, evaluator.py:78): inter = prod(clamp(min(br)-max(tl),0));
    union = (area(box) + area(anchor)) - inter; iou = inter / (union + 1e-8)."""
    b = boxes.astype(F32)[:, None, :]
    a = anchors.astype(F32)[None, :, :]
    tl = np.maximum(b[..., :2], a[..., :2])
    br = np.minimum(b[..., 2:], a[..., 2:])
    sz = np.maximum((br - tl).astype(F32), F32(0))
    inter = (sz[..., 0] * sz[..., 1]).astype(F32)
    bsz = (b[..., 2:] - b[..., :2]).astype(F32)
    asz = (a[..., 2:] - a[..., :2]).astype(F32)
    barea = (bsz[..., 0] * bsz[..., 1]).astype(F32)
    aarea = (asz[..., 0] * asz[..., 1]).astype(F32)
    union = ((barea + aarea).astype(F32) - inter).astype(F32)
    return (inter / (union + F32(1e-8)).astype(F32)).astype(F32)


def match_mask(iou: np.ndarray, thr: float, use_multi: bool = True):
    """positives mask [B,A] (bool) and arg-max anchor per box (int64, lowest index wins).
    Reference loss.py:73-87: (iou > thr) | onehot(argmax)."""
    best = np.argmax(iou, axis=1)          # numpy: first occurrence == torch CPU max(1)
    top1 = np.zeros_like(iou, dtype=bool)
    top1[np.arange(iou.shape[0]), best] = True
    if not use_multi:
        return top1, best
    return (iou > F32(thr)) | top1, best


def tlbr2cthw(b: np.ndarray) -> np.ndarray:
    """anchors.py:18-22 (center = (tl+br)/2, size = br-tl)."""
    c = (b[..., :2] + b[..., 2:]) / b.dtype.type(2)
    s = b[..., 2:] - b[..., :2]
    return np.concatenate([c, s], axis=-1)


def bbox_to_reg_params(anchors: np.ndarray, boxes: np.ndarray) -> np.ndarray:
    """float32 [B,A,4]: reference anchors.py:168-179."""
    bx = tlbr2cthw(boxes.astype(F32))[:, None, :]
    an = tlbr2cthw(anchors.astype(F32))[None, :, :]
    den = (an[..., 2:] + F32(1e-8)).astype(F32)
    trc = ((bx[..., :2] - an[..., :2]).astype(F32) / den).astype(F32)
    thw = np.log((bx[..., 2:] / den).astype(F32)).astype(F32)
    return np.concatenate([trc, thw], axis=2)


def reg_params_to_bbox(anchors: np.ndarray, reg: np.ndarray) -> np.ndarray:
    """float32 [B,A,4] tlbr: reference anchors.py:182-197."""
    an = tlbr2cthw(anchors.astype(F32))[None, :, :]
    reg = reg.astype(F32)
    ctr = ((an[..., 2:] * reg[..., :2]).astype(F32) + an[..., :2]).astype(F32)
    hw = (np.exp(reg[..., 2:]).astype(F32) * an[..., 2:]).astype(F32)
    half = (hw / F32(2)).astype(F32)
    return np.concatenate([(ctr - half).astype(F32), (ctr + half).astype(F32)], axis=2)


# ----------------------------------------------------------------------------------------------
# loss.py restatement (numpy, with analytic gradients)
# ----------------------------------------------------------------------------------------------


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return (F32(1) / (F32(1) + np.exp(-x.astype(F32)).astype(F32))).astype(F32)


def zsg_loss(att: np.ndarray, reg: np.ndarray, annot: np.ndarray, anchors_f32: np.ndarray,
             alpha: float = 0.25, gamma: float = 2.0, lamb_reg: float = 1.0, thr: float = 0.6,
             use_focal: bool = True, use_multi: bool = True, use_softmax: bool = False) -> Dict[str, np.ndarray]:
    """Reference loss.py:43-143.  att [B,A] logits, reg [B,A,4], annot [B,4] (y1x1y2x2 in [-1,1]).
    Returns loss scalars (float64-accumulated, the reference sums in fp32: rel 1e-5 tolerance),
    the positives mask / arg-max ids (exact) and d loss / d att, d loss / d reg (float32)."""
    B, A = att.shape
    att = att.astype(F32)
    reg = reg.astype(F32)
    iou = iou_values(annot, anchors_f32)
    mask, best = match_mask(iou, thr, use_multi)
    t = mask.astype(F32)
    npos_b = mask.sum(axis=1).astype(np.float64)                      # loss.py:94
    npos = float(mask.sum())                                         # loss.py:125

    gt = bbox_to_reg_params(anchors_f32, annot)                       # loss.py:90
    d = (reg - gt).astype(F32)
    ad = np.abs(d)
    sl1 = np.where(ad < 1, F32(0.5) * d * d, ad - F32(0.5)).astype(F32)   # SmoothL1 beta=1 (loss.py:41,91)
    box_b = (sl1.sum(axis=2, dtype=np.float64) * t).sum(axis=1) / npos_b
    box_loss = box_b.mean()
    dsl1 = np.where(ad < 1, d, np.sign(d)).astype(np.float64)
    g_reg = dsl1 * (t[..., None] / npos_b[:, None, None]) * (lamb_reg / B)

    p = _sigmoid(att)
    if use_softmax:                                                   # loss.py:106-109
        assert not use_multi
        x64 = att.astype(np.float64)
        m = x64.max(axis=1, keepdims=True)
        lse = m[:, 0] + np.log(np.exp(x64 - m).sum(axis=1))
        ce = lse - x64[np.arange(B), best]
        cls_sum = ce.sum()
        sm = np.exp(x64 - lse[:, None])
        sm[np.arange(B), best] -= 1.0
        g_att = sm / npos
    else:
        x64 = att.astype(np.float64)
        bce = np.maximum(x64, 0) - x64 * t + np.log1p(np.exp(-np.abs(x64)))   # loss.py:122-123
        if use_focal:                                                 # loss.py:111-118
            w = (t * (1 - p) + (1 - t) * p).astype(np.float64) ** gamma
            w = w * ((1 - t) * alpha + t * (1 - alpha))
        else:
            w = np.ones_like(x64)
        cls_sum = (w * bce).sum()
        g_att = w * (p.astype(np.float64) - t) / npos                 # weights are detached (loss.py:118)
    cls_loss = cls_sum / npos
    nan = bool(np.isnan(box_loss) or np.isnan(cls_loss))
    if nan:                                                           # loss.py:128-133
        box_loss, cls_loss = 0.01, 1.0
        g_att = np.zeros_like(g_att)
        g_reg = np.zeros_like(g_reg)
    return dict(loss=np.float64(lamb_reg * box_loss + cls_loss), cls_ls=np.float64(cls_loss),
                box_ls=np.float64(box_loss), mask=mask, best=best.astype(np.int64), iou=iou,
                g_att=g_att.astype(F32), g_reg=g_reg.astype(F32), nan=nan)


# ----------------------------------------------------------------------------------------------
# evaluator.py restatement (numpy)
# ----------------------------------------------------------------------------------------------


def zsg_eval(att: np.ndarray, reg: np.ndarray, annot: np.ndarray, img_size: np.ndarray,
             anchors_f32: np.ndarray, acc_thr: float = 0.5) -> Dict[str, np.ndarray]:
    """Reference evaluator.py:48-117.  att [B,A] logits; img_size [B,2] = (h,w).
    pred_boxes are pixels (x1,y1,x2,y2) (evaluator.py:96-98)."""
    B, A = att.shape
    score = _sigmoid(att)
    pred_id = np.argmax(score, axis=1)                                # evaluator.py:74-75
    iou = iou_values(annot, anchors_f32)
    exp_id = np.argmax(iou, axis=1)                                   # evaluator.py:78-79
    rows = np.arange(B)

    def pick(ids):
        sel_reg = reg.astype(F32)[rows, ids][:, None, :]             # decode only what is gathered
        boxes = np.stack([reg_params_to_bbox(anchors_f32[ids[b]:ids[b] + 1], sel_reg[b:b + 1])[0, 0]
                          for b in range(B)])
        # evaluator.py:115: diag(IoU_values(best_boxes, annot)) -> box is the "anchors" arg
        ious = np.array([iou_values(boxes[b:b + 1], annot[b:b + 1].astype(F32))[0, 0] for b in range(B)], F32)
        return (ious >= F32(acc_thr)), boxes, ious

    ok_pred, boxes, ious_pred = pick(pred_id)
    ok_best, _, _ = pick(exp_id)
    sz = img_size.astype(F32)
    half = ((boxes + F32(1)) / F32(2)).astype(F32)
    px = half.copy()
    px[:, :2] = sz * half[:, :2]
    px[:, 2:] = sz * half[:, 2:]
    xyxy = px[:, [1, 0, 3, 2]]
    return dict(Acc=F32(ok_pred.astype(F32).mean()), MaxPos=F32(ok_best.astype(F32).mean()),
                pred_boxes=xyxy.astype(F32), pred_scores=score[rows, pred_id], pred_ids=pred_id.astype(np.int64),
                best_ids=exp_id.astype(np.int64), pred_iou=ious_pred, boxes_norm=boxes)


# ----------------------------------------------------------------------------------------------
# seed-only weights (shared by the golden generator and the tests; never 150 MB of committed weights)
# ----------------------------------------------------------------------------------------------

ARCHS = {
    # name: (block kind, blocks per stage)
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
}


def resnet_conv_specs(arch: str):
    """[(name, cout, cin, k, stride, has_bn)] for the torchvision-layout encoder
    (reference mdl.py:148-156, fpn_resnet.py:26-100,262-276)."""
    kind, nblocks = ARCHS[arch]
    exp = 4 if kind == "bottleneck" else 1
    specs = [("conv1", 64, 3, 7, 2)]
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), nblocks), start=1):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 1) else 1
            pre = f"layer{li}.{bi}"
            if kind == "bottleneck":
                specs += [(f"{pre}.conv1", planes, inpl, 1, 1), (f"{pre}.conv2", planes, planes, 3, stride),
                          (f"{pre}.conv3", planes * 4, planes, 1, 1)]
            else:
                specs += [(f"{pre}.conv1", planes, inpl, 3, stride), (f"{pre}.conv2", planes, planes, 3, 1)]
            if bi == 0 and (stride != 1 or inpl != planes * exp):
                specs.append((f"{pre}.downsample.0", planes * exp, inpl, 1, stride))
            inpl = planes * exp
    return specs


VGG_BASE = [64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512]   # ssd_vgg.py:174-177
SSD_EXTRAS = [256, "S", 512, 128, "S", 256, 128, 256, 128, 256]                                        # ssd_vgg.py:179-182
SSD_MBOX = [4, 6, 6, 6, 4, 4]


def ssd_layer_specs():
    """VGG trunk + extras of build_ssd('train', 300) (ssd_vgg.py:117-155): returns (vgg, extras) where vgg is a list of
    ('conv', idx, cin, cout, k, pad, dil) | ('relu', idx) | ('pool', idx, k, s, p, ceil) and extras a list of
    (idx, cin, cout, k, stride, pad)."""
    vgg, cin, idx = [], 3, 0
    for v in VGG_BASE:
        if v == "M":
            vgg.append(("pool", idx, 2, 2, 0, False)); idx += 1
        elif v == "C":
            vgg.append(("pool", idx, 2, 2, 0, True)); idx += 1
        else:
            vgg.append(("conv", idx, cin, v, 3, 1, 1)); vgg.append(("relu", idx + 1)); idx += 2
            cin = v
    vgg.append(("pool", idx, 3, 1, 1, False)); idx += 1                       # pool5
    vgg.append(("conv", idx, 512, 1024, 3, 6, 6)); vgg.append(("relu", idx + 1)); idx += 2     # conv6 (dilated)
    vgg.append(("conv", idx, 1024, 1024, 1, 0, 1)); vgg.append(("relu", idx + 1)); idx += 2    # conv7
    extras, cin, flag, k = [], 1024, False, 0
    for i, v in enumerate(SSD_EXTRAS):
        if cin != "S":
            if v == "S":
                extras.append((k, cin, SSD_EXTRAS[i + 1], (1, 3)[flag], 2, 1)); k += 1
            else:
                extras.append((k, cin, v, (1, 3)[flag], 1, 0)); k += 1
            flag = not flag
        cin = v
    return vgg, extras


def ssd_forward(sd, img, six_hundred=False, prefix="backbone.encoder."):
    """SSD.forward, ssd_vgg.py:54-102: conv4_3 (channel-L2-normalised, no eps), conv7, four extras -> 256-ch maps."""
    vgg, extras = ssd_layer_specs()
    x = img
    sources = []
    for layer in vgg:
        if layer[0] == "conv":
            _, idx, ci, co, k, pad, dil = layer
            x = F.conv2d(x, sd[f"{prefix}vgg.{idx}.weight"], sd[f"{prefix}vgg.{idx}.bias"], 1, pad, dil)
        elif layer[0] == "relu":
            x = F.relu(x)
            if layer[1] == 22:                                  # vgg[0:23] ends with conv4_3's ReLU (ssd_vgg.py:75-80)
                sources.append(x / x.norm(dim=1, keepdim=True))
        else:
            _, idx, k, s_, p_, ceil = layer
            x = F.max_pool2d(x, k, s_, p_, ceil_mode=ceil)
    sources.append(x)
    for (k, ci, co, ks, st, pd) in extras:
        x = F.relu(F.conv2d(x, sd[f"{prefix}extras.{k}.weight"], sd[f"{prefix}extras.{k}.bias"], st, pd))
        if k % 2 == 1:
            sources.append(x)
    outs = [F.conv2d(sources[i], sd[f"{prefix}fproj{i + 1}.weight"], sd[f"{prefix}fproj{i + 1}.bias"]) for i in range(3)] + sources[3:]
    return outs[1:] if six_hundred else outs


def seeded_ssd_state_dict(seed: int = 0, n_anchors: int = 9, emb_dim: int = 300, lstm_dim: int = 128, head_in: int = 514):
    """Seed-only weights with the reference's SSD key names (incl. the never-used loc/conf heads, ssd_vgg.py:157-171)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / (ci * k * k))
        sd[name + ".bias"] = (torch.rand(co, generator=g) - 0.5) * 0.2
    p = "backbone.encoder."
    vgg, extras = ssd_layer_specs()
    for layer in vgg:
        if layer[0] == "conv":
            conv(f"{p}vgg.{layer[1]}", layer[3], layer[2], layer[4])
    conv(p + "fproj1", 256, 512, 1)
    conv(p + "fproj2", 256, 1024, 1)
    conv(p + "fproj3", 256, 512, 1)
    for (k, ci, co, ks, st, pd) in extras:
        conv(f"{p}extras.{k}", co, ci, ks)
    src_ch = [512, 1024, 512, 256, 256, 256]
    for i, (c, nb) in enumerate(zip(src_ch, SSD_MBOX)):
        conv(f"{p}loc.{i}", nb * 4, c, 3)
    for i, (c, nb) in enumerate(zip(src_ch, SSD_MBOX)):
        conv(f"{p}conf.{i}", nb * 21, c, 3)
    base = seeded_state_dict("resnet18", seed + 1, n_anchors, emb_dim, lstm_dim, head_in)
    for k_, v in base.items():
        if k_.startswith("att_reg_box.") or k_.startswith("lstm."):
            sd[k_] = v
    return sd


def fpn_in_channels(arch: str):
    kind, _ = ARCHS[arch]
    e = 4 if kind == "bottleneck" else 1
    return [128 * e, 256 * e, 512 * e]


def seeded_state_dict(arch: str = "resnet50", seed: int = 0, n_anchors: int = 9, emb_dim: int = 300,
                      lstm_dim: int = 128, head_in: int = 514, bn_noise: bool = True, same_atb: bool = True) -> Dict[str, torch.Tensor]:
    """Deterministic random weights keyed by the reference's parameter names:
    He-normal convs, BN gamma~1 beta~0 (slightly perturbed so the affine path is exercised),
    LSTM U(+-1/sqrt(H)), head bias pattern [0,0,0,0,-4]*n_anchors (mdl.py:214-219)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci, k, bias):
        fan = ci * k * k
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * math.sqrt(2.0 / fan)
        if bias:
            sd[name + ".bias"] = (torch.rand(co, generator=g) - 0.5) * 0.2

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + (0.1 * torch.randn(c, generator=g) if bn_noise else torch.zeros(c))
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g) if bn_noise else torch.zeros(c)
        sd[name + ".running_mean"] = torch.zeros(c)
        sd[name + ".running_var"] = torch.ones(c)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    enc = "backbone.encoder."
    for name, co, ci, k, s in resnet_conv_specs(arch):
        conv(enc + name, co, ci, k, bias=False)
        if name == "conv1":
            bn(enc + "bn1", co)
        elif name.endswith("downsample.0"):
            bn(enc + name[:-1] + "1", co)
        else:
            bn(enc + name.replace("conv", "bn"), co)
    c3, c4, c5 = fpn_in_channels(arch)
    fp = "backbone.fpn."
    for name, co, ci, k in (("P7_2", 256, 256, 3), ("P6", 256, c5, 3), ("P5_1", 256, c5, 1), ("P5_2", 256, 256, 3),
                            ("P4_1", 256, c4, 1), ("P4_2", 256, 256, 3), ("P3_1", 256, c3, 1), ("P3_2", 256, 256, 3)):
        conv(fp + name, co, ci, k, bias=True)
    for prefix, ncls in ([("att_reg_box", 5)] if same_atb else [("att_box", 1), ("reg_box", 4)]):
        conv(prefix + ".0.0", 256, head_in, 3, bias=True)
        for i in range(1, 5):
            conv(f"{prefix}.{i}.0", 256, 256, 3, bias=True)
        conv(prefix + ".5", ncls * n_anchors, 256, 3, bias=True)
    if same_atb:
        hb = torch.zeros(5 * n_anchors)
        hb[4::5] = -4.0
        sd["att_reg_box.5.bias"] = hb
    else:                                # mdl.py:221-225
        sd["att_box.5.bias"] = torch.full((n_anchors,), -4.0)
        sd["reg_box.5.bias"] = torch.zeros(4 * n_anchors)
    k = 1.0 / math.sqrt(lstm_dim)
    for suf in ("", "_reverse"):
        sd["lstm.weight_ih_l0" + suf] = (torch.rand(4 * lstm_dim, emb_dim, generator=g) * 2 - 1) * k
        sd["lstm.weight_hh_l0" + suf] = (torch.rand(4 * lstm_dim, lstm_dim, generator=g) * 2 - 1) * k
        sd["lstm.bias_ih_l0" + suf] = (torch.rand(4 * lstm_dim, generator=g) * 2 - 1) * k
        sd["lstm.bias_hh_l0" + suf] = (torch.rand(4 * lstm_dim, generator=g) * 2 - 1) * k
    return sd


def synthetic_batch(B: int, H: int = 300, W: int = 300, T: int = 20, seed: int = 1234, tmax: int = 20):
    """SURVEY.md §8(d) synthetic inputs (batch contract of dat_loader.py:136-144,187-196)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    qvec = torch.randn(B, T, 300, generator=g) * 0.35
    qlens = torch.randint(1, tmax + 1, (B,), generator=g).float()
    qlens[0] = float(tmax)
    c = (torch.rand(B, 2, generator=g) * 1.2 - 0.6)
    s = (torch.rand(B, 2, generator=g) * 0.8 + 0.1)
    annot = torch.cat([c - s / 2, c + s / 2], dim=1).clamp(-1, 1)
    return dict(img=img, qvec=qvec, qlens=qlens, annot=annot, idxs=torch.arange(B).float(),
                img_size=torch.tensor([[360.0, 480.0]]).repeat(B, 1))


def learnable_batch(B: int, S: int = 128, T: int = 20, seed: int = 0):
    """A LEARNABLE synthetic grounding task (no dataset is reachable offline): the annotated box is a bright rectangle on a dim
    noisy background — the box is tied to an image feature, unlike synthetic_batch()'s random boxes — with noise queries.
    Same batch contract (dat_loader.py:136-144,187-196; annot = y1x1y2x2 in [-1, 1] as synthetic_batch).  A ZSGNet trained on it
    with Adam (lr 1e-3) for ~150 steps of 16 reaches Acc@IoU0.5 ~ 1.0 in eval mode: tests/test_gpu_fullshape.py trains the
    HIP path and this oracle side by side on it and compares the accuracies they reach."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, S, S, generator=g) * 0.25
    qvec = torch.randn(B, T, 300, generator=g) * 0.35
    qlens = torch.randint(1, T + 1, (B,), generator=g).float()
    qlens[0] = float(T)
    c = torch.rand(B, 2, generator=g) * 1.0 - 0.5
    s = torch.rand(B, 2, generator=g) * 0.5 + 0.3
    annot = torch.cat([c - s / 2, c + s / 2], dim=1).clamp(-1, 1)
    for b in range(B):
        p0, q0, p1, q1 = [int(round((float(v) + 1) / 2 * (S - 1))) for v in annot[b]]
        img[b, :, p0:p1 + 1, q0:q1 + 1] += 0.6
    return dict(img=img.clamp(0, 1), qvec=qvec, qlens=qlens, annot=annot, idxs=torch.arange(B).float(),
                img_size=torch.tensor([[360.0, 480.0]]).repeat(B, 1))


# ----------------------------------------------------------------------------------------------
# mdl.py / fpn_resnet.py restatement (torch CPU functional, fp32)
# ----------------------------------------------------------------------------------------------


class BNState:
    """Carries running statistics through a functional forward (train-mode BatchNorm2d, momentum .1, eps 1e-5)."""

    def __init__(self, sd: Dict[str, torch.Tensor], training: bool):
        self.sd = sd
        self.training = training

    def __call__(self, x, name):
        sd = self.sd
        if self.training and (name + ".num_batches_tracked") in sd:
            sd[name + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], self.training, 0.1, 1e-5)


def encoder_forward(sd, img, arch: str, bn: BNState, prefix="backbone.encoder."):
    """stem + layer1..4; returns (c3, c4, c5).  Reference mdl.py:148-156; blocks fpn_resnet.py:41-58,80-100."""
    kind, nblocks = ARCHS[arch]
    p = prefix
    x = F.conv2d(img, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(bn(x, p + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nb in enumerate(nblocks, start=1):
        for bi in range(nb):
            q = f"{p}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = x
            if kind == "bottleneck":
                o = F.relu(bn(F.conv2d(x, sd[q + "conv1.weight"]), q + "bn1"))
                o = F.relu(bn(F.conv2d(o, sd[q + "conv2.weight"], None, stride, 1), q + "bn2"))
                o = bn(F.conv2d(o, sd[q + "conv3.weight"]), q + "bn3")
            else:
                o = F.relu(bn(F.conv2d(x, sd[q + "conv1.weight"], None, stride, 1), q + "bn1"))
                o = bn(F.conv2d(o, sd[q + "conv2.weight"], None, 1, 1), q + "bn2")
            if (q + "downsample.0.weight") in sd:
                idt = bn(F.conv2d(x, sd[q + "downsample.0.weight"], None, stride), q + "downsample.1")
            x = F.relu(o + idt)
        feats.append(x)
    return feats[1], feats[2], feats[3]


def fpn_forward(sd, c3, c4, c5, six_hundred: bool = False, prefix="backbone.fpn."):
    """Reference fpn_resnet.py:154-178."""
    def cv(name, x, stride=1, pad=0):
        return F.conv2d(x, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], stride, pad)
    p51 = cv("P5_1", c5)
    p5 = cv("P5_2", p51, 1, 1)
    p41 = cv("P4_1", c4) + F.interpolate(p51, size=c4.shape[2:])
    p4 = cv("P4_2", p41, 1, 1)
    p31 = cv("P3_1", c3) + F.interpolate(p41, size=c3.shape[2:])
    p3 = cv("P3_2", p31, 1, 1)
    p6 = cv("P6", c5, 2, 1)
    p7 = cv("P7_2", F.relu(p6), 2, 1)
    if six_hundred:
        return [p4, p5, p6, p7]
    p8 = F.adaptive_avg_pool2d(p7, 1)
    return [p3, p4, p5, p6, p7, p8]


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """One torch.nn.LSTM cell step, gate order (i, f, g, o)."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, g, o = gates.chunk(4, dim=1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def sort_rank(qlens: torch.Tensor) -> torch.Tensor:
    """rank[b] = position of sample b after a stable descending sort of lengths (mdl.py:309)."""
    perm = torch.sort(qlens, descending=True, stable=True)[1]
    rank = torch.empty_like(perm)
    rank[perm] = torch.arange(len(perm))
    return rank


def query_encoder(sd, qvec, qlens, h0, c0, rank: Optional[torch.Tensor] = None):
    """[B, 2H] = [h_fwd(len-1) || reverse-cell(x[len-1]; h0[1],c0[1])].
    Reference mdl.py:296-336 with a packed bidirectional nn.LSTM: the backward direction's output at the
    last valid token is its FIRST step.  h0/c0 [2,B,H] are indexed by *sorted* position."""
    B = qvec.shape[0]
    if rank is None:
        rank = sort_rank(qlens)
    lens = qlens.long()
    outs = []
    for b in range(B):
        r = int(rank[b])
        h, c = h0[0, r:r + 1], c0[0, r:r + 1]
        for t in range(int(lens[b])):
            h, c = lstm_cell(qvec[b:b + 1, t], h, c, sd["lstm.weight_ih_l0"], sd["lstm.weight_hh_l0"],
                             sd["lstm.bias_ih_l0"], sd["lstm.bias_hh_l0"])
        hr, _ = lstm_cell(qvec[b:b + 1, int(lens[b]) - 1], h0[1, r:r + 1], c0[1, r:r + 1],
                          sd["lstm.weight_ih_l0_reverse"], sd["lstm.weight_hh_l0_reverse"],
                          sd["lstm.bias_ih_l0_reverse"], sd["lstm.bias_hh_l0_reverse"])
        outs.append(torch.cat([h, hr], dim=1))
    return torch.cat(outs, dim=0)


def fuse_lang_grid(feat, we):
    """[feat(256) || we(256) || grid(2: y then x)] -> [B,514,h,w].  Reference mdl.py:69-104."""
    B, _, h, w = feat.shape
    grid = torch.from_numpy(create_grid(h, w)).view(h, w, 2).permute(2, 0, 1).to(feat.dtype)
    return torch.cat([feat, we.view(B, -1, 1, 1).expand(B, we.shape[1], h, w),
                      grid.unsqueeze(0).expand(B, 2, h, w)], dim=1)


def head_forward(sd, x, prefix="att_reg_box.", outc=5):
    """6-conv head (mdl.py:235-244) + permute_correctly (mdl.py:246-254) -> [B, h*w*9, outc]."""
    for i in range(5):
        x = F.relu(F.conv2d(x, sd[f"{prefix}{i}.0.weight"], sd[f"{prefix}{i}.0.bias"], 1, 1))
    x = F.conv2d(x, sd[prefix + "5.weight"], sd[prefix + "5.bias"], 1, 1)
    return x.permute(0, 2, 3, 1).contiguous().view(x.shape[0], -1, outc)


def head_input(feat, we, use_lang=True, use_img=True):
    """What the head sees for one pyramid level (mdl.py:69-104 with the blind variants of mdl.py:363-375):
    full = [feat || we || grid]; language-blind = feat; image-blind = we tile only; both blind = grid only."""
    if use_lang and use_img:
        return fuse_lang_grid(feat, we)
    if use_img:
        return feat
    B, _, h, w = feat.shape
    if use_lang:
        return we.view(B, -1, 1, 1).expand(B, we.shape[1], h, w)
    grid = torch.from_numpy(create_grid(h, w)).view(h, w, 2).permute(2, 0, 1).to(feat.dtype)
    return grid.unsqueeze(0).expand(B, 2, h, w)


def zsgnet_forward(sd, batch, h0, c0, arch="resnet50", training=True, six_hundred=False, rank=None,
                   use_lang=True, use_img=True, do_norm=False):
    """Reference mdl.py:338-403.  Returns dict(att_out [B,A,1], bbx_out [B,A,4], feat_sizes [L,2], num_f_out [1]).
    The encoder always runs (the blind variants still take the pyramid sizes — and, in train mode, the BatchNorm
    running-statistics update — from it, mdl.py:363-375); do_norm = per-pixel channel L2 normalisation of the maps and
    of the language vector, without epsilon (mdl.py:118-130; not applied to `we` in the language-blind call)."""
    bn = BNState(sd, training)
    we = query_encoder(sd, batch["qvec"], batch["qlens"], h0, c0, rank)
    if arch == "ssd_vgg":
        feats = ssd_forward(sd, batch["img"], six_hundred)
    else:
        c3, c4, c5 = encoder_forward(sd, batch["img"], arch, bn)
        feats = fpn_forward(sd, c3, c4, c5, six_hundred)
    wn = we
    if do_norm:
        feats = [f / f.norm(dim=1, keepdim=True) for f in feats]
        wn = we / we.norm(dim=1, keepdim=True)
    xs = [head_input(f, wn, use_lang, use_img) for f in feats]
    if "att_reg_box.5.weight" in sd:                 # shared head (use_same_atb, the paper's configuration)
        ab = torch.cat([head_forward(sd, x) for x in xs], dim=1)
        att, bbx = ab[..., 4:5], ab[..., :4]
    else:                                            # separate heads, mdl.py:383-389
        att = torch.cat([head_forward(sd, x, "att_box.", 1) for x in xs], dim=1)
        bbx = torch.cat([head_forward(sd, x, "reg_box.", 4) for x in xs], dim=1)
    return dict(att_out=att, bbx_out=bbx,
                feat_sizes=torch.tensor([[f.shape[2], f.shape[3]] for f in feats]),
                num_f_out=torch.tensor([len(feats)]), we=we, feats=feats)


def torch_loss(out, annot, anchors_f32: torch.Tensor, alpha=0.25, gamma=2.0, lamb_reg=1.0, thr=0.6):
    """Differentiable torch version of zsg_loss (default flags) used for end-to-end gradient checks and the
    CPU-baseline train step.  Mask / targets come from the numpy matcher above."""
    att = out["att_out"].squeeze(-1)
    reg = out["bbx_out"]
    anc = anchors_f32.numpy()
    iou = iou_values(annot.float().numpy(), anc)
    mask, _ = match_mask(iou, thr, True)
    t = torch.from_numpy(mask.astype(F32)).to(att.dtype)
    gt = torch.from_numpy(bbox_to_reg_params(anc, annot.float().numpy())).to(att.dtype)
    box = (F.smooth_l1_loss(reg, gt, reduction="none").sum(2) * t).sum(1) / t.sum(1)
    box = box.mean()
    p = torch.sigmoid(att).detach()
    w = (t * (1 - p) + (1 - t) * p).pow(gamma) * ((1 - t) * alpha + t * (1 - alpha))
    cls = (w * F.binary_cross_entropy_with_logits(att, t, reduction="none")).sum() / t.sum()
    return dict(loss=lamb_reg * box + cls, cls_ls=cls, box_ls=box)


def feat_sizes_for(H: int, W: int, six_hundred: bool = False) -> List[Tuple[int, int]]:
    """Pyramid sizes of the ResNet+FPN path for an HxW image (300 -> 38,19,10,5,3,1)."""
    def down(n, k, s, p):
        return (n + 2 * p - k) // s + 1
    def chain(n):
        n = down(n, 7, 2, 3)
        n = down(n, 3, 2, 1)       # layer1
        c3 = down(n, 3, 2, 1)
        c4 = down(c3, 3, 2, 1)
        c5 = down(c4, 3, 2, 1)
        p6 = down(c5, 3, 2, 1)
        p7 = down(p6, 3, 2, 1)
        return c3, c4, c5, p6, p7
    hs, ws = chain(H), chain(W)
    lv = list(zip(hs, ws))
    if six_hundred:
        return lv[1:5]
    return lv + [(1, 1)]


def cpu_train_step(sd_params: Dict[str, torch.Tensor], sd_buffers: Dict[str, torch.Tensor], opt, batch, h0, c0,
                   anchors_f32: torch.Tensor, arch="resnet50"):
    """One full reference-order step (utils.py:407-414) on CPU: zero_grad -> forward -> loss -> backward ->
    Adam -> evaluator.  Used as bench.py's cpu_baseline ("port")."""
    opt.zero_grad()
    sd = dict(sd_params)
    sd.update(sd_buffers)
    out = zsgnet_forward(sd, batch, h0, c0, arch=arch, training=True)
    for k in sd_buffers:                       # keep updated running stats
        sd_buffers[k] = sd[k]
    ls = torch_loss(out, batch["annot"], anchors_f32)
    ls["loss"].backward()
    opt.step()
    ev = zsg_eval(out["att_out"].detach().squeeze(-1).numpy(), out["bbx_out"].detach().numpy(),
                  batch["annot"].numpy(), batch["img_size"].numpy(), anchors_f32.numpy())
    return ls, ev


# ----------------------------------------------------------------------------------------------
# PIL.Image.resize restatement (the reference's only resampling call, dat_loader.py:121: default filter = bicubic for RGB)
# ----------------------------------------------------------------------------------------------
def _pil_bicubic(x):
    """Pillow Resample.c bicubic_filter (a = -0.5), vectorised in float64 with Pillow's operation order"""
    a = -0.5
    x = np.abs(np.asarray(x, np.float64))
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def _pil_axis_pass(src: np.ndarray, out_size: int) -> np.ndarray:
    """One pass of Pillow's ImagingResample over axis 0 of a uint8 array [n_in, ...]: per output index the window
    [xmin, xmin + count), float64 weights normalised to 1, rounded to 22-bit fixed point (normalize_coeffs_8bpc), accumulated in
    int32 from 1 << 21 and shifted / clipped to uint8 (clip8)."""
    n_in = src.shape[0]
    if n_in == out_size:                       # (ImagingResample skips a pass that does not change the size)
        return src
    scale = float(np.float32(n_in)) / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    s32 = src.astype(np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in) - xmin
        w = _pil_bicubic((np.arange(xmax) + xmin - center + 0.5) * (1.0 / fscale))
        ww = 0.0
        for v in w:                            # (sequential sum, as the C loop)
            ww += float(v)
        if ww != 0.0:
            w = w / ww
        k = np.where(w < 0, (-0.5 + w * (1 << 22)).astype(np.int64), (0.5 + w * (1 << 22)).astype(np.int64))      # C cast: toward zero
        acc = (1 << 21) + np.tensordot(k, s32[xmin:xmin + xmax], axes=(0, 0))
        out[xx] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out


def pil_resize_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """PIL.Image.fromarray(img).resize((out_w, out_h)) for a uint8 [H, W, 3] image: horizontal pass, then vertical pass
    (ImagingResample, Resample.c), each rounding to uint8."""
    img = np.ascontiguousarray(img, np.uint8)
    tmp = _pil_axis_pass(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)       # over x
    return np.ascontiguousarray(_pil_axis_pass(tmp, out_h))                        # over y
