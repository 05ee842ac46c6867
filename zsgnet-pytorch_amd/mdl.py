"""ZSGNet on MI355X: same module surface as the reference `code/mdl.py` (get_default_net / ZSGNet.forward), with
every device op a hand-written HIP kernel reached through the C ABI (include/zsg.h).

Reference behaviour mirrored here (file:line in /root/reference/code):
  * ZSGNet.forward mdl.py:338-403 — dict in (img [B,3,H,W], qvec [B,T,300], qlens [B]) -> dict out
    (att_out [B,A,1], bbx_out [B,A,4], feat_sizes [L,2] int64, num_f_out [1] int64).
  * RetinaBackBone.encode_feats mdl.py:148-159 (ResNet stem + layer1..4, Bottleneck fpn_resnet.py:61-100,
    BasicBlock :26-58) and FPN_backbone.forward fpn_resnet.py:154-178.
  * BackBone.concat_we mdl.py:69-104 (channel order [feat | language vector | grid y,x]).
  * shared 6-conv head mdl.py:235-244 + permute_correctly :246-254 (free here: NHWC output == [B, h*w*9, 5]).
  * BiLSTM query encoder apply_lstm mdl.py:296-336, random initial state lstm_init_hidden :279-294.

Execution model: for a given input geometry the network is lowered ONCE into two static launch programs
(forward, backward) over preallocated NHWC buffers (`ops.Program`); a step replays them on torch's current stream.
The whole forward is a single autograd node, so the reference trainer's `loss.backward()` / `optimizer.step()`
keep working while no autograd graph is built per layer.
"""
import ctypes as _ct
import math
import os
import types
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from . import anchors as anchors_mod
from ._lib import check, lib, require_gpu, stream_ptr
from .ops import (Level, Program, TView, WinoJobs, apply_main_priority_env, autotune_conv, autotune_wgrad_batch, conv_out, ensure_stream_scratch, dgrad_desc, fwd_desc, igemm_partial_rows, marshal,
                  shared_side_stream, tile_hint, wino_mode, wino_ok)
from .params import ParamStore, pad4, register_named

VGG_BASE = [64, 64, "M", 128, 128, "M", 256, 256, 256, "C", 512, 512, 512, "M", 512, 512, 512]     # ssd_vgg.py:174-177
SSD_EXTRAS = [256, "S", 512, 128, "S", 256, 128, 256, 128, 256]                                          # ssd_vgg.py:179-182

ARCHS = {
    "resnet18": ("basic", (2, 2, 2, 2)),
    "resnet34": ("basic", (3, 4, 6, 3)),
    "resnet50": ("bottleneck", (3, 4, 6, 3)),
    "resnet101": ("bottleneck", (3, 4, 23, 3)),
}


@dataclass
class ConvL:
    name: str
    cin: int
    cout: int
    k: int
    stride: int = 1
    pad: int = 0
    dil: int = 1
    bias: bool = False
    merge_x: bool = False

    @property
    def cpad(self):
        return pad4(self.cin)


@dataclass
class BnL:
    name: str
    c: int
    index: int = 0      # slot in the flat running-stat buffers


BNB_FUSE = os.environ.get("ZSG_BNB_FUSE", "1") != "0"     # BatchNorm-backward sums in the epilogue of the data gradient that completes dout
# BatchNorm statistics / backward sums FINALISED by the last-arriving tile of the producing convolution (csrc/bn_tail.h, round 5): no
# finalize launch, no re-reduction in the apply pass, wherever the launch has <= 128 partial rows per column block ("0": rounds 1-4's
# separate finalize / inline apply; "fwd" / "bwd": one direction only — A/B switches)
FPN_ORDER_DEFAULT = "p6m"
# bn3 + residual + ReLU of a bottleneck applied by the NEXT block's conv1 (zsg_conv_igemm_bnpre) instead of a zsg_bn_apply launch, where the
# activation is at least this large (MB; 0 = never): the apply pass is HBM-bound there and the consumer would read its output again
BN_PRE_MIN_MB = float(os.environ.get("ZSG_BN_PRE_MIN_MB", "40"))
STAGE_INPUTS = os.environ.get("ZSG_STAGE_INPUTS", "1") != "0"      # (A/B: 0 = the separate torch copies of rounds 1-4)
BN_TAIL = os.environ.get("ZSG_BN_TAIL", "1")
MASKED_DOUT = os.environ.get("ZSG_MASKED_DOUT", "1") != "0"      # the completing data gradient stores the ReLU-masked dout = the residual's gradient (bn(): back)
WG_BATCH = os.environ.get("ZSG_WG_BATCH", "1") != "0"      # identical-shape Winograd weight gradients of a stage in ONE launch (_Plan._batch_wgrads)
SK_BWD = os.environ.get("ZSG_SK_BWD", "0") != "0"      # stream-K candidates also for the backward's data gradients (measured slower: ops.autotune_conv)
BN_TAIL_MIN_ROWS = int(os.environ.get("ZSG_BN_TAIL_MIN_ROWS", "0"))      # (A/B: only launches with more partial rows than this finalise in-kernel)
def prep_at() -> str:
    """ZSG_PREP_AT: where the backward's weight images are enqueued on the side stream during the forward (see _Plan._prep_index)."""
    return os.environ.get("ZSG_PREP_AT", "j2")


def lang_at() -> str:
    """ZSG_LANG_AT: where the forward program releases the query encoder + language maps on the side stream: "head" = first (they
    co-run with the stem), "stem" = behind the stem, "j<n>" = behind the n-th join (see _Plan._lower)."""
    return os.environ.get("ZSG_LANG_AT", "j3")


def adam_overlap() -> bool:
    """ZSG_ADAM_OVERLAP=1 (default OFF): with FusedAdam attached the backward does not join the side stream at its end; FusedAdam.step
    updates every parameter behind the stem / first block under the side stream's last weight gradients and the rest after the join
    (bit-identical: tests/test_gpu_determinism.py).  Measured in round 3: 14.30 vs 14.28 ms; off by default."""
    return os.environ.get("ZSG_ADAM_OVERLAP", "0") == "1"


def prep_release_top() -> bool:
    return os.environ.get("ZSG_PREP_RELEASE_TOP", "1") == "1"


class Act(TView):
    """TView + autograd bookkeeping used while lowering (grad buffer, whether it already holds a partial sum, and
    whether the gradient stored there is w.r.t. the pre-ReLU value so producers must apply the ReLU mask)."""

    def __init__(self, buf, B, C, ld, levels, name=""):
        super().__init__(buf, B, C, ld, levels)
        self.name = name
        self.grad: Optional["Act"] = None
        self.gfilled = False
        self.needs_mask = False
        self.requires_grad = True

    def lvl(self, i) -> "Act":
        a = Act(self.buf, self.B, self.C, self.ld, [self.levels[i]], f"{self.name}[{i}]")
        a.requires_grad = self.requires_grad
        return a


class ZSGNet(nn.Module):
    """The main model (reference mdl.py:171-403).  `backbone_kind` in {'retina', 'ssd_vgg'}."""

    def __init__(self, backbone_kind: str = "retina", n_anchors: int = 9, cfg: Any = None, arch: str = "resnet50"):
        super().__init__()
        self.cfg = cfg
        self.backbone_kind = backbone_kind
        self.arch = arch
        self.n_anchors = n_anchors
        self.emb_dim = int(cfg["emb_dim"])
        self.bid = bool(cfg["use_bidirectional"])
        self.lstm_dim = int(cfg["lstm_dim"])
        self.lstm_out_dim = self.lstm_dim * (self.bid + 1)
        self.use_lang = bool(cfg["use_lang"])
        self.use_img = bool(cfg["use_img"])
        self.same_atb = bool(cfg["use_same_atb"])
        if "use_hip_graph" in cfg and cfg["use_hip_graph"]:
            from . import ops as _ops          # opt-in: replay launch ranges as hipGraphs (measured slower on ROCm 7.2)
            _ops.HIP_GRAPH = True
        if backbone_kind not in ("retina", "ssd_vgg"):
            raise ValueError(f"mdl_to_use={backbone_kind!r}: expected 'retina' or 'ssd_vgg' (mdl.py:410-414)")
        self.do_norm = bool(cfg["do_norm"])
        # fpn_resnet.py:173 tests resize_img == [600,600]; ssd_vgg.py:98 tests resize_img[0] >= 600
        self.six_hundred = (list(cfg["resize_img"]) == [600, 600]) if backbone_kind == "retina" else (cfg["resize_img"][0] >= 600)
        self.cf = 256 if self.use_img else 0
        self.cw = self.lstm_out_dim if self.use_lang else 0
        self.use_grid = (self.use_img and self.use_lang) or (not self.use_img and not self.use_lang)
        self.start_dim_head = self.cf + self.cw + (2 if self.use_grid else 0)       # mdl.py:196-209
        self.lstm_state = "randn"          # 'randn' (reference) | 'zeros'

        self.store = ParamStore()
        self.convs: Dict[str, ConvL] = {}
        self.bns: Dict[str, BnL] = {}
        self._declare()
        self.store.allocate(torch.device("cpu"))
        nb = sum(b.c for b in self.bns.values())
        self._rmv = torch.cat([torch.zeros(nb), torch.ones(nb)])      # running means | running variances: ONE buffer, so the
        self._rm, self._rv = self._rmv[:nb], self._rmv[nb:]            # per-forward DDP buffer sync (C2) is one broadcast
        self._nbt = torch.zeros(len(self.bns), dtype=torch.long)
        self._register()
        self.reset_parameters()
        self._plans: Dict[Tuple, "_Plan"] = {}
        self._anchor = None
        self.debug = False

    # ------------------------------------------------------------------------------------------------------
    # declaration (names == reference state_dict keys)
    # ------------------------------------------------------------------------------------------------------
    def _conv(self, name, cin, cout, k, stride=1, pad=0, dil=1, bias=False, merge_x=False) -> ConvL:
        L = ConvL(name, cin, cout, k, stride, pad, dil, bias, merge_x)
        self.convs[name] = L
        self.store.add_conv(name + ".weight", cout, cin, k)
        if bias:
            self.store.add_vec(name + ".bias", cout)
        return L

    def _bn(self, name, c) -> BnL:
        off = sum(b.c for b in self.bns.values())
        L = BnL(name, c, off)
        self.bns[name] = L
        self.store.add_vec(name + ".weight", c)
        self.store.add_vec(name + ".bias", c)
        return L

    def _declare(self):
        if self.backbone_kind == "ssd_vgg":
            self._declare_ssd()
        else:
            self._declare_resnet_fpn()
        self._declare_head_lstm()

    def _declare_ssd(self):
        """SSD300-VGG16 trunk (ssd_vgg.py:117-171): registration order vgg, fproj1-3, extras, loc, conf.  The loc/conf
        multibox heads are created but never used by SSD.forward; they are kept so reference checkpoints load."""
        e = "backbone.encoder."
        self.block_kind, self.nblocks, self.blocks = "vgg", (), []
        self.vgg_layers = []
        cin, idx = 3, 0
        for v in VGG_BASE:
            if v in ("M", "C"):
                self.vgg_layers.append(("pool", idx, 2, 2, 0, v == "C"))
                idx += 1
            else:
                self._conv(f"{e}vgg.{idx}", cin, v, 3, 1, 1, bias=True, merge_x=(cin == 3))
                self.vgg_layers.append(("conv", idx))
                idx += 2
                cin = v
        self.vgg_layers.append(("pool", idx, 3, 1, 1, False))
        idx += 1
        self._conv(f"{e}vgg.{idx}", 512, 1024, 3, 1, 6, 6, bias=True)
        self.vgg_layers.append(("conv", idx))
        idx += 2
        self._conv(f"{e}vgg.{idx}", 1024, 1024, 1, 1, 0, bias=True)
        self.vgg_layers.append(("conv", idx))
        self._conv(e + "fproj1", 512, 256, 1, bias=True)
        self._conv(e + "fproj2", 1024, 256, 1, bias=True)
        self._conv(e + "fproj3", 512, 256, 1, bias=True)
        cin, flag, k = 1024, False, 0
        for i, v in enumerate(SSD_EXTRAS):
            if cin != "S":
                if v == "S":
                    self._conv(f"{e}extras.{k}", cin, SSD_EXTRAS[i + 1], (1, 3)[flag], 2, 1, bias=True)
                else:
                    self._conv(f"{e}extras.{k}", cin, v, (1, 3)[flag], 1, 0, bias=True)
                k += 1
                flag = not flag
            cin = v
        self.n_extras = k
        for nm, mult in (("loc", 4), ("conf", 21)):
            for i, (c, nb) in enumerate(zip((512, 1024, 512, 256, 256, 256), (4, 6, 6, 6, 4, 4))):
                self._conv(f"{e}{nm}.{i}", c, nb * mult, 3, 1, 1, bias=True)

    def _declare_resnet_fpn(self):
        kind, nblocks = ARCHS[self.arch]
        self.block_kind, self.nblocks = kind, nblocks
        exp = 4 if kind == "bottleneck" else 1
        e = "backbone.encoder."
        self._conv(e + "conv1", 3, 64, 7, 2, 3, merge_x=True)
        self._bn(e + "bn1", 64)
        inpl = 64
        self.blocks = []
        for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), nblocks), start=1):
            for bi in range(nb):
                stride = 2 if (bi == 0 and li > 1) else 1
                q = f"{e}layer{li}.{bi}."
                blk = dict(prefix=q, stride=stride, layer=li, last=(bi == nb - 1), ds=False)
                if kind == "bottleneck":       # stride on the 3x3 (torchvision v1.5 == fpn_resnet.py:73-74)
                    self._conv(q + "conv1", inpl, planes, 1)
                    self._bn(q + "bn1", planes)
                    self._conv(q + "conv2", planes, planes, 3, stride, 1)
                    self._bn(q + "bn2", planes)
                    self._conv(q + "conv3", planes, planes * 4, 1)
                    self._bn(q + "bn3", planes * 4)
                else:
                    self._conv(q + "conv1", inpl, planes, 3, stride, 1)
                    self._bn(q + "bn1", planes)
                    self._conv(q + "conv2", planes, planes, 3, 1, 1)
                    self._bn(q + "bn2", planes)
                if bi == 0 and (stride != 1 or inpl != planes * exp):
                    self._conv(q + "downsample.0", inpl, planes * exp, 1, stride, 0)
                    self._bn(q + "downsample.1", planes * exp)
                    blk["ds"] = True
                inpl = planes * exp
                self.blocks.append(blk)
        c3, c4, c5 = 128 * exp, 256 * exp, 512 * exp
        f = "backbone.fpn."          # registration order of FPN_backbone.__init__, fpn_resnet.py:123-152
        self._conv(f + "P7_2", 256, 256, 3, 2, 1, bias=True)
        self._conv(f + "P6", c5, 256, 3, 2, 1, bias=True)
        self._conv(f + "P5_1", c5, 256, 1, 1, 0, bias=True)
        self._conv(f + "P5_2", 256, 256, 3, 1, 1, bias=True)
        self._conv(f + "P4_1", c4, 256, 1, 1, 0, bias=True)
        self._conv(f + "P4_2", 256, 256, 3, 1, 1, bias=True)
        self._conv(f + "P3_1", c3, 256, 1, 1, 0, bias=True)
        self._conv(f + "P3_2", 256, 256, 3, 1, 1, bias=True)

    def _declare_head_lstm(self):
        # shared head (the paper's configuration) or separate attention / box heads (mdl.py:211-225)
        heads = [("att_reg_box", 5)] if self.same_atb else [("att_box", 1), ("reg_box", 4)]
        for prefix, ncls in heads:
            self._conv(prefix + ".0.0", self.start_dim_head, 256, 3, 1, 1, bias=True)
            for i in range(1, 5):
                self._conv(f"{prefix}.{i}.0", 256, 256, 3, 1, 1, bias=True)
            self._conv(prefix + ".5", 256, ncls * self.n_anchors, 3, 1, 1, bias=True)
        H4 = 4 * self.lstm_dim
        for suf in ([""] + (["_reverse"] if self.bid else [])):       # nn.LSTM parameter order
            self.store.add_mat("lstm.weight_ih_l0" + suf, H4, self.emb_dim)
            self.store.add_mat("lstm.weight_hh_l0" + suf, H4, self.lstm_dim)
            self.store.add_vec("lstm.bias_ih_l0" + suf, H4)
            self.store.add_vec("lstm.bias_hh_l0" + suf, H4)

    def _register(self):
        self._param_names = list(self.store.order)
        for name in self._param_names:
            register_named(self, name, self.store.view(name))
        for i, (name, L) in enumerate(self.bns.items()):
            register_named(self, name + ".running_mean", self._rm[L.index:L.index + L.c], buffer=True)
            register_named(self, name + ".running_var", self._rv[L.index:L.index + L.c], buffer=True)
            register_named(self, name + ".num_batches_tracked", self._nbt[i], buffer=True)

    def _rebind(self):
        """Re-point every Parameter / buffer at the (possibly moved) flat storages."""
        mods = dict(self.named_modules())
        for name in self._param_names:
            path, leaf = name.rsplit(".", 1)
            p = mods[path]._parameters[leaf]
            p.data = self.store.view(name)
            p.grad = None
        for i, (name, L) in enumerate(self.bns.items()):
            m = mods[name]
            m._buffers["running_mean"] = self._rm[L.index:L.index + L.c]
            m._buffers["running_var"] = self._rv[L.index:L.index + L.c]
            m._buffers["num_batches_tracked"] = self._nbt[i]
        self._plans = {}

    def _apply(self, fn, recurse=True):
        flat = fn(self.store.flat)
        if flat.dtype != torch.float32:
            raise TypeError("zsgnet-pytorch_amd computes in fp32 only (fp32 MFMA); dtype conversion is not supported")
        self.store.flat = flat
        self.store.grad = torch.zeros_like(flat)
        self._rmv = fn(self._rmv)
        nb = self._rmv.numel() // 2
        self._rm, self._rv = self._rmv[:nb], self._rmv[nb:]
        self._nbt = self._nbt.to(flat.device)
        self._rebind()
        return self

    @property
    def device(self):
        return self.store.flat.device

    @torch.no_grad()
    def reset_parameters(self, seed: Optional[int] = None):
        """Random init (no network access for torchvision's ImageNet weights, mdl.py:411): He-normal encoder convs
        (fpn_resnet.py:266-272), PyTorch-default FPN/head convs, U(+-1/sqrt(H)) LSTM, head bias [0,0,0,0,-4]*A
        (mdl.py:214-219)."""
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        for name in self._param_names:
            p = self.store.view(name)
            e = self.store.entries[name]
            if e.kind == "conv":
                co, ci, k, _ = e.shape
                if name.startswith("backbone.encoder."):
                    p.copy_(torch.randn(e.shape, generator=g) * math.sqrt(2.0 / (k * k * co)))
                else:
                    bound = 1.0 / math.sqrt(ci * k * k)
                    p.copy_((torch.rand(e.shape, generator=g) * 2 - 1) * bound)
            elif name.startswith("lstm."):
                bound = 1.0 / math.sqrt(self.lstm_dim)
                p.copy_((torch.rand(e.shape, generator=g) * 2 - 1) * bound)
            elif name.endswith(".bias") and name[:-5] in self.convs:
                L = self.convs[name[:-5]]
                bound = 1.0 / math.sqrt(L.cin * L.k * L.k)
                p.copy_((torch.rand(e.shape, generator=g) * 2 - 1) * bound)
            elif name.endswith(".weight"):       # BN gamma
                p.fill_(1.0)
            else:                                # BN beta
                p.zero_()
        if self.same_atb:                        # final biases, mdl.py:214-225
            hb = torch.zeros(5 * self.n_anchors)
            hb[4::5] = -4.0
            self.store.view("att_reg_box.5.bias").copy_(hb)
        else:
            self.store.view("att_box.5.bias").fill_(-4.0)
            self.store.view("reg_box.5.bias").zero_()

    def join_grads(self):
        """After a backward with FusedAdam attached the main stream has not yet joined the side stream's last weight gradients
        (FusedAdam.step does, after updating everything else under them): anything ELSE that reads or writes gradients on the
        current stream joins here."""
        ov = getattr(self, "_adam_overlap", None)
        if ov is not None:
            torch.cuda.current_stream().wait_stream(ov[1])
            self._adam_overlap = None

    def join_weight_readers(self):
        """Called before anything WRITES the flat weight buffer on the current stream (optimizer step, load_state_dict):
        a training forward that was never back-propagated (metrics-only forward, discarded loss) leaves its backward weight
        preparation running on the side stream, reading the weights — make the current stream wait for it."""
        self.join_grads()
        for plan in self._plans.values():
            if plan._prep_pending:
                torch.cuda.current_stream().wait_event(plan._prep_ev)
                plan._prep_pending = False

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """Accepts reference checkpoints: strips DDP's 'module.' prefix (utils.py:489) and tolerates torchvision's
        unused `backbone.encoder.fc.*`."""
        if self.device.type == "cuda":
            self.join_weight_readers()
        sd = {}
        for k, v in state_dict.items():
            k = k[7:] if k.startswith("module.") else k
            if k.startswith("backbone.encoder.fc.") or k.startswith("backbone.encoder.fpn."):
                continue
            sd[k] = v
        return super().load_state_dict(sd, strict=strict, **kw)

    # ------------------------------------------------------------------------------------------------------
    # forward / backward
    # ------------------------------------------------------------------------------------------------------
    def refine_tuning(self, inp: Dict[str, Any], steps: int = 9, log=None) -> dict:
        """In-step refinement of the autotuner's near-ties for the training plan of `inp`'s geometry (ops.refine_in_step): alternatives
        the tuner timed within a few percent of its winner ALONE on the GPU are tried in the real step — forward + backward of the
        lowered plan on both streams, a fixed incoming gradient, no optimizer — and kept when the step gets faster.  Only shapes this
        process tuned itself have alternatives (a stamp-matched shipped table has none: nothing happens).  Not under DDP (rank 0's
        choices are broadcast before any rank lowers, dist._sync_tuning).  Returns ops.TUNE_INFO['refined']."""
        from . import ops as _ops
        assert self.training, "refine_tuning measures the training step"
        out = self(inp)                                   # lowers (and tunes) the plan if it is new
        plan = self._plan_for(*self.plan_geometry(inp))
        g = torch.randn(out["att_bbx_out"].shape, device=out["att_bbx_out"].device) * 1e-3
        out["att_bbx_out"].backward(g)

        def measure():
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            o = self(inp)
            o["att_bbx_out"].backward(g)
            ev[0].record()
            for i in range(steps):
                o = self(inp)
                o["att_bbx_out"].backward(g)
                ev[i + 1].record()
            torch.cuda.synchronize()
            t = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
            return t[len(t) // 2]
        res = _ops.refine_in_step(plan._tunables, measure, log=log)
        for p in self.parameters():
            p.grad = None
        return res

    def lstm_init_hidden(self, bs: int):
        """Reference mdl.py:279-294: two CPU draws (hidden_a then hidden_b) per forward, train and eval."""
        n = 2 if self.bid else 1
        if self.lstm_state == "zeros":
            return torch.zeros(n, bs, self.lstm_dim), torch.zeros(n, bs, self.lstm_dim)
        return torch.randn(n, bs, self.lstm_dim), torch.randn(n, bs, self.lstm_dim)

    def _ordered_params(self) -> List[nn.Parameter]:
        """Parameters in flat-storage order (== the order run_backward returns gradients in)."""
        if getattr(self, "_plist", None) is None:
            mods = dict(self.named_modules())
            self._plist = [mods[n.rsplit(".", 1)[0]]._parameters[n.rsplit(".", 1)[1]] for n in self._param_names]
        return self._plist

    def plan_geometry(self, inp: Dict[str, Any]) -> Tuple[int, int, int, int]:
        """(B, H, W, T_plan) of the launch plan forward(inp) will use (T is bucketed: see forward)"""
        img = inp["img"]
        if img.dtype == torch.uint8:
            B, H, W, _ = img.shape
        else:
            B, _, H, W = img.shape
        T = inp["qvec"].shape[1]
        return B, H, W, (20 if T <= 20 else (50 if T <= 50 else T))

    def _plan_for(self, B, H, W, T) -> "_Plan":
        key = (B, H, W, T, self.training)
        if key not in self._plans:
            self._plans[key] = _Plan(self, B, H, W, T, self.training)
        return self._plans[key]

    def forward(self, inp: Dict[str, Any]) -> Dict[str, Any]:
        require_gpu()
        img, qvec, qlens = inp["img"], inp["qvec"], inp["qlens"]
        if img.device.type != "cuda" or self.device.type != "cuda":
            raise RuntimeError("ZSGNet.forward needs the model and the batch on the MI355X (no CPU fallback)")
        # (img may be uint8 [B, H, W, 3] as PIL decodes (dat_loader gpu_normalise): /255 then happens on the GPU.)
        # The collater cuts qvec to the longest query of the batch (dat_loader.py:187-196), so T changes from batch to batch;
        # a launch plan (and its buffers) is built per geometry, so T is bucketed: the plan processes T_plan >= T tokens of
        # zero-padded input — the LSTM kernels stop at each query's own length, so the result does not depend on T_plan.
        B, H, W, Tp = self.plan_geometry(inp)
        plan = self._plan_for(B, H, W, Tp)
        if "h0" in inp:
            h0, c0 = inp["h0"], inp["c0"]
        else:
            h0, c0 = self.lstm_init_hidden(B)
        # ONE differentiable input ties the custom Function into autograd: run_backward writes the parameter gradients
        # straight into the flat gradient buffer that every p.grad views, so autograd has nothing to accumulate per
        # parameter (161 AccumulateGrad nodes cost the host ~0.5 ms per step with the GPU idle at the end of backward)
        if self._anchor is None or self._anchor.device != img.device:
            self._anchor = torch.zeros(1, device=img.device, requires_grad=True)
        plan.expect_backward = torch.is_grad_enabled()      # (grad mode is off inside autograd.Function.forward: decide here)
        out5 = _NetFn.apply(self, plan, img, qvec, qlens, h0, c0, self._anchor)
        if plan.training:
            out5._zsg_g5 = plan.g5_in           # where the loss may write d(loss)/d(out5) directly (loss._LossFn.forward): no copy, no multiply
            out5._zsg_plan = plan
        return dict(att_out=out5[..., 4:5], bbx_out=out5[..., :4], feat_sizes=plan.feat_sizes_t,
                    num_f_out=plan.num_f_out_t, att_bbx_out=out5)


class _NetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, plan, img, qvec, qlens, h0, c0, anchor):
        ctx.net, ctx.plan = net, plan
        out = plan.run_forward(img, qvec, qlens, h0, c0)
        ctx.fwd_id = plan.fwd_id
        return out

    @staticmethod
    def backward(ctx, g5):
        # a plan owns ONE set of activation buffers: the backward must belong to the plan's latest forward
        if ctx.fwd_id != ctx.plan.fwd_id:
            raise RuntimeError("backward of a forward whose activations were overwritten: this geometry's plan ran another "
                               "forward since (one backward per forward per input geometry)")
        ctx.plan.run_backward(g5)
        return (None,) * 8


class _Plan:
    """Static lowering of ZSGNet for one (B, H, W, T, training) geometry."""

    def __init__(self, net: ZSGNet, B: int, H: int, W: int, T: int, training: bool):
        self.net, self.B, self.H, self.W, self.T, self.training = net, B, H, W, T, training
        self.dev = net.device
        apply_main_priority_env()
        self.fwd = Program("fwd")
        self.prep = Program("bwd-prep")
        self._prep_stream, self._prep_ev, self._prep_fwd, self._prep_pending = None, None, -1, False
        self._bwd_fwd = -2               # fwd_id of the forward whose backward ran last (a second backward re-runs the preparation)
        self._rel_ev = torch.cuda.Event()
        self._hc_pin, self._hc_ev = None, None      # pinned ring of host-drawn LSTM states (run_forward)
        self._prep_idx_v = False
        self._out_slots_v = False
        self._adam_ev, self._adam_cut_v = None, False
        self.expect_backward = False
        self._tunables = []              # convolution descriptors of this plan, as lowered (ops.refine_in_step)
        self._wg_log = []                # (launch index, descriptor, src, dy, gradient view, parameter, name) of every convolution weight gradient
        self.g5_from_loss = None         # (fwd_id, scale): the loss kernel wrote d(loss)/d(out5) x scale into g5_in for that forward
        self.bwd = Program("bwd")
        self.bwd.side_batch = 3 if B * H * W <= (4 << 20) else 1      # (ops.SIDE_BATCH: markers vs overlap, measured)
        self.bwd.side_defer = 1      # (ops.SIDE_DEFER; round 5, with the main chain at wave priority 3: 1 wins at every size)
        self.tape = []
        self.acts: Dict[str, Act] = {}
        self.bytes = 0
        self.wt: Dict[str, torch.Tensor] = {}
        self.grad_ready: Dict[str, int] = {}
        self.reducer = None
        self.fwd_id = 0
        self.wt_jobs, self.wt_arena_used = [], 0
        self.wt_arena = self._buf(net.store.total + 4096 * len(net.convs)) if training else None
        self.fold_jobs, self.fold_used, self.fold_rows = [], 0, 0          # eval: BatchNorm folded into the convolutions
        self.fold_arena = self._buf(net.store.total + 8 * len(net.convs)) if (not training and net.bns) else None
        self.wino_jobs = {"fwd": WinoJobs(), "bwd": WinoJobs()}     # filter transforms of the Winograd convolutions (one launch each)
        # work that depends on the weights only / touches buffers nobody reads yet: runs on the side stream at the start of a training
        # forward (Winograd filter transforms, zero-fills of split-K outputs); the main stream waits for it right before the first
        # launch that needs any of it (_wait_idx): an event WAIT, no marker of its own
        self.prep_u, self._u_ev, self._wait_idx = Program("fwd-prep"), None, 1 << 30
        self._side_prep = training and os.environ.get("ZSG_U_ON_SIDE", "1") != "0"
        self._zero_calls = []
        self._lower()
        for i, c in enumerate(self.fwd.calls):          # (positions after the hoisting of the language maps)
            if any(c is z for z in self._zero_calls):
                self._wait_idx = min(self._wait_idx, i)
                break
        wj = self.wino_jobs["fwd"]
        if wj.jobs and self._side_prep:
            # U = G g G^T of every Winograd forward convolution depends on the weights only: in training it runs on the side stream
            # while the main stream converts the image and runs the stem (it was 47 us at the head of the forward's dependent chain)
            self.prep_u.add(lib.zsg_wino_weights, wj.finish(self.dev), len(wj.jobs), wj.blocks, what="wino filter transforms")
            self.prep_u.calls.insert(0, self.prep_u.calls.pop())          # (needed first)
            self.prep_u.lanes.insert(0, self.prep_u.lanes.pop())
            wfns = (lib.zsg_conv_wino, lib.zsg_conv_wino_bnstat)
            self._wait_idx = min(self._wait_idx, next(i for i, c in enumerate(self.fwd.calls) if c[0] in wfns))
        elif wj.jobs:        # eval (after the BatchNorm fold): first launch after the image conversion
            self.fwd.add(lib.zsg_wino_weights, wj.finish(self.dev), len(wj.jobs), wj.blocks, what="wino filter transforms")
            self.fwd.calls.insert(1, self.fwd.calls.pop())
            self.fwd.lanes.insert(1, self.fwd.lanes.pop())
        if training:
            self._finish_prep()
        elif self.fold_jobs:
            import struct
            blob = b"".join(struct.pack("<qqqqqiiii", *j) for j in self.fold_jobs)
            self.fold_jobs_dev = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.dev)

    # ---- allocation helpers --------------------------------------------------------------------------------
    def _buf(self, n, dtype=torch.float32):
        t = torch.zeros(int(n), dtype=dtype, device=self.dev)
        self.bytes += t.numel() * t.element_size()
        return t

    def act(self, name, B, H, W, C, ld=None, requires_grad=True) -> Act:
        ld = ld or C
        a = Act(self._buf(B * H * W * ld), B, C, ld, [Level(0, H, W, H * W * ld)], name)
        a.requires_grad = requires_grad
        self.acts[name] = a
        return a

    def packed(self, name, B, sizes, C, ld=None) -> Act:
        """pyramid levels packed level-major in one buffer"""
        ld = ld or C
        lv, off = [], 0
        for (h, w) in sizes:
            lv.append(Level(off, h, w, h * w * ld))
            off += B * h * w * ld
        a = Act(self._buf(off), B, C, ld, lv, name)
        self.acts[name] = a
        return a

    def like(self, a: Act, name=None) -> Act:
        if len(a.levels) == 1:           # a level of a packed buffer gets a compact twin, not a copy of the whole pack
            l = a.levels[0]
            return Act(self._buf(a.B * l.bstride), a.B, a.C, a.ld, [Level(0, l.H, l.W, l.bstride)], name or (a.name + ".grad"))
        g = Act(self._buf(a.buf.numel()), a.B, a.C, a.ld, a.levels, name or (a.name + ".grad"))
        return g

    def grad_of(self, a: Act) -> Act:
        if a.grad is None:
            a.grad = self.like(a)
        return a.grad

    @staticmethod
    def base(a: Act) -> torch.Tensor:
        """flat tensor starting at the first element of a (single-level or packed) activation"""
        return a.buf[a.levels[0].off:]

    def P(self, name):      # raw parameter storage
        return self.net.store.raw(name)

    def G(self, name):
        """gradient storage of a parameter; remembers the last backward launch that writes it (bucket readiness)"""
        self.grad_ready[name] = len(self.bwd.calls)
        return self.net.store.raw(name, self.net.store.grad)

    # ---- op lowering ---------------------------------------------------------------------------------------------
    def _tune(self, kind, fn, d, *a, **k):
        """autotune_conv + remember the descriptor: ops.refine_in_step may later switch its tile hint in place (ZSGNet.refine_tuning)"""
        r = autotune_conv(kind, fn, d, *a, **k)
        self._tunables.append(d)
        return r

    def _wino_u(self, src_ptr: int, N: int, Cred: int, row_ld: int, tap_ld: int, flip: int):
        """Transformed filter image of one 3x3 convolution (+ a one-off transform so that the tuner times real data);
        the job joins the program's batched transform only if the Winograd kernel wins the tuning."""
        # (not a plan buffer: when the direct kernel wins the tuning nobody keeps a reference and the image is freed; a Winograd
        # launch keeps it alive through its program's argument list.  The transform writes every element, padding included.)
        U = torch.empty(int(lib.zsg_wino_u_elems(Cred, N)), dtype=torch.float32, device=self.dev)
        job = (src_ptr, U.data_ptr(), N, Cred, row_ld, tap_ld, flip)
        one = WinoJobs()
        one.add(*job)
        one.finish(self.dev)
        one.launch(stream_ptr())
        return U, job

    def _ws_now(self):
        """BatchNorm / statistics workspace of the lane being lowered (forward launches on the side stream run concurrently
        with the main stream's, so they get their own)"""
        return self.ws if self._lane == 0 else self.ws_side

    def _join_side(self):
        """the main stream waits for everything lowered on the side stream so far (an empty lane-2 launch)"""
        if self.training:
            self.fwd.add(lib.zsg_memset_f32, self.ws, 0, 0.0, what="join side stream", lane=2)

    def on_side_stream(self):
        """context: forward launches lowered inside go to the side HIP stream (lane 1: each waits for everything enqueued
        on the main stream before it); the consumer of their results must be lowered with join=True (lane 2)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            if not self.training:          # (eval plans are short chains of folded convolutions: keep one stream)
                yield
                return
            keep, self._lane = self._lane, 1
            try:
                yield
            finally:
                self._lane = keep
        return cm()

    @staticmethod
    def _wino_chunks(d, B: int) -> int:
        tb = d.tile_hint & 0xff
        return sum((B * ((d.seg[i].src_H + 1) // 2) * ((d.seg[i].src_W + 1) // 2) + tb - 1) // tb for i in range(d.nseg))

    def conv(self, L: ConvL, src: Act, relu=False, out: Optional[Act] = None, name=None, bn_fuse: Optional[BnL] = None) -> Act:
        """bn_fuse: the output feeds a train-mode BatchNorm — let the epilogue emit the per-tile (sum, sum^2) partials
        (no extra pass over the activation) unless the autotuner chose split-K for this layer."""
        if out is None:
            lv = src.levels
            assert len(lv) == 1
            out = self.act(name or L.name, src.B, conv_out(lv[0].H, L.k, L.stride, L.pad, L.dil),
                           conv_out(lv[0].W, L.k, L.stride, L.pad, L.dil), L.cout)
        d = fwd_desc(src, out, L.cpad, L.cout, L.k, L.stride, L.pad, L.dil, wC=L.cpad, relu=relu, merge_x=L.merge_x)
        bias = self.P(L.name + ".bias") if L.bias else None
        rd = src
        ig_fn, wn_fn = lib.zsg_conv_igemm, lib.zsg_conv_wino
        # a split-K choice would cost this layer its fused BatchNorm statistics: a statistics pass over the output at
        # ~4 TB/s plus two more dependent launches
        pen = 0.0
        if bn_fuse is not None and self.training and not L.bias and not relu:
            pen = 0.008 + out.rows() * L.cout * 4 / 4e9
        wt = self.P(L.name + ".weight")
        pend = getattr(src, "pending", None)
        if pend is not None:
            return self._conv_bnpre(L, src, out, d, wt, pend, bn_fuse)
        wargs = None
        if wino_ok(L.k, L.stride, L.pad, L.dil) and not L.merge_x and wino_mode() != "0":
            U, job = self._wino_u(wt.data_ptr(), L.cout, L.cpad, L.k * L.k * L.cpad, L.cpad, 0)
            wargs = (rd.buf, U, out.buf, bias, None, None, None)
        self._tune("igemm", ig_fn, d, (rd.buf, wt, out.buf, bias, None, None, None), stream_ptr(),
                      split_penalty_ms=pen, wino_args=wargs, wino_fn=wn_fn)
        fn = ig_fn
        if d.use_wino:
            fn, wt = wn_fn, U
            self.wino_jobs["fwd"].add(*job)
        partials = None
        out.bn_chunks = 0
        if bn_fuse is not None and self.training and d.tile_hint and ((d.tile_hint >> 16) & 0xff) <= 1 and not L.bias and not relu:
            if d.use_wino:
                chunks = self._wino_chunks(d, src.B)
            else:
                chunks = igemm_partial_rows(d)
            if chunks * 2 * L.cout * 4 <= self.ws_bytes:
                partials, out.bn_chunks = self._ws_now(), chunks
        pre_zero = None
        if self._side_prep and ((d.tile_hint >> 16) & 0xff) > 1 and len(out.levels) == 1:
            # split-K: the K slices are added with atomics to a ZEROED output.  The zero-fill (a dependent ~6 us launch in front of the
            # convolution when the library issues it) goes to the side-stream preparation instead — nobody reads this buffer between
            # the previous step's backward and this launch — and the convolution is told to accumulate onto its own output.
            lv0 = out.levels[0]
            self.prep_u.add(lib.zsg_memset_f32, out.buf[lv0.off:], out.B * lv0.bstride, 0.0, what="zero:" + L.name)
            pre_zero = out.buf
        tail_n = -1
        if partials is not None and BN_TAIL in ("1", "fwd") and out.bn_chunks > BN_TAIL_MIN_ROWS:
            tail_n = int(lib.zsg_conv_bn_tail_tickets(_ct.byref(d), 1 if d.use_wino else 0))
        out.bn_inline = None
        if tail_n > 0:
            # the convolution's last tile per column block finalises the statistics itself (mean / invstd / running statistics are ready
            # when the launch ends): the plain apply launch follows, nothing in between
            Lb = bn_fuse
            tk = self._buf(tail_n, dtype=torch.int32)
            out.bn_mean, out.bn_invstd = self._buf(Lb.c), self._buf(Lb.c)
            rm, rv = self.net._rm[Lb.index:Lb.index + Lb.c], self.net._rv[Lb.index:Lb.index + Lb.c]
            self.fwd.add(lib.zsg_conv_wino_bnstat if d.use_wino else lib.zsg_conv_igemm_bnstat, d, rd.buf, wt, out.buf, partials, tk,
                         out.bn_mean, out.bn_invstd, rm, rv, 0.1, 1e-5, what=L.name + "+bnstat", lane=self._lane)
        else:
            self.fwd.add(fn, d, rd.buf, wt, out.buf, bias, pre_zero, None, partials, what=L.name, lane=self._lane)
        if pre_zero is not None:
            self._zero_calls.append(self.fwd.calls[-1])
        if tail_n > 0:
            pass
        elif partials is not None and out.bn_chunks <= lib.zsg_bn_inline_max_chunks():
            # few partial rows: the BatchNorm apply launch (the very next launch on this stream: the workspace is still intact)
            # reduces them itself — no finalize launch
            out.bn_inline = partials
            out.bn_mean, out.bn_invstd = self._buf(bn_fuse.c), self._buf(bn_fuse.c)
        elif partials is not None:         # finalize at once: the shared workspace is reused by the next launch
            Lb = bn_fuse
            rows = sum(src.B * d.seg[i].rows_y * d.seg[i].rows_x for i in range(d.nseg))
            out.bn_mean, out.bn_invstd = self._buf(Lb.c), self._buf(Lb.c)
            rm, rv = self.net._rm[Lb.index:Lb.index + Lb.c], self.net._rv[Lb.index:Lb.index + Lb.c]
            self.fwd.add(lib.zsg_bn_stats_from_partials, partials, out.bn_chunks, rows, Lb.c, out.bn_mean, out.bn_invstd, rm, rv, 0.1, 1e-5,
                         what="stats:" + Lb.name, lane=self._lane)
        out.needs_mask = relu
        # the first consumer lowered is the LAST to add to src's gradient in the backward: if src is a train-mode BatchNorm's
        # output, that data gradient completes the BatchNorm's dout and can carry its backward sums (see bn())
        completes = self.training and getattr(src, "bn_out", False) and not getattr(src, "_consumed", False)
        src._consumed = True
        self.tape.append(lambda: self._conv_bwd(L, src, out, completes_bn=completes))
        return out

    def _conv_bnpre(self, L: ConvL, src: Act, out: Act, d, wt, pend, bn_fuse: Optional[BnL]) -> Act:
        """conv() for a source whose closing BatchNorm + residual + ReLU is still pending (bn(defer=True)): this 1x1 convolution applies
        it in its operand loader and writes the activation (src.buf) and its ReLU bits itself — zsg_conv_igemm_bnpre.  Everything
        behind the launch (this convolution's own fused statistics, the tape) is conv()'s."""
        assert self.training and L.k == 1 and L.stride == 1 and L.pad == 0 and not L.bias and bn_fuse is not None and len(src.levels) == 1, \
            "a deferred BatchNorm apply needs a plain 1x1 consumer in front of a BatchNorm"
        x = pend["x"]
        fn = lib.zsg_conv_igemm_bnpre
        pre = (pend["mean"], pend["invstd"], pend["gam"], pend["bet"], pend["residual"].buf, src.buf, pend["rmask"])
        self._tune("igemm", fn, d, (x.buf, wt, out.buf, None, None, None, None, None, None, 0.1, 1e-5) + pre, stream_ptr())
        h = d.tile_hint
        admissible = (h and ((h >> 16) & 0xff) <= 1 and (h & 0xff) in (64, 128) and not (h >> 27) & 1 and not (h >> 28) & 3
                      and not ((h & 0xffff) == 0x8080 and not (h >> 24) & 1))
        if not admissible:
            # no tuner (ZSG_AUTOTUNE=0, heuristic hint 0) or a cached / shipped entry the BatchNorm-applying loader has no variant for
            # (split-K, 64-deep K tiles, stream-K, the streaming 1x1 kernel, the 4-wave 128x128 tile): the 64x64 tile always applies —
            # lowering must not crash on a table it did not make
            d.tile_hint = tile_hint(64, 64, 1)
        lane = 2 if pend["join"] else self._lane
        chunks = igemm_partial_rows(d)
        assert chunks * 2 * L.cout * 4 <= self.ws_bytes
        partials, out.bn_chunks = self._ws_now(), chunks
        tail_n = -1
        if BN_TAIL in ("1", "fwd") and chunks > BN_TAIL_MIN_ROWS:
            tail_n = int(lib.zsg_conv_bn_tail_tickets(_ct.byref(d), 0))
        Lb = bn_fuse
        rm, rv = self.net._rm[Lb.index:Lb.index + Lb.c], self.net._rv[Lb.index:Lb.index + Lb.c]
        out.bn_inline = None
        what = L.name + "+bnpre(" + pend["name"] + ")"
        if tail_n > 0:
            tk = self._buf(tail_n, dtype=torch.int32)
            out.bn_mean, out.bn_invstd = self._buf(Lb.c), self._buf(Lb.c)
            self.fwd.add(fn, d, x.buf, wt, out.buf, partials, tk, out.bn_mean, out.bn_invstd, rm, rv, 0.1, 1e-5, *pre, what=what + "+bnstat", lane=lane)
        else:
            self.fwd.add(fn, d, x.buf, wt, out.buf, partials, None, None, None, None, None, 0.1, 1e-5, *pre, what=what, lane=lane)
            out.bn_mean, out.bn_invstd = self._buf(Lb.c), self._buf(Lb.c)
            if out.bn_chunks <= lib.zsg_bn_inline_max_chunks():
                out.bn_inline = partials
            else:
                rows = src.B * d.seg[0].rows_y * d.seg[0].rows_x
                self.fwd.add(lib.zsg_bn_stats_from_partials, partials, out.bn_chunks, rows, Lb.c, out.bn_mean, out.bn_invstd, rm, rv, 0.1, 1e-5,
                             what="stats:" + Lb.name, lane=self._lane)
        src.pending = None
        out.needs_mask = False
        completes = self.training and getattr(src, "bn_out", False) and not getattr(src, "_consumed", False)
        src._consumed = True
        self.tape.append(lambda: self._conv_bwd(L, src, out, completes_bn=completes))
        return out

    def conv_bn(self, L: ConvL, Lb: BnL, x: Act, relu: bool, residual: Optional[Act] = None, name=None, yname=None) -> Act:
        """conv -> BatchNorm [-> + residual] [-> ReLU].  Training: the two lowered ops (batch statistics from the conv
        epilogue).  Eval: ONE convolution with the BatchNorm folded into its weights / bias (zsg_bn_fold refreshes the
        folded copies at the start of every eval forward), residual add and ReLU in its epilogue."""
        if self.training:
            y = self.conv(L, x, name=yname or (L.name + ".y"), bn_fuse=Lb)
            return self.bn(Lb, y, relu, residual=residual, name=name)
        net = self.net
        n_w = L.cout * L.k * L.k * L.cpad
        w_off = self.fold_used
        b_off = w_off + (n_w + 3) // 4 * 4
        self.fold_used = b_off + (L.cout + 3) // 4 * 4
        assert self.fold_used <= self.fold_arena.numel(), "BN-fold arena too small"
        ents = net.store.entries
        self.fold_jobs.append((ents[L.name + ".weight"].offset, w_off, ents[Lb.name + ".weight"].offset, ents[Lb.name + ".bias"].offset,
                               b_off, self.fold_rows, L.cout, L.k * L.k * L.cpad, Lb.index))
        self.fold_rows += L.cout
        lv = x.levels[0]
        out = self.act(name or Lb.name, x.B, conv_out(lv.H, L.k, L.stride, L.pad, L.dil), conv_out(lv.W, L.k, L.stride, L.pad, L.dil), L.cout)
        d = fwd_desc(x, out, L.cpad, L.cout, L.k, L.stride, L.pad, L.dil, wC=L.cpad, relu=relu, merge_x=L.merge_x)
        wt, bias = self.fold_arena[w_off:w_off + n_w], self.fold_arena[b_off:b_off + L.cout]
        args = (x.buf, wt, out.buf, bias, residual.buf if residual is not None else None, None, None)
        wargs = None
        if wino_ok(L.k, L.stride, L.pad, L.dil) and not L.merge_x and wino_mode() != "0":
            U, job = self._wino_u(wt.data_ptr(), L.cout, L.cpad, L.k * L.k * L.cpad, L.cpad, 0)
            wargs = (x.buf, U) + args[2:]
        self._tune("igemm", lib.zsg_conv_igemm, d, args, stream_ptr(), wino_args=wargs)
        if d.use_wino:
            self.wino_jobs["fwd"].add(*job)
            self.fwd.add(lib.zsg_conv_wino, d, *wargs, what=L.name + "+bn")
        else:
            self.fwd.add(lib.zsg_conv_igemm, d, *args, what=L.name + "+bn")
        return out

    def _wt(self, L: ConvL, cred: int) -> torch.Tensor:
        """transposed weight image [cpad][k*k][cred] for the data gradient (a slice of one arena); all images are
        refreshed by ONE batched transpose launch at the start of backward (built in _finish_prep)"""
        if L.name not in self.wt:
            n = L.cpad * L.k * L.k * cred
            off = self.wt_arena_used
            self.wt_arena_used += (n + 3) // 4 * 4
            assert self.wt_arena_used <= self.wt_arena.numel(), "dgrad weight arena too small"
            self.wt[L.name] = self.wt_arena[off:off + n]
            e = self.net.store.entries[L.name + ".weight"]
            self.wt_jobs.append((e.offset, off, L.cout, L.k * L.k, L.cpad, cred))
            # fill the image once now: the autotuner that follows times the data gradient (and the Winograd filter transform) on real
            # weights, not on the arena's zeros (zero operands clock ~20 % higher: MI355X_MICROARCH.md, DVFS)
            check(lib.zsg_transpose_w(self.net.store.flat.data_ptr() + 4 * e.offset, self.wt[L.name].data_ptr(), L.cout, L.k * L.k, L.cpad, cred,
                                      stream_ptr()), "transpose_w")
        return self.wt[L.name]

    def _finish_prep(self):
        import struct
        if not self.wt_jobs:
            return
        blob, tile0 = b"", 0
        for (so, do, N, T, Cc, ld) in self.wt_jobs:
            tc, tn = (Cc + 63) // 64, (ld + 63) // 64          # 64 x 64 tiles (csrc/misc.hip transpose_w_batched_kernel)
            assert Cc % 4 == 0 and ld % 4 == 0 and so % 4 == 0 and do % 4 == 0
            blob += struct.pack("<qqiiiiiiii", so, do, N, T, Cc, ld, tile0, tc, tn, 0)
            tile0 += T * tc * tn
        self.wt_jobs_dev = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(self.dev)
        self.prep.add(lib.zsg_transpose_w_batched, self.net.store.flat, self.wt_arena, self.wt_jobs_dev, len(self.wt_jobs), tile0,
                      what="transpose all dgrad weight images")
        wj = self.wino_jobs["bwd"]
        if wj.jobs:          # rotated filter transforms of the Winograd data gradients, from the transposed images
            self.prep.add(lib.zsg_wino_weights, wj.finish(self.dev), len(wj.jobs), wj.blocks, what="wino dgrad filter transforms")

    def _conv_bwd(self, L: ConvL, src: Act, out: Act, dy: Optional[Act] = None, completes_bn: bool = False):
        dy = dy or out.grad
        if dy is None:
            return
        dw = fwd_desc(src, dy, L.cpad, L.cout, L.k, L.stride, L.pad, L.dil, wC=L.cpad)
        self.wgrad(dw, src, dy, L.name + ".weight", "wgrad:" + L.name)
        if L.bias:
            base = dy.levels[0].off
            self.bwd.add(lib.zsg_colsum, dy.buf[base:], 1, 0, dy.rows(), dy.ld, 0, L.cout, self.G(L.name + ".bias"), 1,
                         what="bgrad:" + L.name, lane=1)
        if src.requires_grad:
            self.dgrad(L, dy, src, n=L.cpad, completes_bn=completes_bn)

    def wgrad(self, d, src: Act, dy: Act, pname: str, what: str):
        """weight gradient of one parameter, accumulated into the flat gradient buffer (every parameter has exactly one
        wgrad launch per backward; the shared head's pyramid levels are segments of that one launch).  The autotuner's
        trial launches write a scratch image, never the gradient buffer."""
        gw = self.G(pname)
        args = (src.buf, dy.buf, gw, 1, self.wg_ws, self.wg_ws_bytes)
        targs = (src.buf, dy.buf, self.tune_dw, 0, self.wg_ws, self.wg_ws_bytes)
        s0 = d.seg[0]
        wino = (d.wR == 3 and d.wS == 3 and s0.sy == 1 and s0.ty.d0 == -1 and s0.ty.dstep == 1 and not d.merge_x
                and dy.ld % 4 == 0 and wino_mode() != "0")       # 3x3 / stride 1 / pad 1: Winograd F(3x3,2x2) candidates
        self._tune("wgrad", lib.zsg_conv_wgrad, d, targs, stream_ptr(), self.wg_ws_bytes, wino_args=targs if wino else None)
        self.bwd.add(lib.zsg_conv_wgrad_wino if d.use_wino else lib.zsg_conv_wgrad, d, *args, what=what, lane=1)
        self._wg_log.append((len(self.bwd.calls) - 1, d, src, dy, gw, pname, what))

    def dgrad(self, L: ConvL, dy: Act, src: Act, n: int, row0: int = 0, dx: Optional[Act] = None, completes_bn: bool = False):
        """dx (+)= dgrad(dy) for input channels [row0, row0+n) of L; applies src's ReLU mask when required."""
        cred = dy.ld
        assert cred % 4 == 0
        wt = self._wt(L, cred)
        dx = dx or self.grad_of(src)
        d = dgrad_desc(dy, dx, cred, n, L.k, L.stride, L.pad, L.dil)
        wt_off = row0 * L.k * L.k * cred
        if d.zero_fill and not dx.gfilled:      # stride-parity classes without taps get no launch: clear them
            # (a gradient buffer is idle from the previous step's optimizer update until this backward: with the side-stream
            # preparation on, the fill runs there during the forward instead of on the backward's dependent chain)
            (self.prep if self._side_prep else self.bwd).add(lib.zsg_memset_f32, self.base(dx), sum(dx.B * l.H * l.W * dx.ld for l in dx.levels), 0.0,
                                                             what="zero:" + L.name)
            dx.gfilled = True
        mask = None
        if src.needs_mask:
            # the epilogue indexes the mask with the OUTPUT offsets; dx may be a level of a packed buffer while src has
            # its own: rebase the pointer so that mask[o] == src[o - (dx_off - src_off)]
            deltas = {sl.off - dl.off for sl, dl in zip(src.levels, dx.levels)}
            assert len(deltas) == 1 and src.ld == dx.ld, "ReLU mask needs one common offset between src and its gradient"
            mask = src.buf.data_ptr() + 4 * deltas.pop()
            self.bwd.keep.append(src.buf)
        args = (dy.buf, wt[wt_off:], dx.buf, None, dx.buf if dx.gfilled else None, mask, None)
        wargs = None
        if wino_ok(L.k, L.stride, L.pad, L.dil) and wino_mode() != "0":
            U, job = self._wino_u(wt.data_ptr() + 4 * wt_off, n, cred, L.k * L.k * cred, cred, 1)
            wargs = (dy.buf, U) + args[2:]
        # a split-K choice would cost the BatchNorm below its fused backward sums: a pass over dout and x plus a launch
        pen = (0.006 + 2 * dx.rows() * n * 4 / 4e9) if (completes_bn and BNB_FUSE and not d.zero_fill and mask is None) else 0.0
        self._tune("igemm", lib.zsg_conv_igemm, d, args, stream_ptr(), split_penalty_ms=pen, wino_args=wargs, allow_sk=SK_BWD)
        if self._side_prep and ((d.tile_hint >> 16) & 0xff) > 1 and args[4] is None and len(dx.levels) == 1:
            # split-K into a gradient buffer nothing has written yet: its zero-fill moves to the side-stream preparation (see conv())
            lv0 = dx.levels[0]
            self.prep.add(lib.zsg_memset_f32, dx.buf[lv0.off:], dx.B * lv0.bstride, 0.0, what="zero:dgrad:" + L.name)
            args = args[:4] + (dx.buf,) + args[5:]
            if wargs is not None:
                wargs = wargs[:4] + (dx.buf,) + wargs[5:]
        if d.use_wino:
            self.wino_jobs["bwd"].add(*job)
            self.bwd.add(lib.zsg_conv_wino, d, *wargs, what="dgrad:" + L.name)
        else:
            self.bwd.add(lib.zsg_conv_igemm, d, *args, what="dgrad:" + L.name)
        dx.gfilled = True
        # If this launch turns out to be the one that COMPLETES a BatchNorm's dout (the BatchNorm's backward is lowered right
        # after it), its epilogue can also produce that BatchNorm's backward sums: remember how to re-issue it (see bn()).
        covers_all = not d.zero_fill and mask is None and ((d.tile_hint >> 16) & 0xff) <= 1 and d.tile_hint and dx.ld == n and n % 4 == 0
        dx.last_writer = (len(self.bwd.calls) - 1, d, wargs if d.use_wino else args, "dgrad:" + L.name) if covers_all else None

    def bn(self, L: BnL, x: Act, relu: bool, residual: Optional[Act] = None, name=None, join: bool = False, defer: bool = False) -> Act:
        """join: an operand (the residual) was produced on the side stream — the apply launch first joins it (lane 2).
        defer: no apply launch — the one consumer lowered next (a 1x1 convolution, conv()) applies this BatchNorm in its operand loader
        and materialises the activation (out.pending describes it)."""
        net = self.net
        lv = x.levels[0]
        out = self.act(name or L.name, x.B, lv.H, lv.W, L.c)
        rows = x.B * lv.H * lv.W
        fused = self.training and getattr(x, "bn_chunks", 0) > 0
        mean, invstd = (x.bn_mean, x.bn_invstd) if fused else (self._buf(L.c), self._buf(L.c))
        rm, rv = net._rm[L.index:L.index + L.c], net._rv[L.index:L.index + L.c]
        self.ws_need = max(getattr(self, "ws_need", 0), lib.zsg_bn_workspace_bytes(rows, L.c))
        gam, bet = self.P(L.name + ".weight"), self.P(L.name + ".bias")
        if fused:
            pass                          # statistics were finalized right after the producing convolution
        elif self.training:
            self.fwd.add(lib.zsg_bn_stats, x.buf, rows, L.c, mean, invstd, rm, rv, 0.1, 1e-5, self._ws_now(), self.ws_bytes, what=L.name, lane=self._lane)
        else:
            self.fwd.add(lib.zsg_bn_eval_stats, rm, rv, L.c, 1e-5, mean, invstd, what=L.name, lane=self._lane)
        rmask = self._buf((rows * L.c // 4 + 3) // 4) if (relu and self.training) else None     # 4 mask bits per byte
        lane = 2 if (join and self.training) else self._lane
        inl = getattr(x, "bn_inline", None) if fused else None
        defer = defer and self.training and relu and residual is not None and rmask is not None and rows * L.c * 4 >= BN_PRE_MIN_MB * 1e6
        if defer:
            if inl is not None:       # the statistics were left to the apply launch: finalise them now (the very next launch: workspace intact)
                self.fwd.add(lib.zsg_bn_stats_from_partials, inl, x.bn_chunks, rows, L.c, mean, invstd, rm, rv, 0.1, 1e-5, what="stats:" + L.name, lane=lane)
            out.pending = dict(x=x, mean=mean, invstd=invstd, gam=gam, bet=bet, residual=residual, rmask=rmask, join=join, name=L.name)
        elif inl is not None:
            self.fwd.add(lib.zsg_bn_apply_from_partials, x.buf, rows, L.c, inl, x.bn_chunks, gam, bet, residual.buf if residual is not None else None,
                         int(relu), out.buf, rmask, mean, invstd, rm, rv, 0.1, 1e-5, what=L.name, lane=lane)
        else:
            self.fwd.add(lib.zsg_bn_apply, x.buf, rows, L.c, mean, invstd, gam, bet, residual.buf if residual is not None else None,
                         int(relu), out.buf, rmask, what=L.name, lane=lane)

        out.bn_out = self.training

        def back():
            if out.grad is None:
                return
            dx = self.grad_of(x)
            lw = getattr(out.grad, "last_writer", None)
            fuse = (BNB_FUSE and lw is not None and lw[0] == len(self.bwd.calls) - 1 and self.bwd.lanes[lw[0]] == 0
                    and len(out.grad.levels) == 1 and out.grad.levels[0].off == 0 and x.levels[0].off == 0 and out.grad.ld == L.c and x.ld == L.c)
            if fuse:
                # re-issue the data gradient that has just completed dout with the BatchNorm-backward sums in its epilogue
                # (per-tile partial rows, reduced in a fixed order by the finalize launch: deterministic)
                idx, d, a, what = lw
                if d.use_wino:
                    chunks = self._wino_chunks(d, x.B)
                else:
                    chunks = igemm_partial_rows(d)
                fuse = chunks * 2 * L.c * 4 + 2 * L.c * 4 <= self.ws_bytes
            g_out, bits = None, rmask
            if residual is not None and residual.requires_grad:
                # Round 6: where the data gradient that completes dout carries this BatchNorm's backward sums anyway, it also STORES the
                # ReLU-masked gradient (zsg_conv_desc.epi_flags bit 0) — which IS the residual branch's gradient (out = relu(bn(x) +
                # residual)): the residual's gradient buffer becomes an alias of dout, and the apply pass below neither reads the mask nor
                # writes a second output (16 -> 12 bytes per element on the step's sixteen bn3 passes, 645 MB per step at configs[1]).
                alias = (fuse and MASKED_DOUT and relu and rmask is not None and residual.grad is None and len(residual.levels) == 1
                         and residual.levels[0].off == 0 and residual.ld == L.c and residual.C == L.c and residual.buf.numel() == out.grad.buf.numel())
                if alias:
                    residual.grad = out.grad
                    lw[1].epi_flags |= 1
                    bits = None
                else:
                    rg = self.grad_of(residual)
                    assert not rg.gfilled, "residual gradient must be produced first (tape order)"
                    g_out = rg.buf
                    rg.gfilled = True
            if fuse:
                part = self.ws[2 * L.c:]              # (the first 2C floats of the workspace: the finalize launch's coefficients)
                # a = (src, wt|U, out, bias=None, add_src, mask=None, partials=None)
                assert a[3] is None and a[5] is None and a[6] is None
                tail_n = int(lib.zsg_conv_bn_tail_tickets(_ct.byref(d), 1 if d.use_wino else 0)) if (BN_TAIL in ("1", "bwd") and chunks > BN_TAIL_MIN_ROWS) else -1
                if tail_n > 0:
                    # the data gradient's last tile per column block finalises the coefficients and d(gamma) / d(beta): the apply pass
                    # is all that is left of this BatchNorm's backward
                    fn = lib.zsg_conv_wino_bnb_tail if d.use_wino else lib.zsg_conv_igemm_bnb_tail
                    tk = self._buf(tail_n, dtype=torch.int32)
                    coef = self._buf(2 * L.c)
                    self.bwd.calls[idx] = (fn, marshal(fn, (d, a[0], a[1], a[2], a[4], x.buf, mean, invstd, rmask, part, tk, coef,
                                                            self.G(L.name + ".weight"), self.G(L.name + ".bias"), 1), self.bwd.keep), what + "+bnb+fin")
                    self.bwd.add(lib.zsg_bn_bwd_apply, self.base(out.grad), bits, x.buf, rows, L.c, mean, invstd, gam, coef, dx.buf, g_out,
                                 what="bnbwd:" + L.name)
                else:
                    fn = lib.zsg_conv_wino_bnb if d.use_wino else lib.zsg_conv_igemm_bnb
                    self.bwd.calls[idx] = (fn, marshal(fn, (d, a[0], a[1], a[2], a[4], x.buf, mean, invstd, rmask, part), self.bwd.keep), what + "+bnb")
                    self.bwd.add(lib.zsg_bn_backward_from_partials, self.base(out.grad), bits, x.buf, rows, L.c, mean, invstd, gam,
                                 dx.buf, g_out, self.G(L.name + ".weight"), self.G(L.name + ".bias"), 1, part, chunks, self.ws, self.ws_bytes,
                                 what="bnbwd:" + L.name)
            else:
                self.bwd.add(lib.zsg_bn_backward, self.base(out.grad), None, rmask, x.buf, rows, L.c, mean, invstd, gam,
                             dx.buf, g_out, self.G(L.name + ".weight"), self.G(L.name + ".bias"), 1, self.ws, self.ws_bytes,
                             what="bnbwd:" + L.name)
            dx.gfilled = True
        self.tape.append(back)
        return out

    # ---- the network ---------------------------------------------------------------------------------------------------
    def _lower(self):
        net, B, H, W, T = self.net, self.B, self.H, self.W, self.T
        C = net.convs
        BN = net.bns
        e = "backbone.encoder."
        # shared BN workspace: sized generously up-front (largest rows*C is the stem's conv output)
        H1, W1 = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
        self.ws_bytes = 32 << 20         # BN partials: >= zsg_bn_workspace_bytes for every layer, and the conv-epilogue partials
        self.ws = self._buf(self.ws_bytes // 4)
        self.ws_side = self._buf(self.ws_bytes // 4)     # the same for forward launches on the side stream (they run concurrently)
        self._lane = 0
        self.wg_ws_bytes = 256 << 20     # split-K slabs of the weight-gradient kernel (largest: 64 splits x 1.2 M weights)
        self.wg_ws = self._buf(self.wg_ws_bytes // 4)
        self.tune_dw = self._buf(max(e.size for e in net.store.entries.values()) + 64)

        # ---- static inputs ------------------------------------------------------------------------------------------
        self.in_qvec = self._buf(B * T * net.emb_dim)
        self.in_qlens = self._buf(B)
        nd = 2 if net.bid else 1
        self.in_hc = self._buf(2 * nd * B * net.lstm_dim)            # (h0 | c0: ONE host-to-device copy per step when both come from the host)
        self.in_h0, self.in_c0 = self.in_hc[:nd * B * net.lstm_dim], self.in_hc[nd * B * net.lstm_dim:]
        self.img_slot = len(self.fwd.calls)
        x0 = self.act("img_nhwc4", B, H, W, 4, requires_grad=False)
        self.fwd.add(lib.zsg_nchw_to_nhwc4, self.in_qvec, B, 3, H, W, x0.buf, what="img")     # src pointer patched per call

        # ---- query encoder ----------------------------------------------------------------------------------------------
        # The query encoder is independent of the image encoder: its launches go to the side stream (joined where the head
        # first reads `we`), and its backward is replayed right after the head's — not at the very end of the step.
        we, lstm_tape = None, []
        self._hoist = []
        if net.use_lang:
            t0 = len(self.tape)
            we = self._lower_lstm()
            lstm_tape = self.tape[t0:]
            del self.tape[t0:]
        hoist_to = len(self.fwd.calls)            # right behind the query encoder

        # ---- encoder ------------------------------------------------------------------------------------------------------
        # (the image-blind variants still run the encoder, as the reference does, mdl.py:363-375: the pyramid sizes and the
        # train-mode BatchNorm running statistics come from it; no gradient reaches it, so its backward emits nothing)
        feats: List[Act] = []
        if net.backbone_kind == "ssd_vgg":
            feats = self._lower_ssd(x0)
        else:
            H2, W2 = conv_out(H1, 3, 2, 1), conv_out(W1, 3, 2, 1)
            if self.training and os.environ.get("ZSG_STEM_FUSE", "1") != "0":
                x = self._lower_stem_fused(C[e + "conv1"], BN[e + "bn1"], x0, H1, W1, H2, W2)
            else:
                a = self.conv_bn(C[e + "conv1"], BN[e + "bn1"], x0, True, name="stem.a", yname="stem.y")
                x = self.act("pool", B, H2, W2, 64)
                idx = self._buf((B * H2 * W2 * 64 + 3) // 4)      # uint8 indices, stored in a float-sized buffer
                self.fwd.add(lib.zsg_maxpool_fwd, a.buf, B, H1, W1, 64, 3, 2, 1, H2, W2, x.buf, idx, what="maxpool")
                pool_in, pool_out = a, x

                def pool_back():
                    if pool_out.grad is None:
                        return
                    dx = self.grad_of(pool_in)
                    self.bwd.add(lib.zsg_maxpool_bwd, self.base(pool_out.grad), idx, B, H1, W1, 64, 3, 2, 1, H2, W2, dx.buf, what="maxpool_bwd")
                    dx.gfilled = True
                self.tape.append(pool_back)
            taps, lat = {}, {}
            early = self.training and os.environ.get("ZSG_FPN_LATERAL_EARLY", "1") != "0"
            for blk in net.blocks:
                # (a block that is not the last of its stage has ONE first reader, the next block's conv1: that convolution applies this block's
                # closing BatchNorm + residual + ReLU itself where the activation is large, see BN_PRE_MIN_MB)
                big = BN_PRE_MIN_MB > 0 and net.block_kind == "bottleneck" and self.training and not blk["last"]
                x = self._lower_block(blk, x, defer_out=big)
                if blk["last"]:
                    taps[blk["layer"]] = x
                    if early and blk["layer"] in (2, 3):
                        # the pyramid's lateral 1x1 convolutions P3_1 / P4_1 only read C3 / C4: on the side stream as soon as their
                        # input exists, under layer3 / layer4 (whose 19^2 / 10^2 launches leave a quarter of the CUs idle), instead
                        # of in the main stream's chain between layer4 and the head
                        with self.on_side_stream():
                            nm = "P3_1" if blk["layer"] == 2 else "P4_1"
                            lat[blk["layer"]] = self.conv(C["backbone.fpn." + nm], x, name="t3" if blk["layer"] == 2 else "t4")
            feats = self._lower_fpn(taps[2], taps[3], taps[4], lat.get(2), lat.get(3))
        self.feat_sizes = [(f.levels[0].H, f.levels[0].W) for f in feats]
        self.feat_sizes_t = torch.tensor(self.feat_sizes, dtype=torch.long, device=self.dev)
        self.num_f_out_t = torch.tensor([len(feats)], dtype=torch.long, device=self.dev)
        self.tape.extend(lstm_tape)
        self._lower_head(feats, we)
        # image-independent head launches (language / grid maps of conv0) go to the side stream BEHIND the query encoder: a
        # side-stream launch waits for the main-stream work enqueued before it, so its place in the program decides what
        # it can overlap — here the whole image encoder
        moved_c, moved_l = [], []
        for (i0, i1) in reversed(self._hoist):          # cut from the back so the earlier spans' indices stay valid ...
            moved_c[0:0] = self.fwd.calls[i0:i1]
            moved_l[0:0] = self.fwd.lanes[i0:i1]
            del self.fwd.calls[i0:i1], self.fwd.lanes[i0:i1]
        self.fwd.calls[hoist_to:hoist_to] = moved_c      # ... and re-insert in their original order
        self.fwd.lanes[hoist_to:hoist_to] = moved_l
        where = lang_at()
        if self.training and where != "head":
            # the head-of-program side-stream block (query encoder, language maps; ~0.25 ms, read by the head ~4 ms later) is released
            # further into the program: behind the stem ("stem") or behind the n-th join of the main stream with the side stream
            # ("j<n>": a join waits for the WHOLE side stream, so in front of the first one the block delays layer1.0's residual add)
            ln = self.fwd.lanes
            j = next((i for i in range(1, len(ln)) if ln[i] == 0), None)
            if j is not None and j > 1 and all(l == 1 for l in ln[1:j]):
                k = None
                if where == "stem":
                    k = j
                    while k < len(ln) and ln[k] == 0:
                        k += 1
                elif where.startswith("j") and where[1:].isdigit():
                    joins = [i for i in range(j, len(ln)) if ln[i] == 2]
                    n = int(where[1:])
                    if 1 <= n <= len(joins) - 2:          # (never behind the joins in front of the head itself; a network without
                        k = joins[n - 1] + 1              #  residual joins — SSD-VGG — keeps the block at the head of the program)
                if k is not None:
                    blk_c, blk_l = self.fwd.calls[1:j], ln[1:j]
                    self.fwd.calls[1:k] = self.fwd.calls[j:k] + blk_c
                    self.fwd.lanes[1:k] = ln[j:k] + blk_l

        # ---- backward program: replay the tape in reverse ----------------------------------------------------------
        if self.training:
            for emit in reversed(self.tape):
                emit()
            if WG_BATCH:
                self._batch_wgrads()
        self.tape = []

    def _batch_wgrads(self):
        """Round 6: the Winograd weight gradients of a stage's IDENTICAL convolutions (layerN.1 .. layerN.k conv2: fpn_resnet.py:86-100)
        become ONE launch (zsg_conv_wgrad_wino_batched) at the position of the last of them in the backward program.  Weight gradients
        are leaves of the backward graph and every activation / gradient buffer of a plan lives until the next forward, so holding the
        earlier ones back changes nothing but the order in which the side stream does its work: a job batch has njobs x the (n, c)
        blocks, i.e. a fraction of the split-K slabs and a longer stage loop per block (l3_conv2 x 5: 268 -> 205 us,
        profiles/r06_wgrad_batching.txt).  The launch indices recorded for DDP's buckets / the Adam split (grad_ready) are re-based."""
        import ctypes as C_
        groups = {}
        for rec in self._wg_log:
            idx, d, src, dy, gw, pname, what = rec
            # (Winograd launches only: batching the direct 1x1 weight gradients as well was built and measured — 13.07 -> 13.08 ms, the
            # later release of 24 launches costs what their shorter kernels save — and removed again, profiles/r06_wgrad_batching.txt)
            if d.nseg != 1 or not d.use_wino or self.bwd.calls[idx][0] is not lib.zsg_conv_wgrad_wino:
                continue
            # one geometry = one descriptor, byte for byte, the tile hint aside (the jobs of a launch share everything but their pointers)
            dz = type(d).from_buffer_copy(d)
            dz.tile_hint = 0
            sig = (bool(d.use_wino), bytes(dz))
            groups.setdefault(sig, []).append(rec)
        drop, put = set(), {}
        for sig, recs in groups.items():
            recs.sort(key=lambda r: r[0])
            for k in range(0, len(recs), 8):              # (at most ZSG_WG_MAX_JOBS per launch)
                part = recs[k:k + 8]
                if len(part) < 2:
                    continue
                n = len(part)
                d0 = part[0][1]
                db = type(d0).from_buffer_copy(d0)
                srcs, dys, gws = [r[2].buf for r in part], [r[3].buf for r in part], [r[4] for r in part]
                wino = bool(d0.use_wino)
                autotune_wgrad_batch(db, n, wino, srcs, dys, self.tune_dw, self.wg_ws, self.wg_ws_bytes, stream_ptr())
                VP = C_.c_void_p * n
                a_src, a_dy, a_dw = VP(*[t.data_ptr() for t in srcs]), VP(*[t.data_ptr() for t in dys]), VP(*[t.data_ptr() for t in gws])
                self.bwd.keep += [db, a_src, a_dy, a_dw] + srcs + dys + gws
                conv = (C_.byref(db), C_.c_int32(n), C_.cast(a_src, C_.c_void_p), C_.cast(a_dy, C_.c_void_p), C_.cast(a_dw, C_.c_void_p),
                        C_.c_int32(1), C_.c_void_p(self.wg_ws.data_ptr()), C_.c_size_t(self.wg_ws_bytes))
                last = part[-1][0]
                put[last] = (lib.zsg_conv_wgrad_wino_batched, conv, "wgrad x%d:" % n + part[-1][6].split(":", 1)[1] + " .. " + part[0][6].split(":", 1)[1])
                drop.update(r[0] for r in part[:-1])
                for r in part:
                    self.grad_ready[r[5]] = last
                    if r[1] in self._tunables:
                        self._tunables.remove(r[1])
        if not put:
            return
        remap, calls, lanes = {}, [], []
        for i, (c, l) in enumerate(zip(self.bwd.calls, self.bwd.lanes)):
            if i in drop:
                continue
            remap[i] = len(calls)
            calls.append(put.get(i, c))
            lanes.append(l)
        # a dropped launch's index maps to the next kept one (its gradient is complete no earlier than the batch, re-based above)
        nxt = len(calls)
        for i in range(len(self.bwd.calls), -1, -1):
            if i in remap:
                nxt = remap[i]
            else:
                remap[i] = nxt
        self.bwd.calls, self.bwd.lanes = calls, lanes
        self.grad_ready = {k: remap[v] for k, v in self.grad_ready.items()}
        self.n_wgrad_batches = len(put)
        # release policy with batches (ops.SIDE_BATCH): a bottleneck stage now releases ONE large Winograd launch per stage plus its 1x1
        # weight gradients; releasing every second main-stream convolution instead of every third gets the batch going earlier —
        # configs[1] 13.08 -> 13.03 / 13.14 -> 13.05 ms on two boxes (1 and 4 are slower; ResNet-18's basic blocks and SSD-VGG keep 3:
        # profiles/r06_ab_release_final.txt)
        if self.bwd.side_batch == 3 and getattr(self.net, "block_kind", "") == "bottleneck":
            self.bwd.side_batch = 2

    def _lower_stem_fused(self, L: ConvL, Lb: BnL, x0: Act, H1: int, W1: int, H2: int, W2: int) -> Act:
        """conv1 -> bn1 -> relu -> maxpool (mdl.py:149-152) in training: the BatchNorm + ReLU + max-pool are ONE pass over the stem
        activation (zsg_bn_relu_maxpool_fwd / _bwd) — the normalised 150x150x64 map, its ReLU mask and the dense pool gradient
        never exist (the stem activation is 92 MB at B=16: 311 -> 121 MB forward, 552 -> 236 MB backward)."""
        net, B = self.net, self.B
        y = self.conv(L, x0, name="stem.y", bn_fuse=Lb)
        rows = B * H1 * W1
        rm, rv = net._rm[Lb.index:Lb.index + Lb.c], net._rv[Lb.index:Lb.index + Lb.c]
        gam, bet = self.P(Lb.name + ".weight"), self.P(Lb.name + ".bias")
        self.ws_need = max(getattr(self, "ws_need", 0), lib.zsg_bn_workspace_bytes(rows, Lb.c))
        if getattr(y, "bn_chunks", 0) > 0:
            mean, invstd = y.bn_mean, y.bn_invstd
            if y.bn_inline is not None:          # few partial rows (small inputs): nobody else will finalize them
                self.fwd.add(lib.zsg_bn_stats_from_partials, y.bn_inline, y.bn_chunks, rows, Lb.c, mean, invstd, rm, rv, 0.1, 1e-5, what="stats:" + Lb.name)
        else:
            mean, invstd = self._buf(Lb.c), self._buf(Lb.c)
            self.fwd.add(lib.zsg_bn_stats, y.buf, rows, Lb.c, mean, invstd, rm, rv, 0.1, 1e-5, self.ws, self.ws_bytes, what=Lb.name)
        x = self.act("pool", B, H2, W2, Lb.c)
        idx = self._buf((B * H2 * W2 * Lb.c + 3) // 4)      # uint8 indices, stored in a float-sized buffer
        self.fwd.add(lib.zsg_bn_relu_maxpool_fwd, y.buf, B, H1, W1, Lb.c, mean, invstd, gam, bet, 3, 2, 1, H2, W2, x.buf, idx, what="bn1+relu+maxpool")

        def back():
            if x.grad is None:
                return
            dy = self.grad_of(y)
            self.bwd.add(lib.zsg_bn_relu_maxpool_bwd, self.base(x.grad), idx, y.buf, B, H1, W1, Lb.c, mean, invstd, gam, bet, 3, 2, 1, H2, W2, dy.buf,
                         self.G(Lb.name + ".weight"), self.G(Lb.name + ".bias"), 1, self.ws, self.ws_bytes, what="bnbwd+maxpool_bwd:" + Lb.name)
            dy.gfilled = True
        self.tape.append(back)
        return x

    # ---- generic pooling / normalisation lowering (SSD-VGG trunk) -------------------------------------------------
    def _grad_sink(self, a: Act):
        """Where a non-conv producer may write d(a): straight into a.grad when nothing is there yet and no ReLU mask is
        due, else a scratch buffer that `_grad_commit` folds in with the mask (dx (+)= scratch * (a > 0))."""
        g = self.grad_of(a)
        if not g.gfilled and not a.needs_mask:
            return g, None
        return g, self.like(a, a.name + ".gtmp")

    def _grad_commit(self, a: Act, g: Act, tmp: Optional[Act]):
        if tmp is not None:
            n = sum(a.B * l.H * l.W * a.ld for l in a.levels)
            if a.needs_mask:
                self.bwd.add(lib.zsg_relu_bwd, self.base(tmp), self.base(a), n, self.base(g), int(g.gfilled), what="mask+acc:" + a.name)
            else:
                raise AssertionError("accumulation without mask is not needed by any lowered model")
        g.gfilled = True

    def maxpool(self, x: Act, k: int, s: int, p: int, ceil: bool, name: str) -> Act:
        l = x.levels[0]

        def osz(n):
            o = (n + 2 * p - k + (s - 1 if ceil else 0)) // s + 1
            if ceil and (o - 1) * s >= n + p:
                o -= 1
            return o
        Ho, Wo = osz(l.H), osz(l.W)
        out = self.act(name, x.B, Ho, Wo, x.C)
        idx = self._buf((x.B * Ho * Wo * x.C + 3) // 4)
        self.fwd.add(lib.zsg_maxpool_fwd, x.buf, x.B, l.H, l.W, x.C, k, s, p, Ho, Wo, out.buf, idx, what=name)

        def back():
            if out.grad is None:
                return
            g, tmp = self._grad_sink(x)
            self.bwd.add(lib.zsg_maxpool_bwd, self.base(out.grad), idx, x.B, l.H, l.W, x.C, k, s, p, Ho, Wo, (tmp or g).buf, what=name + "_bwd")
            self._grad_commit(x, g, tmp)
        self.tape.append(back)
        return out

    def l2norm(self, x: Act, name: str, out: Optional[Act] = None, lane: int = 0) -> Act:
        """x / ||x||_2 over channels, no epsilon (ssd_vgg.py:80, mdl.py:118-130); x / out may be levels of packed buffers"""
        l = x.levels[0]
        rows = x.B * l.H * l.W
        if out is None:
            out = self.act(name, x.B, l.H, l.W, x.C)
        nrm = self._buf(rows)
        self.fwd.add(lib.zsg_l2norm_fwd, self.base(x), rows, x.C, self.base(out), nrm, what=name, lane=lane)

        def back():
            if out.grad is None:
                return
            g, tmp = self._grad_sink(x)
            self.bwd.add(lib.zsg_l2norm_bwd, self.base(out.grad), self.base(out), nrm, rows, x.C, self.base(tmp or g), what=name + "_bwd")
            self._grad_commit(x, g, tmp)
        self.tape.append(back)
        return out

    def _lower_ssd(self, x0: Act) -> List[Act]:
        """SSD.forward, ssd_vgg.py:54-102 (ReLU fused into every conv epilogue)."""
        net = self.net
        C = net.convs
        e = "backbone.encoder."
        # pyramid sizes of the 300-style trunk: conv4_3, conv7, then the four extras taps
        l0 = x0.levels[0]

        def trunk(n):
            n = (n // 2) // 2                       # two floor-mode 2x2 pools
            c43 = -(-n // 2)                        # ceil-mode pool ('C')
            c7 = c43 // 2
            e1 = conv_out(c7, 3, 2, 1)
            e3 = conv_out(e1, 3, 2, 1)
            e5 = conv_out(e3, 3, 1, 0)
            e7 = conv_out(e5, 3, 1, 0)
            return [c43, c7, e1, e3, e5, e7]
        all_sizes = list(zip(trunk(l0.H), trunk(l0.W)))
        fl = self._pyramid(all_sizes[1:] if net.six_hundred else all_sizes)
        dest = ([None] + fl) if net.six_hundred else fl          # destination of out_sources[i]
        x = x0
        sources = []
        for layer in net.vgg_layers:
            if layer[0] == "conv":
                x = self.conv(C[f"{e}vgg.{layer[1]}"], x, relu=True, name=f"vgg.{layer[1]}")
                if layer[1] == 21:                      # conv4_3 + ReLU == vgg[0:23]
                    sources.append(self.l2norm(x, "conv4_3.norm"))
            else:
                _, idx, k, s, p, ceil = layer
                x = self.maxpool(x, k, s, p, ceil, f"vgg.{idx}")
        sources.append(x)
        for k in range(net.n_extras):
            tap = (k % 2 == 1)
            x = self.conv(C[f"{e}extras.{k}"], x, relu=True, name=f"extras.{k}", out=dest[2 + (k - 1) // 2] if (tap and k > 1) else None)
            if tap:
                sources.append(x)
        outs = [self.conv(C[f"{e}fproj{i + 1}"], sources[i], name=f"fproj{i + 1}", out=dest[i]) for i in range(3)] + sources[3:]
        return outs[1:] if net.six_hundred else outs

    def _lower_block(self, blk, x: Act, defer_out: bool = False) -> Act:
        """fpn_resnet.py:26-58 (BasicBlock), :61-100 (Bottleneck).  Training: a projection shortcut (downsample conv +
        BatchNorm) only depends on the block input, so it is lowered FIRST, on the side stream, and runs concurrently with
        the block's main branch; the last BatchNorm (which adds it) joins."""
        net = self.net
        C, BN, q = net.convs, net.bns, blk["prefix"]
        rd = x
        if blk["ds"]:
            with self.on_side_stream():
                rd = self.conv_bn(C[q + "downsample.0"], BN[q + "downsample.1"], x, False, name=q + "rd", yname=q + "yd")
        join = blk["ds"]
        if net.block_kind == "bottleneck":
            a1 = self.conv_bn(C[q + "conv1"], BN[q + "bn1"], x, True, name=q + "a1", yname=q + "y1")
            a2 = self.conv_bn(C[q + "conv2"], BN[q + "bn2"], a1, True, name=q + "a2", yname=q + "y2")
            if self.training:
                y3 = self.conv(C[q + "conv3"], a2, name=q + "y3", bn_fuse=BN[q + "bn3"])
                return self.bn(BN[q + "bn3"], y3, True, residual=rd, name=q + "out", join=join, defer=defer_out)
            return self.conv_bn(C[q + "conv3"], BN[q + "bn3"], a2, True, residual=rd, name=q + "out", yname=q + "y3")
        a1 = self.conv_bn(C[q + "conv1"], BN[q + "bn1"], x, True, name=q + "a1", yname=q + "y1")
        if self.training:
            y2 = self.conv(C[q + "conv2"], a1, name=q + "y2", bn_fuse=BN[q + "bn2"])
            return self.bn(BN[q + "bn2"], y2, True, residual=rd, name=q + "out", join=join)
        return self.conv_bn(C[q + "conv2"], BN[q + "bn2"], a1, True, residual=rd, name=q + "out", yname=q + "y2")

    def _pyramid(self, sizes) -> List[Act]:
        """The head's input features: all pyramid levels packed level-major in ONE buffer (so every head convolution is
        one grouped launch); the producers write their level in place."""
        self.Fpack = self.packed("head.feat_raw" if self.net.do_norm else "head.feat", self.B, sizes, 256)
        lv = [self.Fpack.lvl(i) for i in range(len(sizes))]
        for i, a in enumerate(lv):
            a.name = f"feat{i}"
        return lv

    def _lower_fpn(self, c3: Act, c4: Act, c5: Act, t3: Optional[Act] = None, t4: Optional[Act] = None) -> List[Act]:
        """fpn_resnet.py:154-178"""
        net, B = self.net, self.B
        C = net.convs
        f = "backbone.fpn."
        hw = lambda a: (a.levels[0].H, a.levels[0].W)
        s6 = tuple(conv_out(v, 3, 2, 1) for v in hw(c5))
        s7 = tuple(conv_out(v, 3, 2, 1) for v in s6)
        if net.six_hundred:
            fl = self._pyramid([hw(c4), hw(c5), s6, s7])
            o3, (o4, o5, o6, o7) = None, fl
        else:
            fl = self._pyramid([hw(c3), hw(c4), hw(c5), s6, s7, (1, 1)])
            o3, o4, o5, o6, o7, o8 = fl
        # The pyramid's output convolutions P5_2 / P4_2 and the whole P6 -> P7 (-> P8) chain are leaves that only the head reads:
        # in training they run on the side stream, concurrently with the lateral / top-down path and the large P3_2.
        if t3 is not None or t4 is not None:
            # the laterals lowered early on the side stream (P3_1 under layer3, P4_1 under layer4) are joined HERE, in front of the pyramid's
            # own side-stream launches: a join waits for the whole side stream
            self._join_side()
        # ZSG_FPN_ORDER (round 5; rocprofv3 showed the head's first convolution waiting 56 us for the side stream's chain P5_2 -> P4_2 -> P6 ->
        # ReLU -> P7 -> pool, which only started behind P5_1): "p6" = the P6 chain (it reads C5 only) is released BEFORE P5_1; "p6m" = also
        # P4_2 on the main stream; "0" = round 3's order.
        order = os.environ.get("ZSG_FPN_ORDER", FPN_ORDER_DEFAULT)
        p6_early = order in ("p6", "p6m") and not net.six_hundred and self.training
        if p6_early:
            # (P5_1 stays the consumer whose data gradient completes layer4's last BatchNorm dout — it carries that BatchNorm's backward
            # sums, which the strided P6 cannot — so the chain's backward entries go BEHIND P5_1's on the tape: the tape is replayed in reverse)
            keep, n0 = getattr(c5, "_consumed", False), len(self.tape)
            c5._consumed = True
            p6, p7, p8 = self._lower_p6_chain(c5, o6, o7, o8)
            p6_tape = self.tape[n0:]
            del self.tape[n0:]
            c5._consumed = keep
        p51 = self.conv(C[f + "P5_1"], c5, name="p51")
        if p6_early:
            self.tape.extend(p6_tape)
        with self.on_side_stream():
            p5 = self.conv(C[f + "P5_2"], p51, out=o5)
        if t4 is None:
            t4 = self.conv(C[f + "P4_1"], c4, name="t4")
        p41 = self._upsample_add(t4, p51, "p41")
        if order == "p6m" and p6_early:
            p4 = self.conv(C[f + "P4_2"], p41, out=o4)
        else:
            with self.on_side_stream():
                p4 = self.conv(C[f + "P4_2"], p41, out=o4)
        if p6_early:
            t3 = t3 if t3 is not None else self.conv(C[f + "P3_1"], c3, name="t3")
            p31 = self._upsample_add(t3, p41, "p31")
            p3 = self.conv(C[f + "P3_2"], p31, out=o3, name="p3")
            self._join_side()
            return [p3, p4, p5, p6, p7, p8]
        # (P3_1 / top-down add / the large P3_2 are lowered BEHIND the P6 -> P7 -> P8 chain: a side-stream launch waits for the main-stream
        # work enqueued before it, so in program order behind P3_2 the chain only started when P3_2 had finished and the head's first
        # convolution waited ~90 us for it; here it runs under P3_1 / P3_2)
        t3_in = t3

        def lower_p3():
            t3 = t3_in if t3_in is not None else self.conv(C[f + "P3_1"], c3, name="t3")
            p31 = self._upsample_add(t3, p41, "p31")
            return self.conv(C[f + "P3_2"], p31, out=o3, name="p3")
        p6_first = os.environ.get("ZSG_FPN_P6_FIRST", "1") != "0"
        if not p6_first:
            p3 = lower_p3()
        side = self.on_side_stream()
        side.__enter__()
        p6 = self.conv(C[f + "P6"], c5, out=o6)
        r6 = self.act("r6", B, p6.levels[0].H, p6.levels[0].W, 256)
        n6 = r6.buf.numel()
        self.fwd.add(lib.zsg_relu_fwd, self.base(p6), n6, r6.buf, what="relu(p6)", lane=self._lane)

        def relu_back():
            if r6.grad is None:
                return
            g = self.grad_of(p6)
            self.bwd.add(lib.zsg_relu_bwd, self.base(r6.grad), self.base(p6), n6, self.base(g), int(g.gfilled), what="relu_bwd(p6)")
            g.gfilled = True
        self.tape.append(relu_back)
        p7 = self.conv(C[f + "P7_2"], r6, out=o7)
        if net.six_hundred:
            side.__exit__(None, None, None)
            if p6_first:
                p3 = lower_p3()
            self._join_side()
            return [p4, p5, p6, p7]           # p3 is computed and dropped, as the reference does (fpn_resnet.py:173-174)
        l7 = p7.levels[0]
        p8 = o8
        self.fwd.add(lib.zsg_avgpool_fwd, self.base(p7), B, l7.H * l7.W, 256, self.base(p8), what="avgpool", lane=self._lane)
        side.__exit__(None, None, None)

        def avg_back():
            if p8.grad is None:
                return
            g = self.grad_of(p7)
            self.bwd.add(lib.zsg_avgpool_bwd, self.base(p8.grad), B, l7.H * l7.W, 256, self.base(g), int(g.gfilled), what="avgpool_bwd")
            g.gfilled = True
        self.tape.append(avg_back)
        if p6_first:
            p3 = lower_p3()
        self._join_side()
        return [p3, p4, p5, p6, p7, p8]

    def _lower_p6_chain(self, c5: Act, o6: Act, o7: Act, o8: Act):
        """P6 -> ReLU -> P7_2 -> global average pool (fpn_resnet.py:175-178 + mdl.py's P8) on the side stream"""
        net, B = self.net, self.B
        C = net.convs
        f = "backbone.fpn."
        with self.on_side_stream():
            p6 = self.conv(C[f + "P6"], c5, out=o6)
            r6 = self.act("r6", B, p6.levels[0].H, p6.levels[0].W, 256)
            n6 = r6.buf.numel()
            self.fwd.add(lib.zsg_relu_fwd, self.base(p6), n6, r6.buf, what="relu(p6)", lane=self._lane)

            def relu_back():
                if r6.grad is None:
                    return
                g = self.grad_of(p6)
                self.bwd.add(lib.zsg_relu_bwd, self.base(r6.grad), self.base(p6), n6, self.base(g), int(g.gfilled), what="relu_bwd(p6)")
                g.gfilled = True
            self.tape.append(relu_back)
            p7 = self.conv(C[f + "P7_2"], r6, out=o7)
            l7 = p7.levels[0]
            p8 = o8
            self.fwd.add(lib.zsg_avgpool_fwd, self.base(p7), B, l7.H * l7.W, 256, self.base(p8), what="avgpool", lane=self._lane)

            def avg_back():
                if p8.grad is None:
                    return
                g = self.grad_of(p7)
                self.bwd.add(lib.zsg_avgpool_bwd, self.base(p8.grad), B, l7.H * l7.W, 256, self.base(g), int(g.gfilled), what="avgpool_bwd")
                g.gfilled = True
            self.tape.append(avg_back)
        return p6, p7, p8

    def _upsample_add(self, a: Act, p: Act, name: str, join: bool = False) -> Act:
        la, lp = a.levels[0], p.levels[0]
        out = self.act(name, a.B, la.H, la.W, a.C)
        self.fwd.add(lib.zsg_upsample_add_fwd, a.buf, p.buf, a.B, lp.H, lp.W, la.H, la.W, a.C, out.buf, what=name,
                     lane=2 if (join and self.training) else 0)

        def back():
            if out.grad is None:
                return
            assert a.grad is None
            a.grad = out.grad                     # identity branch: share the buffer
            g = self.grad_of(p)
            self.bwd.add(lib.zsg_upsample_add_bwd, self.base(out.grad), a.B, lp.H, lp.W, la.H, la.W, a.C, self.base(g), int(g.gfilled),
                         what=name + "_bwd")
            g.gfilled = True
        self.tape.append(back)
        return out

    def _lower_lstm(self) -> Act:
        """mdl.py:296-336.  we [B, 2H] = [h_fwd(len-1) | reverse-cell(x[len-1])]"""
        net, B, T = self.net, self.B, self.T
        E, Hd = net.emb_dim, net.lstm_dim
        H4 = 4 * Hd
        we = self.act("we", B, 1, 1, net.lstm_out_dim)
        dirs = [("", 0)] + ([("_reverse", 1)] if net.bid else [])
        x_all = Act(self.in_qvec, B, E, E, [Level(0, 1, T, T * E)], "qvec")
        x_all.requires_grad = False
        xlast = self.act("xlast", B, 1, 1, E, requires_grad=False)
        for suf, di in dirs:
            Tn = T if di == 0 else 1
            xin = x_all if di == 0 else xlast
            if di == 1:
                self.fwd.add(lib.zsg_lstm_gather_last, self.in_qvec, self.in_qlens, B, T, E, xlast.buf, what="gather_last", lane=1)
            gin = self.act("gin" + suf, B, 1, Tn, H4)
            d = fwd_desc(xin, gin, E, H4, 1, 1, 0, 1, wC=E)
            self.fwd.add(lib.zsg_conv_igemm, d, xin.buf, self.P("lstm.weight_ih_l0" + suf), gin.buf, self.P("lstm.bias_ih_l0" + suf),
                         None, None, None, what="lstm_in" + suf, lane=1)
            gates, cst, hprev = self._buf(B * Tn * H4), self._buf(B * Tn * Hd), self._buf(B * Tn * Hd)
            h0 = self.in_h0[di * B * Hd:(di + 1) * B * Hd]
            c0 = self.in_c0[di * B * Hd:(di + 1) * B * Hd]
            lens = self.in_qlens if di == 0 else None
            self.fwd.add(lib.zsg_lstm_fwd, gin.buf, self.P("lstm.weight_hh_l0" + suf), self.P("lstm.bias_hh_l0" + suf), h0, c0,
                         self.in_qlens, lens, B, Tn, Hd, gates, cst, hprev, we.buf, net.lstm_out_dim, di * Hd, what="lstm" + suf, lane=1)

            def back(suf=suf, di=di, Tn=Tn, xin=xin, gates=gates, cst=cst, hprev=hprev, c0=c0, lens=lens):
                if we.grad is None:
                    return
                dg = Act(self._buf(B * Tn * H4), B, H4, H4, [Level(0, 1, Tn, Tn * H4)], "dgates" + suf)
                self.bwd.add(lib.zsg_lstm_bwd, we.grad.buf, net.lstm_out_dim, di * Hd, self.P("lstm.weight_hh_l0" + suf), gates, cst, c0,
                             self.in_qlens, lens, B, Tn, Hd, dg.buf, what="lstm_bwd" + suf, lane=1)
                d_ih = fwd_desc(xin, dg, E, H4, 1, 1, 0, 1, wC=E)
                self.bwd.add(lib.zsg_conv_wgrad, d_ih, xin.buf, dg.buf, self.G("lstm.weight_ih_l0" + suf), 1, self.wg_ws, self.wg_ws_bytes,
                             what="wgrad:w_ih" + suf, lane=1)
                hp = Act(hprev, B, Hd, Hd, [Level(0, 1, Tn, Tn * Hd)], "hprev" + suf)
                d_hh = fwd_desc(hp, dg, Hd, H4, 1, 1, 0, 1, wC=Hd)
                self.bwd.add(lib.zsg_conv_wgrad, d_hh, hp.buf, dg.buf, self.G("lstm.weight_hh_l0" + suf), 1, self.wg_ws, self.wg_ws_bytes,
                             what="wgrad:w_hh" + suf, lane=1)
                for bname in ("lstm.bias_ih_l0", "lstm.bias_hh_l0"):
                    self.bwd.add(lib.zsg_colsum, dg.buf, 1, 0, B * Tn, H4, 0, H4, self.G(bname + suf), 1, what="bgrad:" + bname + suf, lane=1)
            self.tape.append(back)
        return we

    def _lower_head(self, feats: List[Act], we: Optional[Act]):
        """concat_we (mdl.py:69-104, blind variants :363-375, do_norm :118-130) + the head(s) (mdl.py:211-244, 377-389)
        over all pyramid levels in grouped launches."""
        net, B = self.net, self.B
        sizes = self.feat_sizes
        Cf, Cw, Cg = net.cf, net.cw, (4 if net.use_grid else 0)
        assert [(f.levels[0].H, f.levels[0].W) for f in feats] == sizes
        hc = types.SimpleNamespace(Cf=Cf, Cw=Cw, Cg=Cg, we=we, gridmap=None)
        # do_norm: per-pixel channel L2 normalisation of the maps and of the language vector
        hc.Fp = self.Fpack if Cf else None
        heads_in = feats                         # the Acts whose .grad conv0's data gradient fills
        if net.do_norm and Cf:
            hc.Fp = self.packed("head.feat", B, sizes, 256)
            heads_in = [self.l2norm(f, f"featnorm{i}", out=hc.Fp.lvl(i)) for i, f in enumerate(feats)]
        if net.do_norm and Cw:
            # (side stream, behind the query encoder that produces `we`; hoisted together with the language maps that read it)
            hoist = self.training and Cf
            i0 = len(self.fwd.calls)
            hc.we = self.l2norm(we, "we.norm", lane=(1 if hoist else 2))
            if hoist:
                self._hoist.append((i0, len(self.fwd.calls)))
        if Cg:
            gm = np.zeros((sum(h * w for h, w in sizes), 4), np.float32)
            o = 0
            for (h, w) in sizes:
                gm[o:o + h * w, :2] = anchors_mod.create_grid_np(h, w).reshape(h * w, 2)
                o += h * w
            hc.gridmap = self.packed("head.grid", 1, sizes, 4)
            hc.gridmap.buf.copy_(torch.from_numpy(gm.reshape(-1)))
            hc.gridmap.requires_grad = False
        P = sum(hh * ww for hh, ww in sizes)
        self.A = P * net.n_anchors
        nA = net.n_anchors

        def out_view(buf, nout):                 # [B][P][nout]: the levels concatenated along the anchor axis
            lv, off = [], 0
            for (hh, ww) in sizes:
                lv.append(Level(off * nout, hh, ww, P * nout))
                off += hh * ww
            return Act(buf, B, nout, nout, lv, "head.out")

        if Cf and self.training:                 # runs after every head's conv0 data gradient (reverse tape order)
            def feats_back():
                dF = hc.Fp.grad
                if dF is None:
                    return
                for i, f in enumerate(heads_in):
                    assert f.grad is None
                    f.grad = dF.lvl(i)
                    f.grad.gfilled = True
                    if f.needs_mask:             # SSD extras feed the head post-ReLU: turn d(relu(y)) into d(y) in place
                        n = f.B * f.levels[0].H * f.levels[0].W * f.ld
                        self.bwd.add(lib.zsg_relu_bwd, self.base(f.grad), self.base(f), n, self.base(f.grad), 0, what=f"mask:feat{i}")
            self.tape.append(feats_back)

        self.out5 = out_view(self._buf(B * P * 5 * nA), 5 * nA)
        self.g5_in = self._buf(B * P * 5 * nA) if self.training else None
        if net.same_atb:
            self._head_stack("att_reg_box", 5 * nA, hc, self.out5, self.g5_in)
            return
        # separate heads (mdl.py:220-225, 383-389): their outputs are interleaved into the [B, A, (4 box | 1 att)] tensor the
        # loss / evaluator kernels read, and the incoming gradient is split the same way
        o_att, o_reg = out_view(self._buf(B * P * nA), nA), out_view(self._buf(B * P * 4 * nA), 4 * nA)
        g_att = self._buf(B * P * nA) if self.training else None
        g_reg = self._buf(B * P * 4 * nA) if self.training else None
        self._head_stack("att_box", nA, hc, o_att, g_att)
        self._head_stack("reg_box", 4 * nA, hc, o_reg, g_reg)
        self.fwd.add(lib.zsg_interleave, o_reg.buf, B * P, nA, 4, self.out5.buf, 5, 0, 0, what="out5<-reg")
        self.fwd.add(lib.zsg_interleave, o_att.buf, B * P, nA, 1, self.out5.buf, 5, 4, 0, what="out5<-att")
        if self.training:
            def split_back():
                self.bwd.add(lib.zsg_interleave, g_reg, B * P, nA, 4, self.g5_in, 5, 0, 1, what="g5->reg")
                self.bwd.add(lib.zsg_interleave, g_att, B * P, nA, 1, self.g5_in, 5, 4, 1, what="g5->att")
            self.tape.append(split_back)

    def _head_stack(self, prefix: str, nout: int, hc, out: Act, g_in: Optional[torch.Tensor]):
        """One 6-convolution head `prefix`.{0..4}.0 / .5 (mdl.py:235-244) on the shared input described by hc; writes
        `out` [B][P][nout]; its backward starts from g_in (same layout)."""
        net, B = self.net, self.B
        C = net.convs
        sizes = self.feat_sizes
        Cf, Cw, Cg, we, Fp, gridmap = hc.Cf, hc.Cw, hc.Cg, hc.we, hc.Fp, hc.gridmap
        L0 = C[prefix + ".0.0"]
        W0n = L0.name + ".weight"
        cp = L0.cpad
        assert cp == Cf + Cw + Cg
        h1 = self.packed(prefix + ".h1", B, sizes, 256)
        # conv0 sees [features | language vector (constant over the image) | grid (constant over the batch)] (or a subset,
        # mdl.py:363-375): only the features go through the big implicit GEMM; the rest enters as an additive map
        #   lmap[b][p][n] = G[p][n] + sum_{tap valid at p} V[b][n*9+tap],  V = W0[:, :, :, lang] . we[b],  G = conv(grid, W0[..., grid])
        lmap = None
        if Cw or Cg:
            # None of this depends on the image: in training it runs on the side stream right behind the query encoder (the
            # launches are moved there after lowering, see _hoist_language_maps) and conv0 joins.
            side = 1 if (self.training and Cf) else 0
            i0 = len(self.fwd.calls)
            V = self.act(prefix + ".V", B, 1, 1, 9 * 256, requires_grad=False)          # stays zero without language
            if Cw:
                dv = fwd_desc(we, V, Cw, 9 * 256, 1, 1, 0, 1, wC=cp, wt_ld=cp, wc0=Cf)
                self.fwd.add(lib.zsg_conv_igemm, dv, we.buf, self.P(W0n), V.buf, None, None, None, None, what=prefix + "0.V", lane=(1 if side else 2))
            G = None
            if Cg:
                G = self.packed(prefix + ".G", 1, sizes, 256)
                dg = fwd_desc(gridmap, G, 4, 256, 3, 1, 1, 1, wC=cp, wc0=Cf + Cw)
                self.fwd.add(lib.zsg_conv_igemm, dg, gridmap.buf, self.P(W0n), G.buf, None, None, None, None, what=prefix + "0.G", lane=side)
            lmap = self.packed(prefix + ".lmap", B, sizes, 256)
            # every level in one launch (G and lmap are packed level-major with the same level list)
            hw = torch.tensor([v for hw_ in sizes for v in hw_], dtype=torch.int32)
            self.fwd.add(lib.zsg_head_lang_map_packed, V.buf, G.buf if G is not None else None, B, len(sizes), hw, 256, lmap.buf,
                         what="lmap", lane=side)
            if side:
                self._hoist.append((i0, len(self.fwd.calls)))
                self._join_side()
        if Cf:
            d0 = fwd_desc(Fp, h1, Cf, 256, 3, 1, 1, 1, wC=cp, wc0=0, relu=True)
            a0 = (Fp.buf, self.P(W0n), h1.buf, self.P(L0.name + ".bias"), lmap.buf if lmap is not None else None, None, None)
            wargs = None
            if wino_mode() != "0":
                U0, job0 = self._wino_u(self.P(W0n).data_ptr(), 256, Cf, 9 * cp, cp, 0)
                wargs = (Fp.buf, U0) + a0[2:]
            self._tune("igemm", lib.zsg_conv_igemm, d0, a0, stream_ptr(), wino_args=wargs)
            if d0.use_wino:
                self.wino_jobs["fwd"].add(*job0)
                self.fwd.add(lib.zsg_conv_wino, d0, *wargs, what=L0.name)
            else:
                self.fwd.add(lib.zsg_conv_igemm, d0, *a0, what=L0.name)
        else:                             # image-blind: h1 = relu(lmap + bias), an affine map with scale 1
            one, zero = self._buf(256) + 1.0, self._buf(256)
            self.fwd.add(lib.zsg_bn_apply, lmap.buf, h1.rows(), 256, zero, one, one, self.P(L0.name + ".bias"), None, 1, h1.buf, None,
                         what=L0.name)
        h1.needs_mask = True

        def head0_back():
            dy = h1.grad
            if dy is None:
                return
            gW0 = self.G(W0n)
            if not Cw:       # (with language the bias gradient falls out of the border sums below)
                self.bwd.add(lib.zsg_colsum, dy.buf, 1, 0, dy.rows(), 256, 0, 256, self.G(L0.name + ".bias"), 1, what="bgrad:" + L0.name, lane=1)
            if Cf:
                dwf = fwd_desc(Fp, dy, Cf, 256, 3, 1, 1, 1, wC=cp, wc0=0)
                self.wgrad(dwf, Fp, dy, W0n, "wgrad:" + L0.name)
                self.dgrad(L0, dy, Fp, n=Cf, row0=0, dx=self.grad_of(Fp))
            # The language / grid columns of dW0, the bias gradient and d(we) hang off dy only and feed nothing but the query encoder's
            # backward (itself on the side stream): with features present they are leaves of the main chain and go to the side stream,
            # so that the pyramid's backward starts right behind conv0's data gradient (ZSG_LANG_BWD_SIDE=0: on the main stream, as before).
            # (not with do_norm: the language vector's normalisation has its backward on the main stream, right behind d(we))
            ln = 1 if (Cf and not (net.do_norm and Cw) and os.environ.get("ZSG_LANG_BWD_SIDE", "1") != "0") else 0
            hws_bytes = 16 << 20           # (a workspace of their own: on the main stream they ran concurrently with the side stream's slabs)
            hws = self._buf(hws_bytes // 4) if (Cw or Cg) else None
            if Cw:
                # language columns of dW0 and d(we) from validity-masked sums of dy, themselves nine plain per-image sums
                S = self._buf(2 * B * 9 * 256)
                Q = self._buf(9 * B * 256)
                S1 = Act(S, B, 9 * 256, 9 * 256, [Level(0, 1, 1, 9 * 256)], "head.S1")
                S2 = Act(S, 1, B, B, [Level(B * 9 * 256, 1, 9 * 256, 9 * 256 * B)], "head.S2")
                self.bwd.add(lib.zsg_memset_f32, Q, Q.numel(), 0.0, what="zero:head.Q", lane=ln)
                for i, (h, w) in enumerate(sizes):
                    self.bwd.add(lib.zsg_head_border_sums, self.base(dy.lvl(i)), B, h, w, 256, Q, what=f"bsum{i}", lane=ln)
                self.bwd.add(lib.zsg_head_border_finalize, Q, B, 256, S, self.base(S2), self.G(L0.name + ".bias"), what="bsum.finalize", lane=ln)
                dwl = fwd_desc(we, S1, Cw, 9 * 256, 1, 1, 0, 1, wC=cp, wt_ld=cp, wc0=Cf)
                self.bwd.add(lib.zsg_conv_wgrad, dwl, we.buf, S, gW0, 1, hws, hws_bytes, what="wgrad:" + L0.name + ".lang", lane=ln)
                ent = net.store.entries[W0n]
                Wrows = Act(net.store.flat, 1, Cw, cp, [Level(ent.offset + Cf, 1, 9 * 256, 9 * 256 * cp)], "head.W0rows")
                gwe = self.grad_of(we)
                dwe = fwd_desc(Wrows, S2, Cw, B, 1, 1, 0, 1, wC=Cw, wt_ld=Cw)
                self.bwd.add(lib.zsg_conv_wgrad, dwe, Wrows.buf, S, self.base(gwe), int(gwe.gfilled), hws, hws_bytes, what="dwe:" + prefix, lane=ln)
                gwe.gfilled = True
            if Cg:
                dys = self.packed(prefix + ".dysum", 1, sizes, 256)
                for i, (h, w) in enumerate(sizes):
                    self.bwd.add(lib.zsg_batch_sum, self.base(dy.lvl(i)), B, h * w * 256, self.base(dys.lvl(i)), what=f"dysum{i}", lane=ln)
                dwg = fwd_desc(gridmap, dys, 4, 256, 3, 1, 1, 1, wC=cp, wc0=Cf + Cw)
                self.bwd.add(lib.zsg_conv_wgrad, dwg, gridmap.buf, dys.buf, gW0, 1, hws, hws_bytes, what="wgrad:" + L0.name + ".grid", lane=ln)
            self.grad_ready[W0n] = len(self.bwd.calls)
        self.tape.append(head0_back)
        hs = [h1]
        for i in range(1, 5):
            nxt = self.packed(f"{prefix}.h{i + 1}", B, sizes, 256)
            self.conv(C[f"{prefix}.{i}.0"], hs[-1], relu=True, out=nxt)      # backward via the tape
            hs.append(nxt)
        L5 = C[prefix + ".5"]
        assert L5.cout == nout
        self.conv(L5, hs[-1], relu=False, out=out)
        self.tape.pop()
        if not self.training:
            return
        # ---- last conv backward: the incoming gradient is re-packed to a multiple of 4 channels (16-byte GEMM rows) -----
        npad = pad4(nout)
        P = sum(hh * ww for hh, ww in sizes)
        lvp, off = [], 0
        for (hh, ww) in sizes:
            lvp.append(Level(off * npad, hh, ww, P * npad))
            off += hh * ww
        g5p = Act(self._buf(B * P * npad), B, npad, npad, lvp, prefix + ".g5p")
        h5 = hs[-1]

        def head5_back():
            self.bwd.add(lib.zsg_pad_rows, g_in, B * P, nout, nout, g5p.buf, npad, what="pad g5")
            # column sums of the PADDED copy (16-byte loads; its pad columns are zero and land in the 4-float storage padding)
            self.bwd.add(lib.zsg_colsum, g5p.buf, 1, 0, B * P, npad, 0, npad, self.G(L5.name + ".bias"), 1, what="bgrad:" + L5.name, lane=1)
            dw = fwd_desc(h5, g5p, L5.cpad, nout, 3, 1, 1, 1, wC=L5.cpad)
            self.wgrad(dw, h5, g5p, L5.name + ".weight", "wgrad:" + L5.name)
            self.dgrad(L5, g5p, h5, n=256)
        self.tape.append(head5_back)

    # ---- execution -------------------------------------------------------------------------------------------------------
    def run_forward(self, img, qvec, qlens, h0, c0) -> torch.Tensor:
        net = self.net
        B = self.B
        ensure_stream_scratch(stream_ptr())      # (stream-K launches take their scratch from the stream they run on)
        net.join_grads()
        rel = None
        if prep_release_top() and self._prep_stream is not None:
            # the side stream's weight preparation may start now (behind the optimizer step), not behind the input copies below
            rel = self._rel_ev
            rel.record(torch.cuda.current_stream())
        img = img.contiguous()
        u8 = img.dtype == torch.uint8
        if not u8 and img.dtype != torch.float32:
            img = img.float()
        T = qvec.shape[1]
        nd = 2 if net.bid else 1
        host_hc = h0.device.type == "cpu" and c0.device.type == "cpu"
        staged = (STAGE_INPUTS and host_hc and qvec.is_cuda and qvec.dtype == torch.float32 and qlens.is_cuda
                  and qlens.dtype in (torch.int64, torch.float32) and qlens.numel() == B)
        if staged:
            # ONE launch for the whole input staging (qvec into the zero-padded token bucket, qlens, the host-drawn h0 | c0 read straight
            # from a pinned ring slot, the BatchNorm counters): five torch operations with 5-20 us between them before (rocprofv3: 45 us
            # at the head of every forward).  A ring slot is rewritten only after the launch that read it has finished.
            if self._hc_pin is None:
                self._hc_pin = torch.empty(8, 2, nd, B, net.lstm_dim).pin_memory()
                self._hc_ev = [None] * 8
            k = self.fwd_id % 8
            if self._hc_ev[k] is not None:
                self._hc_ev[k].synchronize()
            torch.stack([h0.float(), c0.float()], out=self._hc_pin[k])
            qv, ql = qvec.contiguous(), qlens.reshape(B).contiguous()
            n_nbt = net._nbt.numel() if self.training else 0
            check(lib.zsg_stage_inputs(qv.data_ptr(), B, T, net.emb_dim, self.T, self.in_qvec.data_ptr(), ql.data_ptr(),
                                       int(ql.dtype == torch.float32), self.in_qlens.data_ptr(),
                                       self._hc_pin[k].data_ptr(), self.in_hc.numel(), self.in_hc.data_ptr(),
                                       net._nbt.data_ptr() if n_nbt else None, n_nbt, stream_ptr()), "stage_inputs")
            if self._hc_ev[k] is None:
                self._hc_ev[k] = torch.cuda.Event()
            self._hc_ev[k].record(torch.cuda.current_stream())
            self._in_keepalive = (qv, ql)
        else:
            qbuf = self.in_qvec.view(B, self.T, net.emb_dim)
            qbuf[:, :T].copy_(qvec, non_blocking=True)
            if T < self.T:
                qbuf[:, T:].zero_()
            self.in_qlens.copy_(qlens.reshape(B), non_blocking=True)
            if host_hc:     # lstm_init_hidden's two host draws: one transfer
                self.in_hc.view(2, nd, B, net.lstm_dim).copy_(torch.stack([h0.float(), c0.float()]), non_blocking=True)
            else:
                self.in_h0.view(nd, B, net.lstm_dim).copy_(h0, non_blocking=True)
                self.in_c0.view(nd, B, net.lstm_dim).copy_(c0, non_blocking=True)
        # patch the one dynamic pointer (the caller's image tensor)
        fn, args, what = self.fwd.calls[self.img_slot]
        import ctypes as C_
        self.fwd.calls[self.img_slot] = (fn, (C_.c_void_p(img.data_ptr()),) + args[1:], what)
        self._img_keepalive = img
        self.fwd_id += 1
        if self.training and not staged:
            net._nbt.add_(1)
        assert self.img_slot == 0
        if not self.training and self.fold_jobs:       # the weights may have changed since the last eval forward: refold (one launch)
            check(lib.zsg_bn_fold(net.store.flat.data_ptr(), net._rm.data_ptr(), net._rv.data_ptr(), 1e-5, self.fold_jobs_dev.data_ptr(),
                                  len(self.fold_jobs), self.fold_rows, self.fold_arena.data_ptr(), stream_ptr()), "bn_fold")
        if u8:
            check(lib.zsg_u8hwc_to_nhwc4(img.data_ptr(), B * self.H * self.W, self.fwd.calls[self.img_slot][1][5], stream_ptr()), "u8hwc_to_nhwc4")
        else:
            self.fwd.run(stream_ptr(), 0, 1, graph=False)          # the one launch with a per-call pointer (the caller's image)
        do_prep = self.training and self.expect_backward and len(self.prep)
        if do_prep or len(self.prep_u):
            # the forward's Winograd filter transforms and the backward's weight images (transposed filters of the data gradients,
            # their Winograd transforms) depend on the weights only: produced here on the side stream, under the forward
            if self._prep_stream is None:
                self._prep_stream, self._prep_ev, self._u_ev = shared_side_stream(), torch.cuda.Event(), torch.cuda.Event()
            if rel is None:
                self._prep_stream.wait_stream(torch.cuda.current_stream())     # after the optimizer step that wrote the weights
            else:
                self._prep_stream.wait_event(rel)
            if len(self.prep_u):
                self.prep_u.run(self._prep_stream.cuda_stream)
                self._u_ev.record(self._prep_stream)

        def run_prep():
            if do_prep:
                self.prep.run(self._prep_stream.cuda_stream)
                self._prep_ev.record(self._prep_stream)
                self._prep_fwd, self._prep_pending = self.fwd_id, True
        # cut points of the forward program: (launch index, action enqueued in front of that launch)
        cuts = []
        if len(self.prep_u):
            cuts.append((self._wait_idx, lambda: torch.cuda.current_stream().wait_event(self._u_ev)))
        k_prep = self._prep_index() if do_prep else None
        if do_prep and k_prep is None:
            run_prep()             # at the head of the side stream's work
        elif do_prep:
            cuts.append((k_prep, run_prep))
        cuts.sort(key=lambda c: c[0])
        slots, out = self._out_slots(), None
        if slots:
            # the [B, A, 5] output is written straight into a FRESH tensor (the launches that produce it take its address per call), so
            # the caller owns it as with the reference's module — the copy out of the plan's static buffer (a dependent 6 us launch
            # between the head's last convolution and the loss kernels) is gone.  The slots are patched BEFORE any launch range of this
            # forward is replayed: they still hold the previous step's tensor, which may have been freed since.
            out = torch.empty(B, self.A, 5, device=self.out5.buf.device, dtype=torch.float32)
            for a in slots:
                a.value = out.data_ptr()
        pos = 1
        for idx, action in cuts:
            if idx > pos:
                self.fwd.run(stream_ptr(), pos, idx, join=False)
                pos = idx
            action()
        self.fwd.run(stream_ptr(), pos)
        if out is not None:
            return out
        return self.out5.buf.view(B, self.A, 5).clone()

    def _out_slots(self):
        """The pointer arguments of the forward program that hold the address of the [B, A, 5] output buffer (the last head convolution, or
        the two interleave launches of separate heads) — None when the address is also baked into another program of the plan or when
        launch ranges are replayed as hipGraphs (captured addresses): run_forward then copies out of the static buffer."""
        if self._out_slots_v is not False:
            return self._out_slots_v
        import ctypes as C_
        from .ops import HIP_GRAPH
        ptr, nbytes = self.out5.buf.data_ptr(), self.out5.buf.numel() * 4

        def holders(prog):
            # every pointer argument that points INTO the buffer; one at an offset cannot be re-based by value and disables the path
            return [a for _, args, _ in prog.calls for a in args if isinstance(a, C_.c_void_p) and a.value is not None and ptr <= a.value < ptr + nbytes]
        mine = holders(self.fwd)
        others = sum(len(holders(pr)) for pr in (self.bwd, self.prep, self.prep_u) if pr is not None)
        ok = (bool(mine) and all(a.value == ptr for a in mine) and not others and not HIP_GRAPH and os.environ.get("ZSG_FRESH_OUT", "1") != "0")
        self._out_slots_v = mine if ok else None
        return self._out_slots_v

    def _prep_index(self):
        """Where in the forward program the backward's weight images (transposed / Winograd-transformed filters: ~0.2 ms of HBM-bound
        launches the backward needs, nothing in the forward does) are enqueued on the side stream: None = in front of everything
        (ZSG_PREP_AT=top), else the launch index they go in front of — `late` = the first launch that waits for the forward's own
        preparation, `j<n>` = behind the n-th join of the main stream with the side stream (the residual blocks' downsample branches),
        so that they do not sit in front of side-stream work the main stream waits for sooner."""
        if self._prep_idx_v is not False:
            return self._prep_idx_v
        mode = prep_at()
        v = None
        joins = [i for i, l in enumerate(self.fwd.lanes) if l == 2]
        if mode == "late" and len(self.prep_u):
            v = self._wait_idx
        elif mode.startswith("j") and mode[1:].isdigit() and len(joins) - 2 >= int(mode[1:]) >= 1:
            v = joins[int(mode[1:]) - 1] + 1       # (never behind the joins in front of the head: SSD-VGG has no others -> top)
        elif mode.isdigit():
            v = min(int(mode), len(self.fwd.calls))
        self._prep_idx_v = v if (v is None or v > 1) else None
        return self._prep_idx_v

    def run_backward(self, g5: torch.Tensor):
        ensure_stream_scratch(stream_ptr())
        net = self.net
        if not self.training:
            raise RuntimeError("backward through an eval-mode plan")
        # Gradients are ACCUMULATED (+=) into the flat gradient buffer, as autograd does into p.grad: the buffer is zeroed by
        # FusedAdam.zero_grad() (one memset, the p.grad views stay), or here when the p.grad were set to None.
        params = net._ordered_params()
        st = stream_ptr()
        if any(p.grad is None for p in params):
            net._grad_reduced = False
            lib.zsg_memset_f32(net.store.grad.data_ptr(), net.store.grad.numel(), 0.0, st)
            for n, p in zip(net._param_names, params):
                p.grad = net.store.view(n, net.store.grad)
        if g5.data_ptr() != self.g5_in.data_ptr():        # (the loss wrote it in place: see ZSGNet.forward)
            self.g5_in.view_as(g5).copy_(g5)
        ddp = getattr(net, "_ddp", None)
        if self._prep_fwd == self.fwd_id and self._bwd_fwd != self.fwd_id:
            torch.cuda.current_stream().wait_event(self._prep_ev)
            self._prep_pending = False
        else:
            # no side-stream preparation for this forward, or a SECOND backward of it (retain_graph): the split-K / strided
            # data-gradient targets the first one accumulated into must be zeroed again — the preparation is idempotent
            if self._prep_fwd == self.fwd_id and self._prep_pending:
                torch.cuda.current_stream().wait_event(self._prep_ev)
                self._prep_pending = False
            self.prep.run(st)
        self._bwd_fwd = self.fwd_id
        if ddp is not None and ddp.active:
            # The reducer SUM-all-reduces the whole (accumulating) gradient buffer: a second backward before zero_grad would
            # reduce the first one's gradients again.  FusedAdam.zero_grad / dropping the p.grad clears the flag.
            if getattr(net, "_grad_reduced", False):
                raise RuntimeError("DDP: second backward without zero_grad() — gradient accumulation is not supported by the "
                                   "flat-buffer reducer (reduced gradients would be all-reduced again)")
            net._grad_reduced = True
            # average over ranks: pre-scale the incoming gradient (backward is linear), then SUM-all-reduce buckets as
            # soon as the launches that fill them are enqueued; the optimizer waits through wait_gradients().
            if not (g5.data_ptr() == self.g5_in.data_ptr() and self.g5_from_loss == (self.fwd_id, 1.0 / ddp.world)):
                self.g5_in.mul_(1.0 / ddp.world)       # (the loss kernel writes the gradient pre-scaled when it can: loss._LossFn.forward)
            self.g5_from_loss = None
            if self.reducer is None:
                ents = net.store.entries
                spans = [(ents[n].offset, (ents[n].size + 3) // 4 * 4, self.grad_ready.get(n, -1)) for n in net._param_names]
                self.reducer = ddp.make_reducer(spans)
            # (join=False: a range that ends at a bucket boundary leaves the side stream's weight gradients running; the
            # bucket's collective waits for both streams, the main stream only joins at the very end)
            nb = len(self.bwd.calls)
            self.reducer.run(nb, lambda i, j: self.bwd.run(st, i, j, join=(j == nb)), side_stream=lambda: self.bwd._side)
            self.reducer.wait()
        else:
            cut = self._adam_cut() if (getattr(net, "_fused_opt", None) is not None and adam_overlap()) else None
            if cut is None:
                self.bwd.run(st)
            else:
                # FusedAdam is attached: the range up to i_cut completes every gradient behind flat offset `off` (all but the stem /
                # first block, whose weight gradients are the tail of the side stream with nothing left to overlap them); the main
                # stream does NOT join the side stream here — FusedAdam.step updates [off, end) behind an event recorded now, and
                # [0, off) after the join (ZSGNet.join_grads is the join for anyone else)
                i_cut, off = cut
                self.bwd.run(st, 0, i_cut, graph=False, join=False)
                if self.bwd._side is None:
                    self.bwd.run(st, i_cut, len(self.bwd.calls), graph=False)
                else:
                    if self._adam_ev is None:
                        self._adam_ev = torch.cuda.Event()
                    self._adam_ev.record(self.bwd._side)
                    self.bwd.run(st, i_cut, len(self.bwd.calls), graph=False, join=False)
                    net._adam_overlap = (self._adam_ev, self.bwd._side, off)

    def _adam_cut(self):
        """(launch index, flat offset): the backward launches [0, index) complete the gradient of every parameter stored at or behind
        `offset`; the parameters in front of it (the stem and the first block: first in the flat buffer, last in the backward) are what
        the final ~dozen launches write.  None when the split is not worth it."""
        if self._adam_cut_v is not False:
            return self._adam_cut_v
        from . import ops as _ops
        self._adam_cut_v = None
        ents, n = self.net.store.entries, len(self.bwd.calls)
        if not _ops.SIDE_STREAM or _ops.HIP_GRAPH or n < 40:
            return None
        tail = [nm for nm in self.net._param_names if self.grad_ready.get(nm, -1) >= n - 12]
        if not tail:
            return None
        off = max(ents[nm].offset + (ents[nm].size + 3) // 4 * 4 for nm in tail)
        rest = [nm for nm in self.net._param_names if ents[nm].offset >= off]
        if off > self.net.store.total // 4 or not rest:
            return None
        i_cut = max(self.grad_ready.get(nm, -1) for nm in rest) + 1
        if i_cut > n - 4:
            return None
        self._adam_cut_v = (i_cut, off)
        return self._adam_cut_v


def map_pretrained_keys(net: "ZSGNet", sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Key mapping of the two encoder initialisations of the reference's get_default_net (mdl.py:406-422):
      * 'retina'  (mdl.py:411 `tvm.resnet50(True)`): a torchvision ResNet state dict — `conv1.weight`, `bn1.*`,
        `layer1.0.conv1.weight`, ..., `fc.*` — becomes `backbone.encoder.<key>` (the unused `fc.*` is dropped);
      * 'ssd_vgg' (mdl.py:415-416 `encoder.vgg.load_state_dict(torch.load('./weights/vgg16_reducedfc.pth'))`): the
        reduced-fc VGG16 trunk — `0.weight`, `0.bias`, `2.weight`, ..., `33.bias` — becomes `backbone.encoder.vgg.<key>`.
    Full ZSGNet checkpoints (keys already `backbone.` / `att_reg_box.` / `lstm.`-prefixed, optionally under DDP's
    `module.`) pass through unchanged."""
    own = set(net.state_dict().keys())
    keys = [k[7:] if k.startswith("module.") else k for k in sd]
    if any(k in own for k in keys):
        return {k2: v for k2, v in zip(keys, sd.values())}
    prefix = "backbone.encoder.vgg." if net.backbone_kind == "ssd_vgg" else "backbone.encoder."
    return {prefix + k: v for k, v in zip(keys, sd.values()) if not k.startswith("fc.")}


def load_pretrained_encoder(net: "ZSGNet", path: str) -> int:
    """Loads an encoder / full-model checkpoint into `net`; returns the number of tensors taken.  Raises when the file
    matches NOTHING of the model (a silent no-op here would train from random weights while the log says 'pretrained')
    or when a matched tensor has the wrong shape."""
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "model_state_dict" in sd:
        sd = sd["model_state_dict"]
    mapped = map_pretrained_keys(net, sd)
    own = net.state_dict()
    hit = {k: v for k, v in mapped.items() if k in own}
    if not hit:
        raise ValueError(f"pretrained_path={path!r}: none of its {len(sd)} tensors matches a parameter of the "
                         f"{net.backbone_kind} model (first keys: {list(sd)[:4]})")
    for k, v in hit.items():
        if tuple(v.shape) != tuple(own[k].shape):
            raise ValueError(f"pretrained_path={path!r}: {k} has shape {tuple(v.shape)}, the model expects {tuple(own[k].shape)}")
    net.load_state_dict(hit, strict=False)
    return len(hit)


def get_default_net(num_anchors=1, cfg=None):
    """Constructs the network based on the config (reference mdl.py:406-422).  'retina' = ResNet + FPN; the encoder
    depth comes from the optional cfg key `resnet_arch` (default resnet50, the reference's hard-coded choice).
    The reference initialises the encoder from torchvision's ImageNet ResNet-50 (mdl.py:411) / `vgg16_reducedfc.pth`
    (mdl.py:415-416); with no network access the same files are taken from cfg `pretrained_path` (torchvision / SSD key
    layouts are mapped by map_pretrained_keys), else the encoder is randomly initialised."""
    kind = cfg["mdl_to_use"]
    arch = cfg["resnet_arch"] if "resnet_arch" in cfg else "resnet50"
    net = ZSGNet(kind, num_anchors, cfg=cfg, arch=arch)
    path = cfg["pretrained_path"] if "pretrained_path" in cfg else ""
    if path:
        n = load_pretrained_encoder(net, path)
        print(f"loaded {n} pretrained tensors from {path}")
    return net
