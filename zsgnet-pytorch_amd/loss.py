"""ZSGLoss on MI355X (reference `code/loss.py`): anchor matching + focal BCE + smooth-L1, forward and backward fused
into two HIP launches (csrc/loss.hip) — no 17460^2 identity matrix, no host synchronisation, NaN branch on device."""
from functools import partial
from typing import Dict

import torch
from torch import nn

from ._lib import lib, check, stream_ptr
from .anchors import create_anchors


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out5, mod, annot):
        B, A, _ = out5.shape
        dev = out5.device
        losses = torch.empty(3, device=dev)
        grad5 = torch.empty_like(out5)
        mod.match_idx = torch.empty(B, dtype=torch.int32, device=dev)
        mod.npos = torch.empty(B, dtype=torch.int32, device=dev)
        wsb = lib.zsg_loss_workspace_bytes(B, A)
        ws = torch.empty((wsb + 7) // 8, dtype=torch.float64, device=dev)
        flags = (1 if mod.use_focal else 0) | (2 if mod.use_multi else 0) | (4 if mod.use_softmax else 0)
        check(lib.zsg_loss_fwd_bwd(out5.data_ptr(), annot.data_ptr(), mod.anchs.data_ptr(), B, A, mod.alpha, float(mod.gamma),
                                   float(mod.lamb_reg), float(mod.cfg["matching_threshold"]), flags, 1.0, losses.data_ptr(),
                                   grad5.data_ptr(), mod.match_idx.data_ptr(), mod.npos.data_ptr(), ws.data_ptr(), wsb,
                                   stream_ptr()), "zsg_loss_fwd_bwd")
        ctx.save_for_backward(grad5)
        ctx.g5_buf = getattr(out5, "_zsg_g5", None)      # the network plan's incoming-gradient buffer, when out5 came from ZSGNet
        mod._last_losses = losses
        # (a view of the 3-float result, not a copy: one dependent launch less between the loss kernels and the backward)
        return losses.narrow(0, 0, 1).view(())

    @staticmethod
    def backward(ctx, g):
        (grad5,) = ctx.saved_tensors
        buf = ctx.g5_buf
        if buf is not None and buf.numel() == grad5.numel() and buf.device == grad5.device:
            out = buf.view_as(grad5)             # one launch writes g * grad5 where the network's backward reads it
            torch.mul(grad5, g, out=out)
            return out, None, None
        return grad5 * g, None, None


class ZSGLoss(nn.Module):
    """Criterion to be minimised (reference loss.py:11-143).  forward(out, inp) -> {'loss','cls_ls','box_ls'}."""

    def __init__(self, ratios, scales, cfg):
        super().__init__()
        self.cfg = cfg
        self.ratios, self.scales = ratios, scales
        self.alpha, self.gamma = cfg["alpha"], cfg["gamma"]
        self.use_focal, self.use_softmax, self.use_multi = cfg["use_focal"], cfg["use_softmax"], cfg["use_multi"]
        self.lamb_reg = cfg["lamb_reg"]
        self.loss_keys = ["loss", "cls_ls", "box_ls"]
        self.anchs = None
        self.get_anchors = partial(create_anchors, ratios=self.ratios, scales=self.scales, flatten=True)

    def forward(self, out: Dict[str, torch.Tensor], inp: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        annot = inp["annot"].contiguous().float()
        if "att_bbx_out" in out:
            out5 = out["att_bbx_out"]
        else:                                    # a foreign model: rebuild the interleaved [B,A,5] layout
            out5 = torch.cat([out["bbx_out"], out["att_out"]], dim=2)
        out5 = out5.contiguous()
        if self.anchs is None:                   # computed once: sizes are fixed (loss.py:64-72); no .item() sync
            fs = out["feat_sizes"]
            if "num_f_out" in out and out["num_f_out"].numel() > 1:
                fs = fs[:int(out["num_f_out"][0])]
            self.anchs = self.get_anchors(fs, device=out5.device)
        assert self.anchs.shape[0] == out5.shape[1], "anchor count does not match the network output"
        loss = _LossFn.apply(out5, self, annot)
        ls = self._last_losses
        return {"loss": loss, "cls_ls": ls[1], "box_ls": ls[2]}


def get_default_loss(ratios, scales, cfg):
    return ZSGLoss(ratios, scales, cfg)
