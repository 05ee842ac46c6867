"""Evaluator on MI355X (reference `code/evaluator.py`): arg-max score anchor -> box decode -> IoU>=thr accuracy, one
HIP launch per batch (csrc/loss.hip: eval_kernel); only the two boxes per sample that are needed are decoded."""
from functools import partial
from typing import Dict

import torch
from torch import nn

from ._lib import lib, check, stream_ptr
from .anchors import create_anchors


class Evaluator(nn.Module):
    """To get the accuracy.  Operates at training time (reference evaluator.py:20-117)."""

    def __init__(self, ratios, scales, cfg):
        super().__init__()
        self.cfg = cfg
        self.ratios, self.scales = ratios, scales
        self.met_keys = ["Acc", "MaxPos"]
        self.anchs = None
        self.get_anchors = partial(create_anchors, ratios=self.ratios, scales=self.scales, flatten=True)
        self.acc_iou_threshold = cfg["acc_iou_threshold"]

    @torch.no_grad()
    def forward(self, out: Dict[str, torch.Tensor], inp: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        annot = inp["annot"].contiguous().float()
        if "att_bbx_out" in out:
            out5 = out["att_bbx_out"].detach()
        else:
            out5 = torch.cat([out["bbx_out"], out["att_out"]], dim=2).detach()
        out5 = out5.contiguous()
        B, A, _ = out5.shape
        dev = out5.device
        if self.anchs is None:
            fs = out["feat_sizes"]
            if "num_f_out" in out and out["num_f_out"].numel() > 1:
                fs = fs[:int(out["num_f_out"][0])]
            self.anchs = self.get_anchors(fs, device=dev)
        img_size = inp["img_size"].contiguous().float()
        metrics = torch.empty(2, device=dev)
        pred_boxes = torch.empty(B, 4, device=dev)
        pred_scores = torch.empty(B, device=dev)
        self.pred_idx = torch.empty(B, dtype=torch.int32, device=dev)
        self.best_idx = torch.empty(B, dtype=torch.int32, device=dev)
        ws = torch.empty((int(lib.zsg_eval_workspace_bytes(B)) + 3) // 4, device=dev)
        check(lib.zsg_eval(out5.data_ptr(), annot.data_ptr(), self.anchs.data_ptr(), img_size.data_ptr(), B, A,
                           float(self.acc_iou_threshold), metrics.data_ptr(), pred_boxes.data_ptr(), pred_scores.data_ptr(),
                           self.pred_idx.data_ptr(), self.best_idx.data_ptr(), ws.data_ptr(), stream_ptr()), "zsg_eval")
        return {"Acc": metrics[0], "MaxPos": metrics[1], "idxs": inp["idxs"], "pred_boxes": pred_boxes, "pred_scores": pred_scores}


def get_default_eval(ratios, scales, cfg):
    return Evaluator(ratios, scales, cfg)
