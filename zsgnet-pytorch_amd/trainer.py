"""Minimal counterpart of the reference `Learner` (code/utils.py:186-708): the hot loop (train_epoch :393-437), the
validation loop (:353-391), the text log, and the checkpoint format (:440-497) — enough to drive the MI355X hot path
from the same CLI.  The progress bars / tensorboard dir of the reference are not reproduced."""
import json
import pickle
import logging
import os
import time
from pathlib import Path
from typing import Dict, List, Optional

import torch

from . import dist as zdist


class SmoothenValue:
    """Exponentially smoothed value (utils.py:123-139, beta = 0.9)."""

    def __init__(self, beta: float):
        self.beta, self.n, self.mov_avg, self.smooth = beta, 0, 0.0, 0.0

    def add_value(self, val: float) -> None:
        self.n += 1
        self.mov_avg = self.beta * self.mov_avg + (1 - self.beta) * val
        self.smooth = self.mov_avg / (1 - self.beta ** self.n)


class Learner:
    def __init__(self, uid: str, data, mdl, loss_fn, cfg, eval_fn, opt_fn, device=torch.device("cuda")):
        self.uid, self.data, self.mdl, self.loss_fn, self.cfg, self.eval_fn, self.opt_fn, self.device = \
            uid, data, mdl, loss_fn, cfg, eval_fn, opt_fn, device
        self.rank = zdist.get_rank()
        self.num_it, self.num_epoch, self.best_met = 0, 0, 0.0
        self.loss_keys, self.met_keys = loss_fn.loss_keys, eval_fn.met_keys
        self.log_dir = Path(cfg["tmp_path"]) / "txt_logs"
        self.model_dir = Path(cfg["tmp_path"]) / "models"
        self.model_file = self.model_dir / f"{uid}.pth"
        self.predictions_dir = Path(cfg["tmp_path"]) / "predictions" / uid          # utils.py:248-250
        if self.rank == 0:
            self.log_dir.mkdir(parents=True, exist_ok=True)
            self.model_dir.mkdir(parents=True, exist_ok=True)
        self.predictions_dir.mkdir(parents=True, exist_ok=True)
        self.logger = logging.getLogger("zsg." + uid)
        self.optimizer, self.lr_scheduler = None, None
        if cfg["resume"] and (cfg["resume_path"] or self.model_file.exists()):
            self.load_model_dict(cfg["resume_path"] or str(self.model_file), cfg["load_opt"])

    # ---- optimizer / scheduler (utils.py:667-691) ----------------------------------------------------------------
    def prepare_optimizer(self, lr: float):
        self.optimizer = self.opt_fn(self.mdl, lr=lr)
        if self.cfg["use_reduce_lr_plateau"]:     # reference steps it with val accuracy in the default mode='min'
            self.lr_scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, factor=self.cfg["reduce_factor"],
                                                                           patience=self.cfg["patience"])

    # ---- checkpoints (utils.py:440-497) -----------------------------------------------------------------------------
    def save_model_dict(self):
        if self.rank != 0:
            return
        ckpt = {"model_state_dict": {k: v.detach().cpu().contiguous() for k, v in self.mdl.state_dict().items()},
                "optimizer_state_dict": self.optimizer.state_dict() if self.optimizer else None,
                "scheduler_state_dict": self.lr_scheduler.state_dict() if self.lr_scheduler else None,
                "num_it": self.num_it, "num_epoch": self.num_epoch, "cfgtxt": json.dumps({k: v for k, v in self.cfg.items() if k != "_frozen"}, default=str),
                "best_met": self.best_met}
        torch.save(ckpt, self.model_file)

    def load_model_dict(self, resume_path: str, load_opt: bool = False):
        """utils.py:440-497.  A missing file is not an error: the reference logs it and starts from scratch (:443-457).
        load_opt: optimizer moments / step, param_groups and the LR scheduler are restored too — the optimizer is created
        here if needed (fit() keeps an existing one), so a resumed run continues bias correction and the reduced LR."""
        if not resume_path or not Path(resume_path).exists():
            self.logger.info("No checkpoint at %r: starting from scratch", resume_path)
            if self.rank == 0:
                print(f"resume: no checkpoint at {resume_path!r}, starting from scratch", flush=True)
            return False
        ckpt = torch.load(resume_path, map_location="cpu")
        net = self.mdl.module if hasattr(self.mdl, "module") else self.mdl
        net.load_state_dict(ckpt["model_state_dict"], strict=self.cfg["strict_load"])
        self.num_it, self.num_epoch, self.best_met = ckpt.get("num_it", 0), ckpt.get("num_epoch", 0), ckpt.get("best_met", 0.0)
        if load_opt and ckpt.get("optimizer_state_dict"):
            if self.optimizer is None:
                self.prepare_optimizer(self.cfg["lr"])
            self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            if self.lr_scheduler is not None and ckpt.get("scheduler_state_dict"):
                self.lr_scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        return True

    # ---- loops ----------------------------------------------------------------------------------------------------------
    def _to_device(self, batch):
        return {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}

    def train_epoch(self) -> Dict[str, float]:
        """the hot loop, utils.py:393-437"""
        self.mdl.train()
        sm_loss = {k: SmoothenValue(0.9) for k in self.loss_keys}
        sm_met = {k: SmoothenValue(0.9) for k in self.met_keys}
        n_img, t0 = 0, time.perf_counter()
        for batch in self.data.train_dl:
            self.num_it += 1
            batch = self._to_device(batch)
            self.optimizer.zero_grad()
            out = self.mdl(batch)
            out_loss = self.loss_fn(out, batch)
            out_loss["loss"].mean().backward()
            self.optimizer.step()
            metric = self.eval_fn(out, batch)
            n_img += batch["img"].shape[0]
            if self.num_it % 2 == 0:             # the reference logs every 2 iterations (utils.py:428); this is its D2H sync
                for k in self.loss_keys:
                    sm_loss[k].add_value(float(out_loss[k].detach()))
                for k in self.met_keys:
                    sm_met[k].add_value(float(metric[k]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res = {k: v.smooth for k, v in sm_loss.items()}
        res.update({k: v.smooth for k, v in sm_met.items()})
        res["images_per_s"] = n_img * zdist.get_world_size() / dt
        return res

    @torch.no_grad()
    def validate(self, dl=None, with_predictions: bool = False):
        """utils.py:353-391 (eval mode; losses / metrics averaged over batches weighted by batch size).  The sums are
        ALL-reduced: every rank steps ReduceLROnPlateau and gates best_met / checkpoints on the same global numbers (the
        reference reduces to rank 0 only, so its replicas' learning rates can drift apart).
        with_predictions: also return the per-sample records the reference pickles — a list of
        {'id': idxs, 'pred_boxes': [x1,y1,x2,y2] pixels, 'pred_scores': float} (utils.py:377-383, README 'Evaluation')."""
        self.mdl.eval()
        dl = dl or self.data.valid_dl
        sums = {k: torch.zeros((), device=self.device) for k in self.loss_keys + self.met_keys}
        n = 0
        recs = []                                        # device tensors; one host copy after the loop
        for batch in dl:
            batch = self._to_device(batch)
            out = self.mdl(batch)
            ls = self.loss_fn(out, batch)
            met = self.eval_fn(out, batch)
            b = batch["img"].shape[0]
            for k in self.loss_keys:
                sums[k] += ls[k].detach() * b
            for k in self.met_keys:
                sums[k] += met[k] * b
            n += b
            if with_predictions:
                recs.append((met["idxs"], met["pred_boxes"], met["pred_scores"]))
        sums["__n"] = torch.tensor(float(n), device=self.device)
        red = zdist.reduce_dict(sums)
        tot = float(red["__n"])
        res = {k: float(red[k]) / tot for k in self.loss_keys + self.met_keys}
        if not with_predictions:
            return res
        preds = []
        if recs:
            ids = torch.cat([r[0].reshape(-1) for r in recs]).cpu().tolist()
            boxes = torch.cat([r[1] for r in recs]).cpu().tolist()
            scores = torch.cat([r[2].reshape(-1) for r in recs]).cpu().tolist()
            preds = [{"id": i, "pred_boxes": bx, "pred_scores": sc} for i, bx, sc in zip(ids, boxes, scores)]
        return res, preds

    def update_prediction_file(self, predictions, pred_file: Path):
        """utils.py:500-509: one pickle per rank ('<rank>_<name>') when distributed — eval_script.evaluate merges them —
        else a single file."""
        pred_file = Path(pred_file)
        if zdist.get_world_size() > 1:
            with open(pred_file.parent / f"{self.rank}_{pred_file.name}", "wb") as f:
                pickle.dump(predictions, f)
            if self.rank == 0 and pred_file.exists():
                pred_file.unlink()
        else:
            with open(pred_file, "wb") as f:
                pickle.dump(predictions, f)

    def fit(self, epochs: int, lr: float):
        if self.optimizer is None:
            self.prepare_optimizer(lr)
        for _ in range(epochs):
            self.num_epoch += 1
            tr = self.train_epoch()
            va, preds = self.validate(with_predictions=True)
            if self.lr_scheduler is not None:
                self.lr_scheduler.step(va["Acc"])
            if self.rank == 0:
                line = f"epoch {self.num_epoch} it {self.num_it} | " + " ".join(f"trn_{k} {v:.4f}" for k, v in tr.items()) + " | " + \
                       " ".join(f"val_{k} {v:.4f}" for k, v in va.items())
                print(line, flush=True)
                with open(self.log_dir / f"{self.uid}.txt", "a") as f:
                    f.write(line + "\n")
            if self.best_met < va[self.met_keys[0]]:       # checkpoint + predictions only on improvement (utils.py:606-611)
                self.best_met = va[self.met_keys[0]]
                self.save_model_dict()
                self.update_prediction_file(preds, self.predictions_dir / f"val_preds_{self.uid}.pkl")
        return tr, va

    def testing(self, dls):
        dls = dls if isinstance(dls, dict) else {"valid": dls}
        out = {}
        for name, dl in dls.items():
            out[name], preds = self.validate(dl, with_predictions=True)
            self.update_prediction_file(preds, self.predictions_dir / f"{name}_preds.pkl")      # utils.py:664-665
            if self.rank == 0:
                print(f"test {name}: " + " ".join(f"{k} {v:.4f}" for k, v in out[name].items()), flush=True)
        return out
