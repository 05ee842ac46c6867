"""Offline accuracy from saved predictions — counterpart of the reference's code/eval_script.py:19-56.

Predictions: the pickle written by `Learner.update_prediction_file` — a list of
{'id': row index into the ground-truth CSV, 'pred_boxes': [x1, y1, x2, y2] pixels, 'pred_scores': float}.
Ground truth: a CSV with a `bbox` column holding "[x1, y1, x2, y2]" (DATA_PREP_README.md:10-11).
A prediction is correct when IoU > acc_iou_thresh — STRICT, unlike the in-loop metric's >= (evaluator.py:117); each
id counts once (DDP's padded sampler repeats samples).  When `pred_file` is missing, the per-rank files
'<rank>_<name>' of a `num_gpus`-rank run are merged into it first (eval_script.py:22-33).

Host-side bookkeeping (pandas + a scalar IoU per row), not part of the GPU hot path.
    python -m zsgnet_pytorch_amd.eval_script <pred_file> <gt_file> [--acc_iou_thresh=0.5] [--num_gpus=N]
"""
import ast
import pickle
import sys
from pathlib import Path

import numpy as np


def box_iou(pred, gt) -> float:
    """IoU of two x1y1x2y2 boxes in fp32 with the reference's operation order (anchors.py:90-116):
    inter / (area_a + area_b - inter + 1e-8)."""
    f = np.float32
    p, g = np.asarray(pred, dtype=f), np.asarray(gt, dtype=f)
    w = max(f(min(p[2], g[2]) - max(p[0], g[0])), f(0))
    h = max(f(min(p[3], g[3]) - max(p[1], g[1])), f(0))
    inter = f(w * h)
    union = f(f(f((p[2] - p[0]) * (p[3] - p[1])) + f((g[2] - g[0]) * (g[3] - g[1]))) - inter)
    return float(f(inter / f(union + f(1e-8))))


def merge_rank_files(pred_file: Path, num_gpus: int) -> None:
    parts = [pred_file.parent / f"{r}_{pred_file.name}" for r in range(num_gpus)]
    missing = [str(p) for p in parts if not p.exists()]
    assert not missing, f"per-rank prediction files missing: {missing}"
    merged = []
    for p in parts:
        with open(p, "rb") as f:
            part = pickle.load(f)
        assert isinstance(part, list)
        merged += part
    with open(pred_file, "wb") as f:
        pickle.dump(merged, f)


def evaluate(pred_file, gt_file, **kwargs):
    """-> (accuracy, n_correct, n_total)"""
    import pandas as pd
    thr = float(kwargs.get("acc_iou_thresh", 0.5))
    pred_file = Path(pred_file)
    if not pred_file.exists():
        assert "num_gpus" in kwargs, f"{pred_file} does not exist: pass num_gpus=N to merge the per-rank files"
        merge_rank_files(pred_file, int(kwargs["num_gpus"]))
    with open(pred_file, "rb") as f:
        predictions = pickle.load(f)
    gt = pd.read_csv(gt_file)
    boxes = [ast.literal_eval(b) if isinstance(b, str) else b for b in gt["bbox"]]
    corr, tot, seen = 0, 0, set()
    for p in predictions:
        ind = int(p["id"])
        if ind in seen:
            continue
        seen.add(ind)
        corr += int(box_iou(p["pred_boxes"], boxes[ind]) > thr)
        tot += 1
    return corr / tot, corr, tot


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    pos = [a for a in argv if not a.startswith("--")]
    kw = {}
    for a in argv:
        if a.startswith("--"):
            k, _, v = a[2:].partition("=")
            kw[k] = ast.literal_eval(v) if v else True
    assert len(pos) == 2, __doc__
    acc, corr, tot = evaluate(pos[0], pos[1], **kw)
    print(f"Acc {acc:.6f} ({corr}/{tot})")
    return acc, corr, tot


if __name__ == "__main__":
    main()
