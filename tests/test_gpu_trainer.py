"""The Learner / CLI plumbing on the GPU with synthetic batches: train -> validate -> checkpoint on
improvement -> prediction pickles in the reference's format -> offline eval_script accuracy == in-loop accuracy."""
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_validate_predictions_and_eval_script(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from zsgnet_pytorch_amd import eval_script
    from zsgnet_pytorch_amd.main_dist import main_dist
    kw = dict(resnet_arch="resnet18", bs=2, bsv=2, resize_img=[96, 96], steps_per_epoch=10, epochs=1, tmp_path=str(tmp_path), synthetic=True)
    learn = main_dist("t0", **{k: str(v) for k, v in kw.items()})
    assert learn.num_it == 10 and learn.num_epoch == 1
    ck = torch.load(learn.model_file, map_location="cpu") if learn.model_file.exists() else None
    # predictions of the validation pass: written together with the checkpoint when the metric improved (utils.py:606-611)
    vp = learn.predictions_dir / "val_preds_t0.pkl"
    assert (ck is not None) == vp.exists()
    # test pass: always writes '<name>_preds.pkl' (utils.py:646-665)
    res = learn.testing(learn.data.test_dl)
    pf = learn.predictions_dir / "synthetic_test_preds.pkl"
    preds = pickle.load(open(pf, "rb"))
    dl = learn.data.test_dl["synthetic_test"]
    assert isinstance(preds, list) and len(preds) == len(dl) * 2 and set(preds[0]) == {"id", "pred_boxes", "pred_scores"}
    assert [int(p["id"]) for p in preds] == list(range(len(preds))) and len(preds[0]["pred_boxes"]) == 4
    # ground-truth CSV in pixels, x1y1x2y2 (what the reference CSVs hold): the evaluator's own conversion of `annot`
    dl.epoch -= 1                                            # replay the same batches
    rows = []
    for bt in dl:
        a, sz = bt["annot"].double(), bt["img_size"].double()
        px = (a + 1) / 2 * torch.cat([sz, sz], 1)            # y1 x1 y2 x2 in pixels
        rows += px[:, [1, 0, 3, 2]].tolist()
    with open(tmp_path / "gt.csv", "w") as f:
        f.write("img_id,bbox,query\n")
        for i, b in enumerate(rows):
            f.write(f'{i}.jpg,"{b}",q\n')
    acc, corr, tot = eval_script.evaluate(pf, tmp_path / "gt.csv")
    assert tot == len(preds)
    assert abs(acc - res["synthetic_test"]["Acc"]) < 1e-6, "offline accuracy must equal the in-loop metric (no box sits exactly at IoU 0.5)"
    # resume from the checkpoint + validation only
    if ck is not None:
        assert set(ck) >= {"model_state_dict", "optimizer_state_dict", "num_it", "num_epoch", "best_met"}
        again = main_dist("t0", **{k: str(v) for k, v in dict(kw, resume=True, only_val=True).items()})
        assert again.num_it == 10
