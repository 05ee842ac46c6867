"""Configuration (reference `code/extended_config.py` + `configs/cfg.json`): the same ~40 flags with the same defaults,
addressable as cfg['k'] and cfg.k, and the same override rule (key must exist, value must keep its type;
extended_config.py:78-88).  yacs is not available offline, so this is a small dict subclass."""
import ast
import copy
from typing import Any, Dict

DEFAULTS: Dict[str, Any] = {
    # configs/cfg.json:1-43
    "ds_to_use": "refclef", "bs": 16, "nw": 4, "bsv": 16, "nwv": 4, "lr": 1e-4, "devices": 0, "opt_fn": "Adam",
    "opt_fn_params": {"betas": [0.9, 0.99]}, "do_norm": False, "use_same_atb": True, "mdl_to_use": "retina",
    "resize_img": [300, 300], "tmp_path": "./tmp", "use_multi": True, "use_focal": True, "use_softmax": False,
    "alpha": 0.25, "gamma": 2, "ratios": "[1/2, 1, 2]", "scales": "[1, 2**(1/3), 2**(2/3)]", "scale_factor": 4,
    "emb_dim": 300, "matching_threshold": 0.6, "epochs": 10, "use_bidirectional": True, "lstm_dim": 128,
    "use_reduce_lr_plateau": True, "patience": 2, "reduce_factor": 0.1, "lamb_reg": 1, "resume_path": "",
    "resume": True, "load_opt": False, "strict_load": True, "load_normally": True, "acc_iou_threshold": 0.5,
    "use_lang": True, "use_img": True,
    # extended_config.py:13-21
    "device": "cuda", "local_rank": 0, "do_dist": False, "only_val": False, "only_test": False, "num_gpus": 1,
    # extensions of this build (documented in DESIGN.md)
    "resnet_arch": "resnet50",      # the reference hard-codes resnet50 (mdl.py:411); configs 1 and 5 need 18 / 101
    "pretrained_path": "",          # local checkpoint instead of the torchvision download
    "synthetic": True,              # synthetic batches instead of the CSV datasets
    "steps_per_epoch": 50,
    "use_hip_graph": False,
    "word_vectors": "",             # .npz word-vector table (`words`, `vectors`) used when spaCy is not installed
    "gpu_img_normalise": True,      # images travel as uint8 HWC; /255 + NHWC4 on the GPU (bit-identical to the host path)
    # configs/ds_info.json: where each dataset's images and csv files live (override with --ds_info.<name>.<key>=...)
    "ds_info": {name: {"data_dir": f"./data/{root}", "img_dir": f"./data/{imgs}",
                       **{f"{s}_csv_file": f"./data/{csv}/csv_dir/{f}.csv" for s, f in (("trn", trn), ("val", "val"), ("test", "test"))}}
                for name, root, imgs, csv, trn in (
                    ("flickr30k", "flickr30k", "flickr30k/flickr30k_images", "flickr30k", "train_flat"),
                    ("refclef", "referit/refclef", "referit/saiapr_tc12_images", "referit", "train_flat"),
                    ("flickr30k_c0", "flickr30k", "flickr30k/flickr30k_images", "flickr30k_c0", "train"),
                    ("flickr30k_c1", "flickr30k", "flickr30k/flickr30k_images", "flickr30k_c1", "train"),
                    ("vg_split_c2", "visual_genome/vg_split", "visual_genome", "vg_split_c2", "train"),
                    ("vg_split_c3", "visual_genome/vg_split", "visual_genome", "vg_split_c3", "train"))},
}


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.get("_frozen", False) and k != "_frozen":
            raise AttributeError("cfg is frozen")
        self[k] = v

    def freeze(self):
        dict.__setitem__(self, "_frozen", True)

    def clone(self) -> "Cfg":
        c = Cfg(copy.deepcopy({k: v for k, v in self.items() if k != "_frozen"}))
        return c


def get_cfg(**overrides) -> Cfg:
    cfg = Cfg(copy.deepcopy(DEFAULTS))
    return update_from_dict(cfg, overrides)


def _decode(v):
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def update_from_dict(cfg: Cfg, dct: Dict[str, Any], key_maps: Dict[str, str] = None) -> Cfg:
    """extended_config.py:46-90: every key must already exist and keep its type."""
    for full_key, v in dct.items():
        d = cfg
        parts = full_key.split(".")
        for sub in parts[:-1]:
            assert sub in d, f"key {full_key} doesnot exist"
            d = d[sub]
        sub = parts[-1]
        assert sub in d, f"key {full_key} doesnot exist"
        old = d[sub]
        val = v if isinstance(old, str) else _decode(v)
        if isinstance(old, float) and isinstance(val, int) and not isinstance(val, bool):
            val = float(val)
        assert isinstance(val, type(old)), f"key {full_key}: expected {type(old).__name__}, got {type(val).__name__}"
        d[sub] = val
    return cfg


def ratios_scales(cfg):
    """main_dist.py:24-31: ratios / scales are strings in the json and are eval'd."""
    import numpy as np
    ratios = eval(cfg["ratios"], {}) if not isinstance(cfg["ratios"], list) else cfg["ratios"]
    sc = eval(cfg["scales"], {}) if not isinstance(cfg["scales"], list) else cfg["scales"]
    return ratios, cfg["scale_factor"] * np.array(sc)
