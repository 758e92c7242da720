"""Attribute-style config node (stand-in for detectron2's CfgNode, which the reference uses in
trainer_complete.py:26 and which is not installed here) and YAML loading for the reference's
LightningCLI config files (configs/train_*.yaml: top-level seed_everything / trainer / model.hparams)."""
from __future__ import annotations

import yaml


class CfgNode(dict):
    """dict with attribute access, recursively (the subset of detectron2.config.CfgNode the
    reference relies on: cfg.MODEL.NUM_MODEL, cfg.TOKEN.PAD, cfg.DATA ... )."""

    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e

    def __setattr__(self, name, value):
        self[name] = value


def load_cli_config(path):
    """Parse a reference-style LightningCLI YAML -> (seed, trainer kwargs dict, hparams dict)."""
    with open(path) as f:
        raw = yaml.safe_load(f)
    # PyYAML reads "1e-4" as a string (YAML 1.1); the reference's jsonargparse coerces it
    hp = dict(raw.get("model", {}).get("hparams", {}))
    if isinstance(hp.get("LR"), str):
        hp["LR"] = float(hp["LR"])
    return raw.get("seed_everything"), dict(raw.get("trainer", {})), hp
