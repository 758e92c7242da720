"""`python evaluate.py --data_path DIR --exp_path lightning_logs/version_X [--threshold 0.5] [--num_bits 9]`

Offline re-scoring with the reference's command line, inputs and outputs (reference evaluate.py:15-79): every
`<exp_path>/pred_jsons/<name>.json` written by `trainer_*.py test` is dequantised (data_utils.py:15-21) and scored
against the CONTINUOUS ground truth `<data_path>/infos/<name>.json` (`coords`); per-sample precision / recall / F1 go
to `<exp_path>/metrics.json` and the means are printed in percent.  Samples whose prediction is empty (side-face
inputs with nothing detected, trainer_sideface.py:46-52) take no part, as in the reference.

The scoring itself is `plankassembly_amd.metric.PlankScorer` - the same object the trainers' validation / test hooks
use - so this file only has to pair up the two directories.
"""
import argparse
import json
import os

import numpy as np
import torch

from plankassembly_amd.datasets import dequantize_values
from plankassembly_amd.metric import PlankScorer


def _read(path):
    with open(path) as f:
        return json.load(f)


def scored_pairs(data_path, exp_path, num_bits):
    """Yield (sample name, dequantised predicted planks, continuous ground-truth planks) for every prediction file
    that holds at least one plank, in file-name order."""
    pred_dir = os.path.join(exp_path, "pred_jsons")
    for entry in sorted(os.listdir(pred_dir)):
        quantised = np.array(_read(os.path.join(pred_dir, entry))["prediction"])
        if quantised.size == 0:
            continue
        truth = np.array(_read(os.path.join(data_path, "infos", entry))["coords"])
        yield entry.split(".")[0], torch.from_numpy(dequantize_values(quantised, num_bits)), torch.from_numpy(truth)


def evaluate(data_path, exp_path, threshold=0.5, num_bits=9, verbose=True):
    scorer = PlankScorer(threshold)
    per_sample = {name: scorer.add(planks, truth) for name, planks, truth in scored_pairs(data_path, exp_path, num_bits)}
    with open(os.path.join(exp_path, "metrics.json"), "w") as f:
        json.dump(per_sample, f)
    means = scorer.means(sync=False)
    if verbose:
        for label, value in zip(("prec", "rec", "f1"), means):
            print("%10s %0.3f" % (label, value * 100))
    return (*means, per_sample)


if __name__ == "__main__":
    cli = argparse.ArgumentParser()
    cli.add_argument("--data_path", metavar="DIR", default="data/data/complete", help="dataset source root.")
    cli.add_argument("--exp_path", type=str, default="lightning_logs/version_X", help="log path.")
    cli.add_argument("--threshold", type=float, default=0.5, help="threshold")
    cli.add_argument("--num_bits", type=int, default=9, help="number of bits")
    evaluate(**vars(cli.parse_args()))
