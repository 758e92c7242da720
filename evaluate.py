"""`python evaluate.py --data_path DIR --exp_path lightning_logs/version_X [--threshold 0.5] [--num_bits 9]`

Same command line, inputs and outputs as the reference's evaluate.py:15-79: every `pred_jsons/<name>.json` written
by `trainer_*.py test` is dequantised (data_utils.py:15-21) and re-scored against the CONTINUOUS ground truth
`<data_path>/infos/<name>.json` (`coords`) with the Hungarian box matcher; per-sample precision / recall / F1 go to
`<exp_path>/metrics.json`, the means are printed in percent.  Empty predictions (side-face samples with nothing
detected, trainer_sideface.py:46-52) are skipped, as in the reference.
"""
import argparse
import json
import os

import numpy as np
import torch

from plankassembly_amd.datasets import dequantize_values
from plankassembly_amd.metric import build_criterion, build_matcher


def evaluate(data_path, exp_path, threshold=0.5, num_bits=9, verbose=True):
    filenames = sorted(os.listdir(os.path.join(exp_path, "pred_jsons")))
    matcher = build_matcher(threshold)
    criterion = build_criterion()
    metrics = dict()
    for filename in filenames:
        name = filename.split(".")[0]
        with open(os.path.join(exp_path, "pred_jsons", filename)) as f:
            pred_data = json.load(f)
        with open(os.path.join(data_path, "infos", filename), "r") as f:
            gt_data = json.load(f)
        pred = np.array(pred_data["prediction"])
        if len(pred) == 0:
            continue
        pred = torch.from_numpy(dequantize_values(pred, num_bits))
        gt = torch.from_numpy(np.array(gt_data["coords"]))
        prec, recal, f1 = matcher(pred[1:], gt[1:])
        criterion.update(prec, recal, f1)
        metrics[name] = {"precision": float(prec), "recall": float(recal), "fmeasure": float(f1)}
    with open(os.path.join(exp_path, "metrics.json"), "w") as f:
        json.dump(metrics, f)
    prec, recal, fscore = criterion.compute(sync=False)
    if verbose:
        print("%10s %0.3f" % ("prec", prec * 100))
        print("%10s %0.3f" % ("rec", recal * 100))
        print("%10s %0.3f" % ("f1", fscore * 100))
    return float(prec), float(recal), float(fscore), metrics


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--data_path", metavar="DIR", default="data/data/complete", help="dataset source root.")
    parser.add_argument("--exp_path", type=str, default="lightning_logs/version_X", help="log path.")
    parser.add_argument("--threshold", type=float, default=0.5, help="threshold")
    parser.add_argument("--num_bits", type=int, default=9, help="number of bits")
    a = parser.parse_args()
    evaluate(a.data_path, a.exp_path, a.threshold, a.num_bits)
