"""Box matching metric of the evaluation callers.

``HungarianMatcher`` follows reference third_party/matcher.py:15-78 (3-D axis-aligned IoU between
predicted and ground-truth planks, Hungarian assignment with cost -1 where IoU > threshold, TP
counted where the matched IoU >= threshold) and ``Criterion`` follows reference
plankassembly/metric.py:6-30 (running sums of precision / recall / F1 and a count, summed over
ranks).  Written from scratch on numpy/scipy: <= 21 boxes per sample, CPU work by nature.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from .distributed import allreduce_metric_sums

LARGE_COST = 100000


def pairwise_iou_3d(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a [N,6], b [M,6] as (x1,y1,z1,x2,y2,z2) -> IoU [N,M] (0 where the union is empty)."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 6)
    b = np.asarray(b, dtype=np.float64).reshape(-1, 6)
    va = np.prod(a[:, 3:] - a[:, :3], axis=1)
    vb = np.prod(b[:, 3:] - b[:, :3], axis=1)
    lo = np.maximum(a[:, None, :3], b[None, :, :3])
    hi = np.minimum(a[:, None, 3:], b[None, :, 3:])
    inter = np.prod(np.clip(hi - lo, 0, None), axis=2)
    union = va[:, None] + vb[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(inter > 0, inter / union, 0.0)
    return iou


class HungarianMatcher:
    def __init__(self, threshold: float = 0.5):
        assert threshold != 0, "threshold cant be 0"
        self.threshold = threshold

    def __call__(self, pred_boxes, boxes):
        pb = pred_boxes.detach().cpu().numpy() if torch.is_tensor(pred_boxes) else np.asarray(pred_boxes)
        gb = boxes.detach().cpu().numpy() if torch.is_tensor(boxes) else np.asarray(boxes)
        n_pred, n_gt = len(pb), len(gb)
        iou = pairwise_iou_3d(pb, gb)
        cost = np.full((n_pred, n_gt), LARGE_COST)
        cost[iou > self.threshold] = -1
        r, c = linear_sum_assignment(cost)
        tp = float(np.sum(iou[r, c] >= self.threshold))
        prec = torch.tensor(tp / n_pred if n_pred else 0.0)
        rec = torch.tensor(tp / n_gt if n_gt else 0.0)
        f1 = prec * rec * 2 / (prec + rec + 1e-10)
        return prec, rec, f1


def build_matcher(threshold):
    return HungarianMatcher(threshold)


class Criterion:
    """Stand-in for the reference's torchmetrics Metric (torchmetrics is not installed here)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.sums = torch.zeros(4, dtype=torch.float64)      # precision, recall, fmeasure, total

    def update(self, prec, rec, f1):
        self.sums += torch.tensor([float(prec), float(rec), float(f1), 1.0], dtype=torch.float64)

    def compute(self, sync=True):
        s = self.sums.clone()
        if sync:
            s = allreduce_metric_sums(s)
        total = s[3].clamp(min=1.0) if s[3] == 0 else s[3]
        return s[0] / total, s[1] / total, s[2] / total


def build_criterion():
    return Criterion()


class PlankScorer:
    """One sample at a time: match predicted planks against ground-truth planks (row 0 of both is the overall bounding
    box and takes no part - reference trainer_complete.py:80, evaluate.py:54), feed the running means, hand the three
    numbers back.  The validation / test hooks of the trainers and the offline re-scoring of evaluate.py all go
    through here, so a sample is scored the same way wherever it is scored."""

    def __init__(self, threshold: float):
        self.matcher = build_matcher(threshold)
        self.criterion = build_criterion()

    def add(self, planks, truth):
        scores = self.matcher(planks[1:], truth[1:])
        self.criterion.update(*scores)
        return {"precision": float(scores[0]), "recall": float(scores[1]), "fmeasure": float(scores[2])}

    def means(self, sync=True):
        """(precision, recall, fmeasure) averaged over every sample added since the last call; resets."""
        out = tuple(float(x) for x in self.criterion.compute(sync=sync))
        self.criterion.reset()
        return out
