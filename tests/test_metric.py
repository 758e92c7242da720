"""G9: matcher / metric parity with the reference's third_party.matcher on fixed box sets."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from plankassembly_amd.metric import Criterion, HungarianMatcher, pairwise_iou_3d


def test_matcher_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN, "matcher.npz"))
    m = HungarianMatcher(0.5)
    for i in range(int(z["n"])):
        p, r, f = m(torch.as_tensor(z[f"pred{i}"]), torch.as_tensor(z[f"gt{i}"]))
        assert np.allclose([float(p), float(r), float(f)], z[f"prf{i}"], atol=1e-7), i


def test_iou_exactly_half_counts_as_tp_but_gets_no_cost_bonus():
    iou = pairwise_iou_3d(np.array([[0, 0, 0, 2, 1, 1]]), np.array([[0, 0, 0, 1, 1, 1]]))
    assert iou[0, 0] == 0.5
    p, r, f = HungarianMatcher(0.5)(np.array([[0, 0, 0, 2, 1, 1]]), np.array([[0, 0, 0, 1, 1, 1]]))
    assert float(p) == 1.0 and float(r) == 1.0


def test_criterion_running_mean():
    c = Criterion()
    c.update(1.0, 0.5, 2 / 3)
    c.update(0.0, 0.0, 0.0)
    p, r, f = c.compute(sync=False)
    assert abs(float(p) - 0.5) < 1e-12 and abs(float(r) - 0.25) < 1e-12 and abs(float(f) - 1 / 3) < 1e-12
