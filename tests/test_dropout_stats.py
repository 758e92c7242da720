"""CPU statistics of the Linear-output dropout decisions (tests/dropout_masks.py restates csrc/pa_device.h drop_keep_rc; the GPU
tests pin the restatement to the kernels' decisions bit for bit).  ADVICE r4: with one 24 x 24-bit product per decision, rows that
share a 22-bit hash shared their whole mask (correlation 0.9999 among the first 2048 rows, ~100 pairs above 0.2; column pairs up
to 0.85).  The two-product form must keep every pair at the level of independent Bernoulli draws."""
import numpy as np

import dropout_masks as DM


def _max_offdiag_corr(x):
    x = x.astype(np.float64)
    x = x - x.mean(axis=1, keepdims=True)
    x /= np.linalg.norm(x, axis=1, keepdims=True) + 1e-30
    c = x @ x.T
    np.fill_diagonal(c, 0.0)
    return float(np.abs(c).max()), int((np.abs(c) > 0.2).sum() // 2)


def test_linear_dropout_masks_of_different_rows_and_columns_are_uncorrelated():
    rows, cols, p = 2048, 1024, 0.2
    for seed in (987654321, 4711, 20240917):
        keep = DM.linear_keep(seed, np.arange(rows), cols, p)
        assert abs(keep.mean() - (1 - p)) < 2e-3
        # independent draws: correlation of two rows over 1024 columns ~ N(0, 1/32); the maximum over 2.1 M pairs sits near 0.17
        worst_r, n_r = _max_offdiag_corr(keep)
        worst_c, n_c = _max_offdiag_corr(keep.T[:1024])
        assert worst_r < 0.22 and n_r <= 2, (seed, worst_r, n_r)
        assert worst_c < 0.16 and n_c == 0, (seed, worst_c, n_c)          # columns: 2048 samples each, sigma = 1/45
        # per-row / per-column keep rates: binomial spread only
        assert np.abs(keep.mean(axis=1) - 0.8).max() < 6 * np.sqrt(0.16 / cols)
        assert np.abs(keep.mean(axis=0) - 0.8).max() < 6 * np.sqrt(0.16 / rows)
