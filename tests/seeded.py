"""Seeded weights for the large-shape parity fixtures (test infrastructure).

The headline model (d_model 512, 6+6 layers) has 32.5 M parameters = 130 MB: too large to commit.
Instead both sides re-create the SAME state_dict from a seed: ``tests/golden/make_golden_large.py``
loads it into the real reference model (build container) and stores only outputs; the tests load it
into the oracle and into the HIP model.  numpy's ``default_rng`` (PCG64) bit stream and the float
arithmetic below are platform independent, and every tensor has its own stream keyed by its name, so
neither the iteration order nor the set of keys matters.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch


def seeded_tensor(key: str, shape, seed: int, gains=None) -> torch.Tensor:
    rng = np.random.default_rng([int(seed), zlib.crc32(key.encode())])
    shape = tuple(int(s) for s in shape)
    gain = 1.0
    for pat, g in (gains or {}).items():
        if pat in key:
            gain *= g
    if len(shape) >= 2:
        fan_out, fan_in = shape[0], shape[1]
        bound = np.sqrt(6.0 / (fan_in + fan_out))              # xavier-uniform, as the reference's init (models.py:78-83)
        a = rng.uniform(-bound, bound, size=shape)
    elif "norm" in key and key.endswith("weight"):
        a = 1.0 + 0.2 * rng.standard_normal(size=shape)
    else:
        a = 0.1 * rng.standard_normal(size=shape)
    return torch.from_numpy((a * gain).astype(np.float32))


def seeded_state_dict(key_shapes, seed: int, gains=None) -> "OrderedDict[str, torch.Tensor]":
    """key_shapes: iterable of (name, shape) - e.g. ``((k, v.shape) for k, v in model.state_dict().items())``."""
    return OrderedDict((k, seeded_tensor(k, s, seed, gains)) for k, s in key_shapes)


# gains that make a random-init model decode diverse tokens / fire pointers (random-init models
# collapse to one repeated token): larger embeddings -> hidden states follow the inputs; larger heads -> wider margins
LARGE_GAINS = {"input_embeddings.": 4.0, "query_": 4.0, "vocab_head.weight": 6.0, "pointer_head.weight": 24.0,
               "switch_head.weight": 2.0}
