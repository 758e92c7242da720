"""Golden vectors for a training step UNDER DROPOUT, produced by the REAL reference model (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dropout.py

torch draws dropout masks from its Philox stream, which nothing else can reproduce; what CAN be pinned is WHERE the
reference applies dropout and what it does with a mask.  This script builds the reference `PlankModel` with dropout 0.2 on
fixture_small's weights and batch (and on fixture_ragged's: no input_type, an empty row), replaces torch's two dropout entry
points by a rule that depends only on the order of the calls and the tensor shape

    keep(call n, shape) = torch.rand(shape, generator=Generator().manual_seed(7000 + n)) >= p,   survivors * 1 / (1 - p)

- `torch.nn.functional.dropout` (nn.Dropout: dropout1/2/3 and the feed-forward dropout of the Transformer layers) and
`torch.nn.functional.scaled_dot_product_attention` (attention-probability dropout: `F.multi_head_attention_forward` calls
it with `dropout_p`; the replacement is softmax(QK^T / sqrt(dh) + mask) -> rule -> @ V, which is the explicit path of the
reference's pinned torch 1.10) - runs forward + backward and stores loss, accuracy, hiddens and every gradient in
fixture_dropout.npz.  tests/test_oracle_golden.py replays the same rule through `oracle.plank_oracle._drop`: the oracle's
dropout SITES and their ORDER are thereby pinned against the reference, and the GPU tests compare the HIP step under dropout
with that oracle (tests/test_headline_gpu.py, masks from tests/dropout_masks.py)."""
import math
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.nn.functional as F

from plankassembly.models import PlankModel          # the reference (namespace package)

TOKEN = types.SimpleNamespace(END=512, PAD=513)
P = 0.2


class Rule:
    def __init__(self):
        self.n, self.shapes = 0, []

    def mask(self, shape, dtype):
        keep = torch.rand(tuple(shape), generator=torch.Generator().manual_seed(7000 + self.n)) >= P
        self.n += 1
        self.shapes.append(tuple(shape))
        return keep.to(dtype) / (1.0 - P)


def run(fixture, with_type):
    z = np.load(os.path.join(HERE, fixture))
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    batch = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch::")}
    m = PlankModel(64, 4, 128, P, "relu", True, 2, 2, 3, 2, 4, 6, 65, 36, 514, TOKEN)
    m.load_state_dict(sd)
    m.train()
    rule = Rule()

    def dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        assert abs(p - P) < 1e-12
        return x * rule.mask(x.shape, x.dtype)

    def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
        assert not is_causal and not kw.get("enable_gqa", False)
        s = (q @ k.transpose(-1, -2)) * (scale if scale is not None else 1.0 / math.sqrt(q.shape[-1]))
        if attn_mask is not None:
            s = s + attn_mask if attn_mask.dtype != torch.bool else s.masked_fill(~attn_mask, float("-inf"))
        a = torch.softmax(s, dim=-1)
        if dropout_p > 0.0:
            assert abs(dropout_p - P) < 1e-12
            a = a * rule.mask(a.shape, a.dtype)
        return a @ v

    old = F.dropout, F.scaled_dot_product_attention
    F.dropout, F.scaled_dot_product_attention = dropout, sdpa
    try:
        out = m(batch)
        out["loss"].backward()
    finally:
        F.dropout, F.scaled_dot_product_attention = old
    res = {"loss": np.float32(out["loss"].item()), "accuracy": np.float32(float(out["accuracy"])),
           "calls": np.int64(rule.n), "shapes": np.array([s + (0,) * (4 - len(s)) for s in rule.shapes], dtype=np.int64)}
    for n, p in m.named_parameters():
        res["grad::" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    assert rule.n == 4 * 2 + 6 * 2, rule.n
    return res


def main():
    out = {}
    for tag, fixture, with_type in (("small", "fixture_small.npz", True), ("ragged", "fixture_ragged.npz", False)):
        for k, v in run(fixture, with_type).items():
            out[f"{tag}::{k}"] = v
        print(tag, "loss", out[f"{tag}::loss"], "calls", out[f"{tag}::calls"])
    np.savez_compressed(os.path.join(HERE, "fixture_dropout.npz"), **out)


if __name__ == "__main__":
    main()
