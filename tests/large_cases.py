"""Case definitions shared by tests/golden/make_golden_large.py (which runs the real reference) and the parity
tests (oracle on the CPU, HIP model on the GPU): shapes, seeds, batches and the gradient summary format."""
from __future__ import annotations

import numpy as np
import torch

from plankassembly_amd.data import SynthSpec, synth_batch

# gains applied on top of the seeded xavier / normal init (tests/seeded.py).  A plain random-init model collapses:
# attention averages put the same large vector into every row, the token-dependent part of the residual stream decays
# layer by layer and greedy decoding repeats one token.  Damped attention/FFN output
# projections and larger embeddings / heads keep the rows distinct, so the decode fixture has diverse tokens, firing
# pointers and wide top-2 margins.
GAINS = {"input_embeddings.": 8.0, "query_": 16.0, "vocab_head.weight": 6.0, "pointer_head.weight": 24.0,
         "switch_head.weight": 1.0, ".bias": 0.2, "out_proj.weight": 0.15, "multihead_attn.out_proj.weight": 3.0,
         "linear2.weight": 0.5}
SLICE = (8, 32)      # leading rows x cols of every gradient kept in the fixtures

BIG = dict(d=512, h=8, ff=1024, ne=6, nd=6, gains=GAINS)
CASES = {
    "headline": dict(BIG, max_in=1025, max_out=128, B=2, wseed=58, bseed=5, lines=(8, 255), with_type=True,
                     decode_b=4, decode_seed=6),
    # the reference's shipped default, configs/train_complete.yaml:41-42 (MAX_INPUT_LENGTH 1200 -> S = 1199, 300 rows of
    # input_pos; MAX_OUTPUT_LENGTH 128) = BASELINE configs[1] at model level
    "complete": dict(BIG, max_in=1200, max_out=128, B=2, wseed=31, bseed=41, lines=(8, 299), with_type=True),
    "visible": dict(BIG, max_in=1000, max_out=128, B=2, wseed=12, bseed=7, lines=(8, 249), with_type=True),
    "sideface": dict(BIG, max_in=300, max_out=128, B=16, wseed=13, bseed=9, lines=(0, 74), with_type=False, empty_rows=(3, 11)),
    # BASELINE configs[4] / SURVEY 8d "T = 1024 variant": MAX_OUTPUT_LENGTH 1024.  Train step with T = 1024 (171 rows of
    # query_pos_embedding, 514 + 1024 labels) and a greedy decode that runs ALL 1024 steps: `no_end` pushes the END logit
    # down (vocab_head.bias[512] -= 50) so that no row ever samples END and the reference's own loop (models.py:267-307)
    # cannot stop early.
    # Its head gains differ from GAINS: with 1 000 pointer candidates the pointer softmax is flat unless the pointer logits
    # are as peaked as the vocabulary logits; with these the pointer fires 119 times (81 times past step 128, targets up to
    # step 223+) and the smallest relative top-2 margin of the 2 048 sampled rows is 2.1e-4.
    "t1024": dict(BIG, max_in=1025, max_out=1024, B=2, wseed=58, bseed=21, lines=(8, 255), planks=(2, 170), with_type=True,
                  decode_b=2, decode_seed=22, no_end=True,
                  gains=dict(GAINS, **{"pointer_head.weight": 120.0, "vocab_head.weight": 4.0})),
    "live": dict(d=64, h=4, ff=128, ne=2, nd=2, gains={}, max_in=65, max_out=36, B=4, wseed=3, bseed=2022,
                 lines=(3, 15), planks=(2, 5), with_type=True, all_grads=True),
    # NORMALIZE_BEFORE: False (reference models.py:60-62: the flag lands in torch's layer_norm_eps slot -> per-layer LayerNorm
    # eps = 0.0, and no encoder.norm is built; layers stay post-norm).  No shipped config uses it; it is part of build_model's
    # surface.  Larger LayerNorm inputs than `live` so that eps = 0 vs 1 is far outside the tolerance.
    "eps0": dict(d=64, h=4, ff=128, ne=2, nd=2, gains={}, max_in=65, max_out=36, B=4, wseed=5, bseed=2023,
                 lines=(3, 15), planks=(2, 5), with_type=True, all_grads=True, normalize_before=False),
    # ACTIVATION: gelu (reference models.py:60-61,66-67 hand cfg.MODEL.ACTIVATION to torch's Transformer layers; no shipped config
    # uses it): train step with every gradient + a greedy decode, GELU in torch's exact erf form
    "gelu": dict(d=64, h=4, ff=128, ne=2, nd=2, gains=GAINS, max_in=65, max_out=36, B=4, wseed=7, bseed=2024,
                 lines=(3, 15), planks=(2, 5), with_type=True, all_grads=True, activation="gelu", decode_b=4, decode_seed=8),
}


def make_empty_rows(batch, rows):
    """Turn the given samples into the side-face 'nothing detected' input: [END, PAD, PAD, ...] with zero ids
    (reference sideface_data.py:137-213 with no faces)."""
    for r in rows:
        for k in batch:
            if k.startswith("input") and k != "input_mask":
                batch[k][r] = 0
        batch["input_value"][r] = 513
        batch["input_value"][r, 0] = 512
        batch["input_mask"][r] = batch["input_value"][r] == 513
    return batch


def case_batch(c, decode=False, batch_size=None):
    spec = SynthSpec(c["max_in"], c["max_out"], c["lines"], c.get("planks", (2, 21)), c["with_type"])
    B = batch_size or (c["decode_b"] if decode else c["B"])
    b = synth_batch(B, spec, seed=c["decode_seed"] if decode else c["bseed"])
    b.pop("name")
    if c.get("empty_rows") and not decode:
        make_empty_rows(b, [r for r in c["empty_rows"] if r < B])
    return b


def grad_summary(grads):
    """{name: tensor} -> flat dict of per-parameter L2 norm / sum (float64) and the leading SLICE."""
    res = {}
    for n, g in grads.items():
        g = g.detach().to(torch.float32).cpu()
        g2 = g.reshape(g.shape[0], -1) if g.dim() > 1 else g.reshape(1, -1)
        res["gnorm::" + n] = np.float64(g.double().norm().item())
        res["gsum::" + n] = np.float64(g.double().sum().item())
        res["gmax::" + n] = np.float64(g.abs().max().item())
        res["gslice::" + n] = g2[:SLICE[0], :SLICE[1]].numpy().copy()
    return res


def case_shapes(c):
    """(name, shape) of every state_dict entry of the case's model (from the drop-in module's own layout, which
    tests/test_model_surface.py pins to the reference's key names, shapes and order)."""
    import types
    from plankassembly_amd.models import PlankModel
    m = PlankModel(c["d"], c["h"], c["ff"], 0.0, c.get("activation", "relu"), c.get("normalize_before", True), c["ne"], c["nd"], 3, 2, 4, 6,
                   c["max_in"], c["max_out"], 514, types.SimpleNamespace(END=512, PAD=513))
    return [(k, tuple(v.shape)) for k, v in m.state_dict().items()]


def suppress_end(sd):
    """END can never be the argmax: the 1024-step decode fixture must run every step (see CASES['t1024'])."""
    sd["vocab_head.bias"] = sd["vocab_head.bias"].clone()
    sd["vocab_head.bias"][512] -= 50.0
    return sd


def case_state_dict(c):
    from seeded import seeded_state_dict
    sd = seeded_state_dict(case_shapes(c), c["wseed"], c["gains"])
    return suppress_end(sd) if c.get("no_end") else sd


def case_oracle_cfg(c):
    from oracle import plank_oracle as O
    return O.OracleCfg(d_model=c["d"], n_head=c["h"], d_ff=c["ff"], n_enc=c["ne"], n_dec=c["nd"],
                       max_input_length=c["max_in"], max_output_length=c["max_out"],
                       normalize_before=c.get("normalize_before", True), activation=c.get("activation", "relu"))


def load_large(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"fixture_{name}.npz"))
    return {k: z[k] for k in z.files}
