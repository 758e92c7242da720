"""Synthetic tokenised-drawing batches that follow the reference batch contract.

The layout mirrors what the reference's CPU dataloader hands the model
(reference plankassembly/datasets/line_data.py:34-109, sideface_data.py:137-213):

* every ``input_*`` row has length ``MAX_INPUT_LENGTH - 1``: 4 tokens per line,
  then END (512), then PAD (513); the id tensors are 0 at END/PAD;
* ``input_mask = input_value == PAD``;
* ``output_value`` has length ``MAX_OUTPUT_LENGTH``: 6 tokens per plank (plank 0 is
  the overall bounding box), END, PAD;
* ``output_label = 514 + attach`` where a pointer is attached, else the value;
* ``output_mask = output_value == PAD``.

Only the *distribution* is synthetic (there is no network for the real dataset);
the recipe is the one written down in SURVEY.md section 8(d).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

END = 512
PAD = 513
VOCAB = 514


def pointer_mask_row(i: int, sz: int) -> np.ndarray:
    """Closed form of the reference's ``_generate_pointer_mask`` row ``i`` (models.py:91-101)."""
    row = np.zeros(sz, dtype=bool)
    if i < 6:
        return row
    for j in range(sz):
        if j < 6:
            row[j] = j == i % 6
        else:
            row[j] = (j % 6) == ((i % 6) + 3) % 6
    return row


@dataclass
class SynthSpec:
    max_input_length: int = 1200
    max_output_length: int = 128
    n_lines: tuple = (8, 299)     # inclusive range of line (or side-face) count per sample
    n_planks: tuple = (2, 21)     # inclusive range of planks (incl. the bbox plank)
    with_type: bool = True        # sideface batches carry no ``input_type``
    attach_prob: float = 0.5
    valid_pointers: bool = True   # draw pointer targets from the reference's pointer mask


def synth_sample(rng: np.random.Generator, spec: SynthSpec):
    S = spec.max_input_length - 1
    T = spec.max_output_length
    lo, hi = spec.n_lines
    n = int(rng.integers(lo, hi + 1))
    n = min(n, (S - 1) // 4)
    views = np.sort(rng.integers(0, 3, size=n))
    values = rng.integers(0, 512, size=(n, 4))
    types = rng.integers(0, 2, size=n)
    if n:
        _, counts = np.unique(views, return_counts=True)
        pos = np.concatenate([np.arange(c) for c in counts])
    else:
        pos = np.zeros(0, dtype=np.int64)

    def padded(body, fill):
        out = np.full(S, fill, dtype=np.int64)
        out[: len(body)] = body
        return out

    value = padded(np.append(values.reshape(-1), END), PAD)
    sample = {
        "input_value": value,
        "input_pos": padded(np.repeat(pos, 4), 0),
        "input_coord": padded(np.arange(4 * n) % 4, 0),
        "input_view": padded(np.repeat(views, 4), 0),
    }
    if spec.with_type:
        sample["input_type"] = padded(np.repeat(types, 4), 0)
    sample["input_mask"] = value == PAD

    plo, phi = spec.n_planks
    p = int(rng.integers(plo, phi + 1))
    p = min(p, (T - 1) // 6)
    seq = rng.integers(0, 512, size=6 * p)
    attach = np.full(6 * p, -1, dtype=np.int64)
    for i in range(6, 6 * p):
        if rng.random() < spec.attach_prob:
            if spec.valid_pointers:
                cand = np.nonzero(pointer_mask_row(i, i))[0]
            else:
                cand = np.arange(i)
            if len(cand):
                j = int(rng.choice(cand))
                attach[i] = j
                seq[i] = seq[j]
    out_value = np.full(T, PAD, dtype=np.int64)
    out_value[: 6 * p] = seq
    out_value[6 * p] = END
    label = np.full(T, -1, dtype=np.int64)
    label[: 6 * p] = attach
    lab = np.where(label != -1, label + VOCAB, out_value)
    sample["output_value"] = out_value
    sample["output_label"] = lab
    sample["output_mask"] = out_value == PAD
    return sample


def synth_batch(batch_size: int, spec: SynthSpec, seed: int = 2022, device=None):
    """A collated batch (dict of LongTensor/BoolTensor [B, L]) plus ``name``."""
    rng = np.random.default_rng(seed)
    samples = [synth_sample(rng, spec) for _ in range(batch_size)]
    batch = {"name": [f"synth_{seed}_{i:04d}" for i in range(batch_size)]}
    for key in samples[0]:
        arr = np.stack([s[key] for s in samples])
        t = torch.from_numpy(arr)
        batch[key] = t.to(device) if device is not None else t
    return batch


def spec_for(kind: str, max_input_length=None, max_output_length=None) -> SynthSpec:
    """Named workloads of SURVEY.md section 8(d) / BASELINE.json ``configs``."""
    if kind == "complete":
        s = SynthSpec(1200, 128, (8, 299), (2, 21), True)
    elif kind == "visible":
        s = SynthSpec(1000, 128, (8, 249), (2, 21), True)
    elif kind == "sideface":
        s = SynthSpec(300, 128, (0, 74), (2, 21), False)
    elif kind == "headline":          # d_model=512, seq=1024 (BASELINE.json metric)
        s = SynthSpec(1025, 128, (8, 255), (2, 21), True)
    elif kind == "decode":            # batch 256, max_len 1024
        s = SynthSpec(1025, 1024, (255, 255), (2, 21), True)
    else:
        raise ValueError(kind)
    if max_input_length is not None:
        s.max_input_length = max_input_length
    if max_output_length is not None:
        s.max_output_length = max_output_length
    return s


class DevicePrefetcher:
    """Iterate prepared batches one step ahead of the training loop.

    What runs where (round 5, tools/prep_probe.py on MI355X: 200 headline steps, bf16, batch 16):
    * the host -> device copies of batch i + 1 always run on a side stream while step i computes (the reference gets the same
      overlap from its DataLoader workers, trainer_complete.py:35-43);
    * when the number of valid encoder rows is known on the HOST - the collate function built the padding mask on the CPU (the mask
      is a CPU tensor: counted here before the copy), or the batch carries ``_n_valid`` - the two preparation launches (row packing,
      embedding-row grouping) go to the MAIN stream in front of the step: +0.04 ms per step over batches prepared ahead;
    * otherwise (device-resident batch, count unknown) ``prepare_batch`` runs one step ahead on the side stream, because its device
      -> host read of the count must not drain the main stream: +0.10 ms per step - not the read (a host-known count on the side
      stream measures the same) but the two launches sharing the CUs with the step's one-block-per-CU grids.
    """

    def __init__(self, model, batches):
        self.model = model
        self.it = iter(batches)
        self.side = torch.cuda.Stream()
        self.nxt = None
        self._stage()

    def _stage(self):
        try:
            raw = next(self.it)
        except StopIteration:
            self.nxt = None
            return
        msk = raw.get("input_mask")
        known = raw.get("_n_valid")
        if known is None and torch.is_tensor(msk) and not msk.is_cuda and getattr(self.model, "unpad", False):
            known = int(msk.numel() - int(msk.to(torch.bool).sum()))       # CPU collate: the count is free
        # (no wait on the main stream: a collated batch does not depend on the step in flight, and waiting would park
        # prepare_batch's device -> host read behind that whole step)
        with torch.cuda.stream(self.side):
            if known is not None:
                dev = self.model._flat.device
                moved = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in raw.items()}
                moved["_n_valid"] = known
                self.nxt = ("main", moved)
            else:
                self.nxt = ("done", self.model.prepare_batch(raw))

    def __iter__(self):
        return self

    def __next__(self):
        if self.nxt is None:
            raise StopIteration
        main = torch.cuda.current_stream()
        main.wait_stream(self.side)
        kind, cur = self.nxt
        for t in _tensors_of(cur):
            t.record_stream(main)                                # allocated on the side stream, consumed on the main one
        if kind == "main":
            cur = self.model.prepare_batch(cur)                  # two launches on the main stream, no device -> host read
        self._stage()
        return cur


def _tensors_of(obj):
    if torch.is_tensor(obj):
        if obj.is_cuda:
            yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors_of(v)
