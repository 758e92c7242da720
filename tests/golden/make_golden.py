"""Generate the golden vectors in tests/golden/ by running the REAL reference model.

Runs only in the build container (needs /root/reference; nothing of the reference is
copied -- only its inputs/outputs are stored as data).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures written:
  fixture_small.npz   d=64,H=4,dff=128,L=2+2,S=64,T=36,B=4: state_dict (briefly trained so
                      greedy decode emits diverse tokens, pointers and END), batch,
                      G1 memory/hiddens/dists/loss/accuracy, G2 all parameter grads,
                      G3 params after one torch.optim.Adam step, G4 samples/attach,
                      G6 _create_dist on fixed hiddens (train; eval sz=5,6,36)
  fixture_ragged.npz  G7: ragged padding batch incl. an "empty" row, sideface style
                      (no input_type): loss, grads of two tables, samples/attach
  fixture_tiny.npz    G8: BASELINE config 0 (d=128, 2+2, dff=256, B=4, S=1199): loss curve
  pointer_mask.npz    G5: _generate_pointer_mask(128)
  matcher.npz         G9: HungarianMatcher P/R/F1 on fixed box sets
"""
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

from plankassembly.models import PlankModel          # the reference (namespace package)
from plankassembly_amd.data import SynthSpec, synth_batch

TOKEN = types.SimpleNamespace(END=512, PAD=513)


def build(d, h, ff, ne, nd, max_in, max_out, dropout=0.0):
    torch.manual_seed(1234)
    m = PlankModel(d, h, ff, dropout, "relu", True, ne, nd, 3, 2, 4, 6, max_in, max_out, 514, TOKEN)
    # non-trivial 1-D parameters (torch leaves LN affine at 1/0 and MHA biases at 0)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 1:
                if "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return m


def model_batch(batch):
    return {k: v for k, v in batch.items() if k != "name"}


def np_state(m):
    return {"sd::" + k: v.detach().numpy().copy() for k, v in m.state_dict().items()}


def np_batch(batch):
    return {"batch::" + k: v.numpy() for k, v in batch.items() if k != "name"}


def train_briefly(m, batch, steps, lr=2e-3):
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    for i in range(steps):
        opt.zero_grad()
        out = m(model_batch(batch))
        out["loss"].backward()
        opt.step()
    return float(out["loss"])


def capture_train(m, batch):
    """G1 + G2: forward intermediates and every parameter gradient."""
    m.train()
    m.zero_grad()
    b = model_batch(batch)
    inputs = {k: v for k, v in b.items() if k[:5] == "input"}
    emb_in = m._embed_input(inputs)
    emb_out = m._embed_output(b["output_value"][:, :-1])
    memory = m.encoder(emb_in, src_key_padding_mask=b["input_mask"])
    tgt_mask = m._generate_square_subsequent_mask(emb_out.size(1))
    hid = m.decoder(emb_out, memory, tgt_mask=tgt_mask, tgt_key_padding_mask=b["output_mask"],
                    memory_key_padding_mask=b["input_mask"])
    dists = m._create_dist(hid)
    out = m(b)
    out["loss"].backward()
    res = {
        "g1::memory": memory.detach().numpy(), "g1::hiddens": hid.detach().numpy(),
        "g1::dists": dists.detach().numpy(), "g1::loss": np.float32(out["loss"].item()),
        "g1::accuracy": np.float32(float(out["accuracy"])),
    }
    for n, p in m.named_parameters():
        res["g2::" + n] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
    return res


def capture_adam(m, batch, lr=1e-4):
    """G3: one torch.optim.Adam(lr) step from the captured gradients (fresh state)."""
    import copy
    m2 = copy.deepcopy(m)
    m2.train()
    opt = torch.optim.Adam(m2.parameters(), lr=lr)
    opt.zero_grad()
    m2(model_batch(batch))["loss"].backward()
    opt.step()
    return {"g3::" + k: v.detach().numpy().copy() for k, v in m2.state_dict().items()}


def capture_eval(m, batch):
    m.eval()
    with torch.no_grad():
        out = m(model_batch(batch))
    res = {"g4::samples": out["samples"].numpy(), "g4::attach": out["attach"].numpy()}
    for i, pr in enumerate(out["predicts"]):
        res[f"g4::predict{i}"] = pr.numpy()
    for i, gt in enumerate(out["groundtruths"]):
        res[f"g4::groundtruth{i}"] = gt.numpy()
    return res


def capture_create_dist(m, d):
    g = torch.Generator().manual_seed(7)
    res = {}
    h = torch.randn(2, 36, d, generator=g) * 2.0
    res["g6::hiddens"] = h.numpy()
    m.train()
    with torch.no_grad():
        res["g6::train"] = m._create_dist(h.clone()).numpy()
        m.eval()
        for sz in (5, 6, 36):
            res[f"g6::eval{sz}"] = m._create_dist(h[:, :sz].clone()).numpy()
    return res


def main():
    torch.set_num_threads(8)
    # ---------------------------------------------------------------- small fixture
    spec = SynthSpec(65, 36, (3, 15), (2, 5), True)
    batch = synth_batch(4, spec, seed=2022)
    m = build(64, 4, 128, 2, 2, 65, 36)
    final = train_briefly(m, batch, 300)
    print("small fixture: loss after brief training", final)
    out = {}
    out.update(np_state(m)); out.update(np_batch(batch))
    out.update(capture_train(m, batch))
    out.update(capture_adam(m, batch))
    out.update(capture_eval(m, batch))
    out.update(capture_create_dist(m, 64))
    print("  samples[0]", out["g4::samples"][0][:20], "attach[0]", out["g4::attach"][0][:20])
    print("  label==samples", (out["g4::samples"][:, :12] == batch["output_value"][:, :12].numpy()).mean())
    np.savez_compressed(os.path.join(HERE, "fixture_small.npz"), **out)

    # ---------------------------------------------------------------- ragged / sideface
    spec = SynthSpec(65, 36, (0, 15), (2, 5), False)
    rb = synth_batch(6, spec, seed=7)
    # force one "empty" row: [END, PAD, ...]
    for k in rb:
        if k.startswith("input") and k != "input_mask":
            rb[k][0] = 0
    rb["input_value"][0] = 513
    rb["input_value"][0, 0] = 512
    rb["input_mask"][0] = rb["input_value"][0] == 513
    m2 = build(64, 4, 128, 2, 2, 65, 36)
    train_briefly(m2, rb, 200)
    out = {}
    out.update(np_state(m2)); out.update(np_batch(rb))
    cap = capture_train(m2, rb)
    for k in ("g1::loss", "g1::accuracy", "g1::hiddens", "g2::input_embeddings.input_value.weight",
              "g2::input_embeddings.input_type.weight", "g2::input_embeddings.input_view.weight",
              "g2::decoder.layers.1.multihead_attn.in_proj_weight"):
        out[k] = cap[k]
    out.update(capture_eval(m2, rb))
    np.savez_compressed(os.path.join(HERE, "fixture_ragged.npz"), **out)

    # ---------------------------------------------------------------- tiny BASELINE config 0
    spec = SynthSpec(1200, 128, (8, 299), (2, 21), True)
    tb = synth_batch(4, spec, seed=2022)
    m3 = build(128, 8, 256, 2, 2, 1200, 128)
    sd0 = np_state(m3)
    m3.train()
    opt = torch.optim.Adam(m3.parameters(), lr=1e-4)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        o = m3(model_batch(tb))
        o["loss"].backward()
        opt.step()
        losses.append(o["loss"].item())
    print("tiny losses", losses)
    out = {"g8::losses": np.array(losses, dtype=np.float32), "g8::seed": np.int64(2022)}
    # the state dict is 3.4 MB in fp32; store fp16-rounded? no: parity needs exact weights,
    # so the test re-creates them from a seeded generator instead (see build_tiny_state()).
    np.savez_compressed(os.path.join(HERE, "fixture_tiny.npz"), **out, **sd0)

    # ---------------------------------------------------------------- pointer mask
    pm = PlankModel(64, 4, 128, 0.0, "relu", True, 1, 1, 3, 2, 4, 6, 65, 128, 514, TOKEN)
    np.savez_compressed(os.path.join(HERE, "pointer_mask.npz"),
                        mask=pm._generate_pointer_mask(128).numpy().astype(np.uint8))

    # ---------------------------------------------------------------- matcher (G9)
    from third_party.matcher import HungarianMatcher
    mt = HungarianMatcher(0.5)
    rng = np.random.default_rng(5)
    cases = {}
    def rand_boxes(n):
        lo = rng.integers(0, 400, size=(n, 3))
        ext = rng.integers(1, 112, size=(n, 3))
        return np.concatenate([lo, lo + ext], 1).astype(np.int64)
    sets = []
    gt = rand_boxes(7)
    pred = gt.copy(); pred[:, 3:] += rng.integers(-6, 7, size=(7, 3)); pred = pred[rng.permutation(7)]
    sets.append((pred, gt))
    sets.append((rand_boxes(5), rand_boxes(9)))
    sets.append((np.zeros((0, 6), dtype=np.int64), rand_boxes(3)))
    # IoU exactly 0.5: box [0,0,0,2,1,1] vs [0,0,0,1,1,1]
    sets.append((np.array([[0, 0, 0, 2, 1, 1]]), np.array([[0, 0, 0, 1, 1, 1]])))
    sets.append((np.concatenate([gt[:4], rand_boxes(2)]), gt))
    for i, (pb, gb) in enumerate(sets):
        pr, rc, f1 = mt(torch.as_tensor(pb), torch.as_tensor(gb))
        cases[f"pred{i}"] = pb; cases[f"gt{i}"] = gb
        cases[f"prf{i}"] = np.array([float(pr), float(rc), float(f1)], dtype=np.float64)
        print("matcher case", i, cases[f"prf{i}"])
    cases["n"] = np.int64(len(sets))
    np.savez_compressed(os.path.join(HERE, "matcher.npz"), **cases)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
