"""Pin the CPU oracle at the BENCHMARKED shapes (d_model 512, dh 64, 6+6 layers; S = 1024 / 999 / 299) and on a
live-loss small fixture against golden vectors of the real reference model (tests/golden/make_golden_large.py).
The weights are re-created from the seed on both sides (tests/seeded.py).  CPU only."""
import numpy as np
import pytest
import torch

import large_cases as LC
from oracle import plank_oracle as O


def _run_train(c):
    sd = LC.case_state_dict(c)
    batch = LC.case_batch(c)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = O.train_forward(p, LC.case_oracle_cfg(c), batch, return_all=True)
    out["loss"].backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    return sd, batch, out, grads


@pytest.mark.parametrize("name", ["headline", "complete", "visible", "sideface", "live", "eps0", "gelu", "t1024"])
def test_oracle_float64_matches_the_reference_module_in_float64(name):
    """The GPU gate compares the f32 HIP path with the oracle evaluated in float64 (tests/test_headline_gpu.py oracle_f64).
    That evaluation is pinned here against the REAL reference module run in float64 (fixture entries g64::*): in double
    precision the two computations agree to ~1e-12, i.e. the oracle is the reference's function, not merely close to it in f32."""
    c = LC.CASES[name]
    g = LC.load_large(name)
    torch.set_num_threads(8)
    sd = LC.case_state_dict(c)
    batch = LC.case_batch(c)
    p = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    torch.set_default_dtype(torch.float64)
    try:
        out = O.train_forward(p, LC.case_oracle_cfg(c), batch)
        out["loss"].backward()
    finally:
        torch.set_default_dtype(torch.float32)
    assert abs(float(out["loss"].detach()) - float(g["g64::loss"])) <= 1e-10 * max(1.0, abs(float(g["g64::loss"])))
    for k, v in p.items():
        gr = v.grad if v.grad is not None else torch.zeros_like(v)
        g2 = gr.reshape(gr.shape[0], -1) if gr.dim() > 1 else gr.reshape(1, -1)
        scale = float(g["g64::gmax::" + k])
        err = float(np.abs(g2[:LC.SLICE[0], :LC.SLICE[1]].numpy() - g["g64::gslice::" + k]).max())
        assert err <= 1e-12 + 1e-8 * scale, (k, err, scale)
        n_ref = float(g["g64::gnorm::" + k])
        assert abs(float(gr.norm()) - n_ref) <= 1e-12 + 1e-8 * n_ref, (k, float(gr.norm()), n_ref)


@pytest.mark.parametrize("name", ["headline", "complete", "visible", "sideface", "live", "eps0", "gelu", "t1024"])
def test_oracle_train_matches_reference_at_large_shapes(name):
    c = LC.CASES[name]
    g = LC.load_large(name)
    torch.set_num_threads(8)
    sd, batch, out, grads = _run_train(c)
    assert abs(float(out["loss"].detach()) - float(g["g::loss"])) <= 2e-5 * max(1.0, abs(float(g["g::loss"])))
    assert abs(out["accuracy"] - float(g["g::accuracy"])) < 1e-7
    valid = ~batch["input_mask"]
    mem = out["memory"].detach()
    ref_norm = torch.from_numpy(g["g::memory_norm"])
    assert float((mem.norm(dim=-1) - ref_norm)[valid].abs().max()) <= 2e-4 * float(ref_norm[valid].max())
    rows = torch.arange(0, mem.shape[1], 37)[:24]
    got = mem[:, rows, :LC.SLICE[1]]
    assert float((got - torch.from_numpy(g["g::memory_slice"]))[valid[:, rows]].abs().max()) < 1e-4
    hid = out["hiddens"].detach()
    assert float((hid[:, :, :64] - torch.from_numpy(g["g::hiddens_slice"])).abs().max()) < 1e-4
    got = LC.grad_summary(grads)
    for k in grads:
        scale = float(g["g::gmax::" + k])
        err = float(np.abs(got["gslice::" + k] - g["g::gslice::" + k]).max())
        assert err <= 1e-5 + 1e-4 * scale, (k, err, scale)
        n_ref = float(g["g::gnorm::" + k])
        assert abs(float(got["gnorm::" + k]) - n_ref) <= 1e-6 + 1e-4 * n_ref, (k, got["gnorm::" + k], n_ref)
    if c.get("all_grads"):
        for k, v in grads.items():
            ref = torch.from_numpy(g["gfull::" + k])
            assert float((v - ref).abs().max()) <= 1e-6 + 1e-4 * float(ref.abs().max()), k
        assert float(g["g::loss"]) > 1.0, "the live fixture must carry a real loss"


def test_oracle_greedy_decode_matches_reference_at_headline_shape():
    c = LC.CASES["headline"]
    g = LC.load_large("headline")
    torch.set_num_threads(8)
    sd = LC.case_state_dict(c)
    db = LC.case_batch(c, decode=True)
    with torch.no_grad():
        s, a, marg = O.greedy_decode_cached(sd, LC.case_oracle_cfg(c), db, return_margins=True)
    assert np.array_equal(s.numpy(), g["d::samples"]) and np.array_equal(a.numpy(), g["d::attach"])
    assert s.shape == (c["decode_b"], c["max_out"]) and int((a >= 0).sum()) > 0
    assert float(marg.min()) > 1e-3, "fixture chosen so that no argmax is a near-tie"


def test_oracle_greedy_decode_1024_steps_matches_reference():
    """BASELINE configs[4] length: MAX_OUTPUT_LENGTH 1024, every step run (END suppressed), against the tokens of the
    reference's own O(T^2) loop (tests/golden/fixture_t1024.npz: 2 x 1024 tokens, 119 pointer copies, 81 of them past
    step 128)."""
    c = LC.CASES["t1024"]
    g = LC.load_large("t1024")
    torch.set_num_threads(8)
    sd = LC.case_state_dict(c)
    db = LC.case_batch(c, decode=True)
    with torch.no_grad():
        s, a = O.greedy_decode_cached(sd, LC.case_oracle_cfg(c), db)
    assert s.shape == (2, 1024) and not bool((s == 512).any())
    assert np.array_equal(s.numpy(), g["d::samples"]) and np.array_equal(a.numpy(), g["d::attach"])
    assert int((g["d::attach"][:, 128:] >= 0).sum()) >= 50 and float(g["d::margins"].min()) > 1e-4
