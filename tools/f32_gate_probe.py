"""Diagnostic for the f32 gate (tests/test_headline_gpu.py): per case, every gradient tensor of the f32 HIP step whose distance from
the float64 oracle exceeds 1e-5 + 1e-4 * scale, next to the f32 oracle's own distance.  python tools/f32_gate_probe.py [cases...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import large_cases as LC
import test_headline_gpu as TH
from oracle import plank_oracle as O

def probe(name, c, batch):
    sd = LC.case_state_dict(c)
    ref, r64 = TH.oracle_f64(c, sd, batch)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    O.train_forward(p, LC.case_oracle_cfg(c), batch)["loss"].backward()
    r32 = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()}
    m = TH.hip_model(c, "f32", sd)
    out, mem, hid, grads = TH.run_hip_train(m, batch)
    fb = TH.ForcedBranches(m, batch)
    _, rfb = TH.oracle_f64(c, sd, batch, relu=fb)
    over = [(k, float((g.double() - rfb[k]).abs().max()), float(rfb[k].abs().max())) for k, g in grads.items()]
    over = [(k, e, s_) for k, e, s_ in over if e > 1e-5 + 1e-4 * s_]
    worst = max(float((g.double() - rfb[k]).abs().max()) / max(float(rfb[k].abs().max()), 1e-6) for k, g in grads.items())
    print(f"== {name}: loss hip {out['loss'].item():.7f} f64 {float(ref['loss']):.7f}; on the device's ReLU branches ({fb.flips} ties): "
          f"worst relative error {worst:.2e}, beyond the bound: {over}")
    n_over = 0
    for k, g in grads.items():
        r = r64[k]
        scale = float(r.abs().max())
        e_hip = float((g.double() - r).abs().max()); e_ref = float((r32[k].double() - r).abs().max())
        if e_hip > 1e-5 + 1e-4 * scale or e_ref > 1e-5 + 1e-4 * scale:
            n_over += 1
            print(f"   {k:55s} scale {scale:.3e}  hip {e_hip:.2e} ({e_hip / max(scale, 1e-12):.2e} rel)  torch-f32 {e_ref:.2e} ({e_ref / max(scale, 1e-12):.2e} rel)")
    print(f"   -> {n_over} of {len(grads)} tensors beyond the bound for either implementation")

names = sys.argv[1:] or ["headline", "complete", "visible", "sideface", "t1024", "b16-above"]
for n in names:
    if n.startswith("b16-"):
        c, batch, _ = TH._b16_case(n[4:])
    elif n == "sideface64":
        c = LC.CASES["sideface"]; batch = LC.case_batch(c, batch_size=64)
    else:
        c = LC.CASES[n]; batch = LC.case_batch(c)
    probe(n, c, batch)
