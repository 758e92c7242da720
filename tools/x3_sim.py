"""CPU simulation of the bf16x3 ("split") matrix products: every matmul of the oracle's train step (Linears, QK^T, PV, heads) is
replaced by hi*hi + hi*lo + lo*hi of the operands' bf16 hi / lo parts (exact products, float64 accumulation = an upper bound on
what f32 accumulation inside the MFMA can add is NOT included: only the split error), forward and backward, and loss / memory /
hiddens / every gradient are compared with the float64 oracle under the f32 gate of tests/test_headline_gpu.py
(1e-4; 1e-5 + 1e-4 * scale).  Usage: python tools/x3_sim.py [case] [attn=1|0]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
import torch
import large_cases as LC
from oracle import plank_oracle as O


def split(x):
    hi = x.to(torch.bfloat16).to(x.dtype)
    lo = (x - hi).to(torch.bfloat16).to(x.dtype)
    return hi, lo


class MM3(torch.autograd.Function):
    """c = a @ b with every product (forward and both gradients) as the three-term bf16 split."""
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return MM3.mm(a, b)

    @staticmethod
    def mm(a, b):
        ah, al = split(a.float()); bh, bl = split(b.float())
        ah, al, bh, bl = ah.double(), al.double(), bh.double(), bl.double()
        return (torch.matmul(ah, bh) + torch.matmul(ah, bl) + torch.matmul(al, bh)).float()

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = MM3.mm(g, b.transpose(-1, -2))
        gb = MM3.mm(a.transpose(-1, -2), g)
        while gb.dim() > b.dim():
            gb = gb.sum(0)
        if gb.shape != b.shape:          # broadcast batch dims of b
            for i, (s0, s1) in enumerate(zip(gb.shape, b.shape)):
                if s1 == 1 and s0 != 1:
                    gb = gb.sum(i, keepdim=True)
        return ga, gb


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "headline"
    attn = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
    torch.set_num_threads(8)
    c = LC.CASES[name]
    sd, batch = LC.case_state_dict(c), LC.case_batch(c)
    cfg = LC.case_oracle_cfg(c)
    # split evaluation: patch the oracle's matmuls
    orig_linear = O.linear
    O.linear = lambda x, w, b=None: (MM3.apply(x, w.t()) if b is None else MM3.apply(x, w.t()) + b)
    if attn:
        orig_mm = torch.Tensor.__matmul__
        torch.Tensor.__matmul__ = lambda a, b: MM3.apply(a, b)
    try:
        p = {k: v.detach().clone().float().requires_grad_(True) for k, v in sd.items()}
        gate = {}

        def rec(key, x):
            gate[key] = (x > 0).detach()
            return x * gate[key]
        o = O.train_forward(p, cfg, batch, return_all=True, relu=rec)
        o["loss"].backward()
    finally:
        O.linear = orig_linear
        if attn:
            torch.Tensor.__matmul__ = orig_mm
    valid = ~batch["input_mask"]
    # float64 reference on the split run's ReLU branches where float64's own pre-activation is within tau of zero
    taus = [float(t) for t in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["2e-5"])]
    for tau in taus:
        st = dict(flips=0, far=0.0)

        def forced(key, x, tau=tau, st=st):
            natural = x > 0
            near = x.detach().abs() <= tau
            if key.startswith("encoder."):
                near = near & valid[:, :, None]
            f = torch.where(near, gate[key], natural)
            st["flips"] += int((f != natural).sum())
            dis = (gate[key] != natural)
            if key.startswith("encoder."):
                dis = dis & valid[:, :, None]
            if bool(dis.any()):
                st["far"] = max(st["far"], float(x.detach().abs()[dis].max()))
            return x * f
        p64 = {k: v.detach().clone().double().requires_grad_(True) for k, v in sd.items()}
        torch.set_default_dtype(torch.float64)
        r = O.train_forward(p64, cfg, batch, return_all=True, relu=forced)
        r["loss"].backward()
        torch.set_default_dtype(torch.float32)
        report(name, attn, o, r, p, p64, valid, tau, st)


def report(name, attn, o, r, p, p64, valid, tau, st):
    print(f"--- tau {tau:g}: {st['flips']} branches taken from the split run; largest |x_f64| among ALL disagreeing branches {st['far']:.2e}")
    print(f"[{name}] attn split {attn}: loss {float(o['loss']):.7f} vs {float(r['loss']):.7f}  diff {abs(float(o['loss']) - float(r['loss'])):.2e} (gate 1e-4)")
    print(f"    memory max err {float((o['memory'].double() - r['memory'])[valid].abs().max()):.2e}  hiddens {float((o['hiddens'].double() - r['hiddens']).abs().max()):.2e} (gate 1e-4)")
    worst, nfail = ("", 0.0), 0
    for k in p:
        g, g64 = p[k].grad, p64[k].grad
        if g is None:
            continue
        err, scale = float((g.double() - g64).abs().max()), float(g64.abs().max())
        ratio = err / (1e-5 + 1e-4 * scale)
        nfail += ratio > 1
        if ratio > worst[1]:
            worst = (k, ratio)
    print(f"    gradients: worst {worst[1]:.3f} x the bound ({worst[0]}); {nfail} tensors beyond it", flush=True)


main()
