"""Does a Linear pay for weights that are not in a cache?  One decoder-size and one encoder-size bf16 GEMM, timed by rocprofv3 in three
states of the WEIGHT operand (activations are rewritten by an element-wise kernel right before every launch, as in the step):
  hot    - the same weights as the launch before (L2 / Infinity Cache resident)
  fresh  - the weights were just written by a copy kernel (the state behind the bf16x3 split kernel: Infinity Cache, not this XCD's L2)
  cold   - a 1 GiB fill ran in between (weights come from HBM: the state of a train step's weights)
Run:  rocprofv3 --kernel-trace -d out -o t -- python tools/gemm_cold.py ; python tools/gemm_cold.py --report out/.../t_results.db"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import sqlite3, re
    c = sqlite3.connect(sys.argv[2])
    rows = c.execute("select name, grid_x / workgroup_x, end - start from kernels order by start").fetchall()
    # phases are separated by marker launches (a 3-element cos_)
    seq = [(re.sub(r"\(anonymous namespace\)::", "", n), g, d) for n, g, d in rows]
    out, phase = {}, -1
    for n, g, d in seq:
        if "cos" in n.lower():
            phase += 1
            continue
        if "gemm" in n and phase >= 0:
            out.setdefault(phase, []).append(d / 1e3)
    for ph in out: out[ph] = out[ph][:60]                               # (the warm-up launches of the next shape follow the last phase)
    names = [f"{shape} {state}" for shape in ("2048x512x512", "2048x1536x512", "8704x512x512", "8704x1536x512") for state in ("hot", "fresh", "cold")]
    for ph, v in sorted(out.items()):
        v = sorted(v)
        print(f"{names[ph] if ph < len(names) else ph:24s} n {len(v):4d}  median {v[len(v) // 2]:7.2f} us  min {v[0]:7.2f}  p90 {v[int(len(v) * 0.9)]:7.2f}")
    sys.exit(0)
import torch
from plankassembly_amd import ops
torch.manual_seed(0)
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB
mark = torch.empty(3, dtype=torch.float32, device="cuda")
N_IT = 60
for M, N, K in ((2048, 512, 512), (2048, 1536, 512), (8704, 512, 512), (8704, 1536, 512)):
    x0 = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    x = x0.clone()
    w0 = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    w = w0.clone()
    bias = torch.zeros(N, device="cuda")
    for _ in range(5):
        ops.gemm(x, w, bias=bias)
    torch.cuda.synchronize()
    for state in ("hot", "fresh", "cold"):
        mark.cos_()                                                      # phase marker
        for _ in range(N_IT):
            if state == "cold":
                big.fill_(0.0)
            if state == "fresh":
                w.copy_(w0)
            x.copy_(x0)                                                  # activations: always just written
            y = ops.gemm(x, w, bias=bias)
    torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
