"""The packed (variable-length) encoder self-attention of the training step, a few launches (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plankassembly_amd import ops
B, S, D, H = 16, 1024, 512, 8
g = torch.Generator(device="cuda").manual_seed(1)
rng = np.random.default_rng(2022)
lens = [4 * int(rng.integers(8, 256)) + 1 for _ in range(B)]
cu, order = ops.pack_lengths(lens, "cuda") if hasattr(ops, "pack_lengths") else (None, None)
n = int(cu[-1])
qkv = torch.randn(n, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
do = torch.randn(n, D, device="cuda", generator=g).to(torch.bfloat16)
kw = dict(drop_p=0.2, drop_seed=1)
for _ in range(5):
    o, lse = ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, order=order, **kw)
    ops.attn_varlen_bwd(do, q, k, v, o, lse, H, cu, cu, B, S, S, order=order, **kw)
torch.cuda.synchronize()
print("ok", n, lens)
