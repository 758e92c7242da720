"""HIP-event timing of the LayerNorm kernels at the train step's two row counts (packed encoder rows, decoder rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops

def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

d = 512
for rows in (2048, 8000, 9700, 16384):
    z = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
    dy = torch.randn(rows, d, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(z, g, b, 1.0)
    dg, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
    tf = t(lambda: ops.layernorm_fwd(z, g, b, 1.0))
    tb = t(lambda: ops.layernorm_bwd(dy, z, g, mean, rstd, dg, db))
    tbd = t(lambda: ops.layernorm_bwd(dy, z, g, mean, rstd, dg, db, drop_p=0.2, drop_seed=5))
    mb = rows * d * 2 / 1e6
    print(f"rows {rows:6d}: fwd {tf:6.2f} us ({2 * mb / tf / 1e3:5.2f} TB/s)  bwd {tb:6.2f} us ({3 * mb / tb / 1e3:5.2f} TB/s)  "
          f"bwd+dropout {tbd:6.2f} us ({4 * mb / tbd / 1e3:5.2f} TB/s)")
