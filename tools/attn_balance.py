"""Where does the packed (variable-length) self-attention lose against the padded benchmark shape?  Times the forward
and the backward pair for several length sets with the SAME kernels: the training batch's mixed lengths, equal lengths
of the same total work, full-length rows, and many short rows.  HIP events over REPS launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plankassembly_amd import ops
D, H, REPS = 512, 8, 30


def run(name, lens, drop=0.2):
    B, S = len(lens), max(lens)
    cu, order = ops.pack_lengths(lens, "cuda")
    n = int(cu[-1])
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(n, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    do = torch.randn(n, D, device="cuda", generator=g).to(torch.bfloat16)
    kw = dict(drop_p=drop, drop_seed=1)
    o, lse = ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, order=order, **kw)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, order=order, **kw)
        ops.attn_varlen_bwd(do, q, k, v, o, lse, H, cu, cu, B, S, S, order=order, **kw)
    ev[0].record()
    for _ in range(REPS):
        ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, order=order, **kw)
    ev[1].record()
    for _ in range(REPS):
        ops.attn_varlen_bwd(do, q, k, v, o, lse, H, cu, cu, B, S, S, order=order, **kw)
    ev[2].record()
    torch.cuda.synchronize()
    tf = ev[0].elapsed_time(ev[1]) / REPS * 1e3
    tb = ev[1].elapsed_time(ev[2]) / REPS * 1e3
    fl = 4.0 * sum(l * l for l in lens) * D
    print(f"{name:34s} rows {n:6d}  fwd {tf:7.1f} us {fl / tf / 1e6:6.0f} TF   bwd {tb:7.1f} us {2.5 * fl / tb / 1e6:6.0f} TF"
          f"   lens {sorted(lens)[:3]}..{sorted(lens)[-3:]}", flush=True)


for seed in (2022, 7, 11):
    rng = np.random.default_rng(seed)
    lens = [4 * int(rng.integers(8, 256)) + 1 for _ in range(16)]
    run(f"headline mixed seed {seed}", lens)
    eq = int(round((sum(l * l for l in lens) / 16) ** 0.5))
    run(f"  equal lengths, same work ({eq})", [eq] * 16)
    run(f"  no dropout, mixed", lens, drop=0.0)
run("16 x 1021", [1021] * 16)
run("16 x 1024", [1024] * 16)
run("32 x 512", [512] * 32)
run("64 x 256", [256] * 64)
run("64 x 299 (sideface)", [299] * 64)
run("128 x 128", [128] * 128)
