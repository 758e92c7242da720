"""Packed encoder self-attention of the training step with and without range blocks (pa_attn_args.ws; csrc/attention.hip
decode_unit_split): forward and backward (dQ + dK/dV launches), HIP events over REPS launches, the SAME kernels both ways.
PA_ATTN_SPLIT_KMAX / PA_ATTN_SPLIT_PMAX are read once per process: run once per setting (tools/r06_attn_split.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plankassembly_amd import ops
D, H, REPS = 512, 8, 40


def time_pair(lens, ws, drop):
    B, S = len(lens), max(lens)
    cu, order = ops.pack_lengths(lens, "cuda")
    n = int(cu[-1])
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(n, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    do = torch.randn(n, D, device="cuda", generator=g).to(torch.bfloat16)
    kw = dict(drop_p=drop, drop_seed=1, order=order, ws=ws)
    o, lse = ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, **kw)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, **kw)
        ops.attn_varlen_bwd(do, q, k, v, o, lse, H, cu, cu, B, S, S, **kw)
    ev[0].record()
    for _ in range(REPS):
        ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, **kw)
    ev[1].record()
    for _ in range(REPS):
        ops.attn_varlen_bwd(do, q, k, v, o, lse, H, cu, cu, B, S, S, **kw)
    ev[2].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / REPS * 1e3, ev[1].elapsed_time(ev[2]) / REPS * 1e3, n


def run(name, lens, drop=0.2):
    B, S = len(lens), max(lens)
    n = sum(lens)
    ws = ops.attn_split_ws(n, B, H, "cuda", L_max=S)
    f0, b0, _ = time_pair(lens, None, drop)
    f1, b1, _ = time_pair(lens, ws, drop) if ws is not None else (float("nan"), float("nan"), 0)
    fl = 4.0 * sum(l * l for l in lens) * D
    print(f"{name:30s} rows {n:6d}  fwd {f0:6.1f} -> {f1:6.1f} us ({fl / f0 / 1e6:5.0f} -> {fl / f1 / 1e6:5.0f} TF)   "
          f"bwd {b0:6.1f} -> {b1:6.1f} us ({2.5 * fl / b0 / 1e6:5.0f} -> {2.5 * fl / b1 / 1e6:5.0f} TF)", flush=True)


print("PA_ATTN_SPLIT_KMAX", os.environ.get("PA_ATTN_SPLIT_KMAX", "8"), "PA_ATTN_SPLIT_PMAX", os.environ.get("PA_ATTN_SPLIT_PMAX", "2"), flush=True)
for seed in (2022, 7, 11):
    rng = np.random.default_rng(seed)
    lens = [4 * int(rng.integers(8, 256)) + 1 for _ in range(16)]
    run(f"headline mixed seed {seed}", lens)
    if seed == 2022:
        run("  no dropout", lens, drop=0.0)
run("16 x 1021", [1021] * 16)
run("16 x 1199 (complete max)", [1199] * 16)
run("8 x 1021 + 24 x 200", [1021] * 8 + [200] * 24)
