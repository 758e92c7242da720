"""Per-block cycle stamps of the packed forward attention (debug build with -DPA_ATTN_TRACE, see csrc/attention.hip):
    PLANK_HIP_LIB=tools/ubench/libplank_trace.so python tools/attn_trace.py
prints, per length set, the mean cycles a block spends before it knows its rows (lookup chain), until the first K/V
tile has landed, per key step, and in the epilogue."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plankassembly_amd import ops, _lib as L
D, H = 512, 8


def run(name, lens, drop=0.2):
    B, S = len(lens), max(lens)
    cu, order = ops.pack_lengths(lens, "cuda")
    n = int(cu[-1])
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(n, 3 * D, device="cuda", generator=g).to(torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    kw = dict(drop_p=drop, drop_seed=1)
    for _ in range(3):
        ops.attn_varlen_fwd(q, k, v, H, cu, cu, B, S, S, order=order, **kw)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    fn = L.lib().pa_attn_trace_read
    fn.restype = C.c_int
    rc = fn(buf.ctypes.data_as(C.c_void_p), C.c_int32(buf.size))
    assert rc == 0, rc
    t = buf.reshape(-1, 8).astype(np.int64)
    t = t[t[:, 0] > 0]
    t = t[t[:, 4] > 0]
    lookup, first, loop, epi, steps = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5]
    span = t[:, 4].max() - t[:, 0].min()
    print(f"{name:28s} blocks {len(t):5d}  lookup {lookup.mean():7.0f}  first tile {first.mean():7.0f}  loop {loop.mean():8.0f} "
          f"({(loop / np.maximum(steps, 1)).mean():6.0f}/step, {steps.mean():4.1f} steps)  epilogue {epi.mean():6.0f}  "
          f"block total {(t[:, 4] - t[:, 0]).mean():8.0f}  kernel span {span}  start spread {t[:, 0].max() - t[:, 0].min()}", flush=True)


rng = np.random.default_rng(2022)
run("headline mixed", [4 * int(rng.integers(8, 256)) + 1 for _ in range(16)])
run("16 x 1024", [1024] * 16)
run("64 x 256", [256] * 64)
run("128 x 128", [128] * 128)
run("128 x 128 no dropout", [128] * 128, drop=0.0)
