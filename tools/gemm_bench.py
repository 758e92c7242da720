"""Micro-benchmark of pa_gemm at the model's shapes (HIP-event timed).  python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops

def t(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters

def rnd(*s): return torch.randn(*s, device="cuda").to(torch.bfloat16)

rows = []
for (M, N, K, tag) in [(16384, 1536, 512, "enc in_proj"), (16384, 512, 512, "enc out_proj"), (16384, 1024, 512, "ffn1"),
                       (16384, 512, 1024, "ffn2"), (2048, 1536, 512, "dec in_proj"), (2048, 512, 512, "dec out_proj"),
                       (16384, 1024, 512, "dec cross kv")]:
    x, w, dy = rnd(M, K), rnd(N, K), rnd(M, N)
    bias = torch.zeros(N, device="cuda")
    fl = 2.0 * M * N * K
    a = t(lambda: ops.gemm(x, w, bias=bias))
    b = t(lambda: ops.gemm(dy, w, b_kcontig=False))
    c = t(lambda: ops.gemm(dy, x, a_kcontig=False, b_kcontig=False, out_dtype=torch.float32, splitk=max(1, min(16, 512 // ((N // 128) * (K // 128))))))
    rows.append((tag, M, N, K, a * 1e6, fl / a / 1e12, b * 1e6, fl / b / 1e12, c * 1e6, fl / c / 1e12))
print(f"env NOGLDS={os.environ.get('PA_GEMM_NOGLDS')} GRID={os.environ.get('PA_GEMM_GRID')}")
print(f"{'shape':14s} {'M':>6s} {'N':>5s} {'K':>5s} | {'fwd us':>8s} {'TF':>6s} | {'dX us':>8s} {'TF':>6s} | {'dW us':>8s} {'TF':>6s}")
for r in rows:
    print(f"{r[0]:14s} {r[1]:6d} {r[2]:5d} {r[3]:5d} | {r[4]:8.1f} {r[5]:6.0f} | {r[6]:8.1f} {r[7]:6.0f} | {r[8]:8.1f} {r[9]:6.0f}")
# empty-launch overhead reference
z = torch.zeros(1, device="cuda")
print("python+launch floor (torch add):", t(lambda: z.add_(1)) * 1e6, "us")
