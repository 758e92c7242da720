"""pa_gemm_ln against pa_gemm + pa_layernorm_fwd at the step's shapes (HIP events, 50 launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (256, 1120, 2048, 4096, 7940, 8192, 8704):
    for K in (512, 1024):
        for drop in (0.0, 0.2):
            x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = (torch.randn(512, K, device="cuda") * 0.05).to(torch.bfloat16)
            b, g, be = torch.randn(512, device="cuda"), torch.ones(512, device="cuda"), torch.zeros(512, device="cuda")
            r = torch.randn(M, 512, device="cuda").to(torch.bfloat16)
            def sep():
                z = ops.gemm(x, w, bias=b, residual=r, drop_p=drop, drop_seed=1)
                ops.layernorm_fwd(z, g, be, 1e-5)
            tz = t_us(lambda: ops.gemm(x, w, bias=b, residual=r, drop_p=drop, drop_seed=1))
            ts = t_us(sep)
            tf = t_us(lambda: ops.gemm_ln(x, w, g, be, 1e-5, bias=b, residual=r, drop_p=drop, drop_seed=1))
            print(f"M {M:5d} K {K:5d} drop {drop:.1f}: gemm {tz:6.1f} us, gemm + LN {ts:6.1f} us, fused {tf:6.1f} us", flush=True)
