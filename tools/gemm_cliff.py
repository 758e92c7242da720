"""Rows just past 8 192 make a 128x128 tiling of an N = 512 Linear 257..288 tiles: one more than the CUs.  Times the
step's Linear shapes around that edge with the default kernel choice and with the ring kernel forced (PA_GEMM_V3=2).
    python tools/gemm_cliff.py            (spawns itself once per setting)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(512, 512), (512, 1024), (512, 1536), (1024, 512), (1536, 512)]
MS = [7936, 8192, 8200, 8448, 8704, 9216, 9728]
if len(sys.argv) > 1:
    import torch
    from plankassembly_amd import ops
    for N, K in SHAPES:
        for M in MS:
            x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
            w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
            bias = torch.zeros(N, device="cuda")
            for _ in range(5):
                ops.gemm(x, w, bias=bias)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(40):
                ops.gemm(x, w, bias=bias)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 40 * 1e3
            print(f"{sys.argv[1]:8s} N {N:5d} K {K:5d} M {M:5d}  {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)
else:
    for tag, env in (("default", {}), ("ring", {"PA_GEMM_V3": "2"}), ("wide", {"PA_GEMM_WIDE": "1"}), ("small", {"PA_GEMM_SMALL_MAX": "256"}),
                     ("pair", {"PA_GEMM_V3": "0"})):
        subprocess.run([sys.executable, __file__, tag], env={**os.environ, **env}, check=False)
