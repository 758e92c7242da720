"""Run one GEMM shape a few times (for PMC profiling).  python tools/gemm_one.py M N K [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
bias = torch.zeros(N, device="cuda")
for _ in range(iters):
    y = ops.gemm(x, w, bias=bias)
torch.cuda.synchronize()
print(float(y.float().abs().mean()))
