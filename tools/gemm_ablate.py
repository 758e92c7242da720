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
out = []
for (M, N, K) in [(16384, 1536, 512), (2048, 512, 512), (16384, 512, 1024)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16); w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.zeros(N, device="cuda")
    out.append(f"{M}x{N}x{K}: {t(lambda: ops.gemm(x, w, bias=bias))*1e6:7.1f} us")
print(f"DBG={os.environ.get('PA_GEMM_DBG','0'):>2s} NOGLDS={os.environ.get('PA_GEMM_NOGLDS','0')} | " + " | ".join(out))
