import os, sys, torch
sys.path.insert(0, "/root/repo")
from plankassembly_amd import ops
def t(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
Bm, M, N, K = 6, 8704, 1024, 512
a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(Bm, N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(Bm, N, device="cuda")
out = torch.empty(M, Bm * N, dtype=torch.bfloat16, device="cuda")
o3 = out.view(M, Bm, N).permute(1, 0, 2)
a3 = a[None].expand(Bm, M, K)
print("PA_GEMM_BIG", os.environ.get("PA_GEMM_BIG"), "cross-KV batched 6 x 8704x1024x512:", round(t(lambda: ops.gemm(a3, w, bias=bias, out=o3)), 1), "us")
for (M2, N2, K2) in [(8704, 1536, 512), (8704, 1024, 512), (10240, 1536, 512), (7168, 1536, 512)]:
    x = torch.randn(M2, K2, device="cuda").bfloat16(); w2 = (torch.randn(N2, K2, device="cuda") * 0.05).bfloat16(); b2 = torch.randn(N2, device="cuda")
    print("   ", M2, N2, K2, round(t(lambda: ops.gemm(x, w2, bias=b2)), 1), "us")
