"""Forward Linears of the encoder at the step's row counts (bias + dropout + residual epilogues as the step launches them), timed with
HIP events over back-to-back pa_gemm calls on PREBUILT argument blocks (ctypes call ~3 us: the GPU stays the bottleneck).
Kernel family by environment (read once per process): default / PA_GEMM_BIG=1 / PA_GEMM_BIG=2 [PA_GEMM_BIG_TN=..]."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import _lib as L
lib = L.lib()
REPS = 200


def args(M, N, K, relu=False, drop=0.2, res=True):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    g = L.GemmArgs()
    g.A, g.B, g.C, g.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
    g.R = r.data_ptr() if r is not None else None
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
    g.batch, g.a_kcontig, g.b_kcontig, g.in_dtype, g.out_dtype = 1, 1, 1, L.PA_BF16, L.PA_BF16
    g.alpha, g.relu, g.aux_scale, g.drop_p, g.drop_seed, g.splitk = 1.0, int(relu), 1.0, drop, 7, 1
    return g, (x, w, bias, out, r)


def time_it(g):
    st = L.stream()
    for _ in range(10):
        L.check(lib.pa_gemm(C.byref(g), st), "pa_gemm")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        lib.pa_gemm(C.byref(g), st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


print("PA_GEMM_BIG", os.environ.get("PA_GEMM_BIG"), "PA_GEMM_BIG_TN", os.environ.get("PA_GEMM_BIG_TN"), flush=True)
print(f"{'rows':>6s} | " + " | ".join(f"{n:>4d}x{k:<4d} {tag:9s}" for n, k, tag in [(1536, 512, 'in_proj'), (1024, 512, 'ffn1+relu'), (512, 512, 'out_proj'), (512, 1024, 'ffn2')]))
for M in (7188, 7924, 8192, 8704, 9736):
    row = []
    for N, K, relu, res in [(1536, 512, False, False), (1024, 512, True, False), (512, 512, False, True), (512, 1024, False, True)]:
        g, keep = args(M, N, K, relu=relu, drop=0.0 if N == 1536 else 0.2, res=res)
        us = time_it(g)
        row.append(f"{us:7.1f} us {2.0 * M * N * K / us / 1e6:5.0f} TF")
        del keep
    print(f"{M:6d} | " + " | ".join(row), flush=True)
