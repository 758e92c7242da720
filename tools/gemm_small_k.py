"""The decoder-side Linears of the train step (2 048 rows, 64 x 64-tile kernel `gemm3s_kernel`): time against K at fixed N, to read
the cost of one 64-deep K tile ("item") and the launch's fixed part.  HIP events over back-to-back pa_gemm calls on prebuilt
argument blocks (as tools/gemm_tiles.py).  python tools/gemm_small_k.py [rows]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from plankassembly_amd import _lib as L
lib = L.lib()
REPS = 300


def args(M, N, K, res=True, drop=0.2):
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(M, N, device="cuda").bfloat16() if res else None
    g = L.GemmArgs()
    g.A, g.B, g.C, g.bias = x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr()
    g.R = r.data_ptr() if r is not None else None
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldr = M, N, K, K, K, N, N
    g.batch, g.a_kcontig, g.b_kcontig, g.in_dtype, g.out_dtype = 1, 1, 1, L.PA_BF16, L.PA_BF16
    g.alpha, g.relu, g.aux_scale, g.drop_p, g.drop_seed, g.splitk = 1.0, 0, 1.0, drop, 7, 1
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


M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
KS = (64, 128, 256, 512, 1024, 1536, 2048)
print(f"rows {M}; us per launch (bias + dropout + residual epilogue)")
print(f"{'N':>6s} | " + " | ".join(f"K {k:5d}" for k in KS) + " | us per 64-deep tile (512 -> 1536)")
for N in (512, 1024, 1536):
    row = []
    for K in KS:
        g, keep = args(M, N, K)
        row.append(time_it(g))
        del keep
    slope = (row[5] - row[3]) / 16
    print(f"{N:6d} | " + " | ".join(f"{u:7.2f}" for u in row) + f" | {slope:6.3f}", flush=True)
