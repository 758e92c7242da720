// Standalone cycle trace of the GEMM kernel (block 0, thread 0 timestamps).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPA_GEMM_TRACE -I../../plankassembly_amd/csrc gemm_trace.hip -o gemm_trace
#include "../../plankassembly_amd/csrc/gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 1536, K = argc > 3 ? atoi(argv[3]) : 512;
    void *A, *B, *C; float* bias;
    hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)M * N * 2); hipMalloc(&bias, N * 4);
    hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, N * 4);
    pa_gemm_args g; memset((void*)&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.batch = 1;
    g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = PA_BF16; g.out_dtype = PA_BF16; g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        int zero = 0; hipMemcpyToSymbol(HIP_SYMBOL(pa_trace_n), &zero, 4);
        hipEventRecord(e0);
        int rc = pa_gemm(&g, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d rc %d  %.1f us  %.0f TF\n", rep, rc, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    }
    int n; hipMemcpyFromSymbol(&n, HIP_SYMBOL(pa_trace_n), 4);
    std::vector<unsigned long long> tr(n);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(pa_trace), n * 8);
    unsigned long long prev = n ? tr[1] : 0, t0 = prev;
    for (int i = 0; i + 1 < n && i < 900; i += 2) { printf("tag %llu  +%6llu  (t=%llu)\n", tr[i], tr[i + 1] - prev, tr[i + 1] - t0); prev = tr[i + 1]; }
    return 0;
}
