// Back-to-back launch latency of pa_gemm for small shapes (no instrumentation).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../plankassembly_amd/csrc gemm_lat.hip -o gemm_lat
// Usage: gemm_lat M N K [a_kc b_kc splitk]
#include "../../plankassembly_amd/csrc/gemm.hip"
#include <stdio.h>
#include <string.h>
int main(int argc, char** argv) {
    int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    int akc = argc > 4 ? atoi(argv[4]) : 1, bkc = argc > 5 ? atoi(argv[5]) : 1, sk = argc > 6 ? atoi(argv[6]) : 1;
    void *A, *B, *C, *ws; float* bias;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&B, (size_t)N * K * 2); (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&ws, (size_t)sk * M * N * 4);
    (void)hipMemset(A, 0x3c, (size_t)M * K * 2); (void)hipMemset(B, 0x3c, (size_t)N * K * 2); (void)hipMemset(bias, 0, N * 4);
    pa_gemm_args g; memset((void*)&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = sk > 1 ? nullptr : bias; g.ws = ws; g.M = M; g.N = N; g.K = K;
    g.lda = akc ? K : M; g.ldb = bkc ? K : N; g.ldc = N; g.batch = 1;
    g.a_kcontig = akc; g.b_kcontig = bkc; g.in_dtype = PA_BF16; g.out_dtype = sk > 1 ? PA_F32 : PA_BF16; g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = sk;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        int rc = 0;
        for (int i = 0; i < 50; ++i) rc |= pa_gemm(&g, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%5d %5d %5d kc%d%d sk%d  rc %d  %.2f us/launch  %.0f TF\n", M, N, K, akc, bkc, sk, rc, ms * 1e3 / 50, 2.0 * M * N * K / (ms * 1e-3 / 50) / 1e12);
    }
    return 0;
}
