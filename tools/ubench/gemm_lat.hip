// Back-to-back launch latency of pa_gemm for small shapes (no instrumentation).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../plankassembly_amd/csrc gemm_lat.hip -o gemm_lat
// Usage: gemm_lat M N K [a_kc b_kc splitk res]      (res 1: bf16 residual added in the epilogue)
#include "../../plankassembly_amd/csrc/gemm.hip"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
int main(int argc, char** argv) {
    int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    int akc = argc > 4 ? atoi(argv[4]) : 1, bkc = argc > 5 ? atoi(argv[5]) : 1, sk = argc > 6 ? atoi(argv[6]) : 1;
    const int res = argc > 7 ? atoi(argv[7]) : 0;
    void *A, *B, *C, *ws; float* bias;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&B, (size_t)N * K * 2); (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&ws, (size_t)sk * M * N * 4);
    (void)hipMemset(A, 0x3c, (size_t)M * K * 2); (void)hipMemset(B, 0x3c, (size_t)N * K * 2); (void)hipMemset(bias, 0, N * 4);
    pa_gemm_args g; memset((void*)&g, 0, sizeof(g));
    g.A = A; g.B = B; g.C = C; g.bias = sk > 1 ? nullptr : bias; g.ws = ws; g.M = M; g.N = N; g.K = K;
    g.lda = akc ? K : M; g.ldb = bkc ? K : N; g.ldc = N; g.batch = 1;
    void* Rb = nullptr;
    if (res && sk == 1) { (void)hipMalloc(&Rb, (size_t)M * N * 2); (void)hipMemset(Rb, 0, (size_t)M * N * 2); g.R = Rb; g.ldr = N; }
    g.a_kcontig = akc; g.b_kcontig = bkc; g.in_dtype = PA_BF16; g.out_dtype = sk > 1 ? PA_F32 : PA_BF16; g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = sk;
    // correctness spot check on small problems (random data, CPU reference)
    if ((long long)M * N * K <= (1LL << 28) && akc && bkc && sk == 1) {
        std::vector<uint16_t> ha((size_t)M * K), hb((size_t)N * K), hc((size_t)M * N);
        uint32_t r = 12345u;
        auto rnd = [&]() { r = r * 1664525u + 1013904223u; float f = ((r >> 9) & 0xffff) / 65536.0f - 0.5f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
        for (auto& x : ha) x = rnd();
        for (auto& x : hb) x = rnd();
        (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        int rc = pa_gemm(&g, 0); (void)hipDeviceSynchronize();
        (void)hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
        auto f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; };
        double maxerr = 0; int bad = 0;
        for (int m = 0; m < M; m += (M > 256 ? 37 : 1)) for (int n = 0; n < N; n += (N > 256 ? 11 : 1)) {
            double acc = 0; for (int k = 0; k < K; ++k) acc += (double)f(ha[(size_t)m * K + k]) * f(hb[(size_t)n * K + k]);
            double e = fabs(acc - f(hc[(size_t)m * N + n])); if (e > maxerr) maxerr = e; if (e > 0.05 + 0.02 * fabs(acc)) ++bad;
        }
        printf("check rc %d max abs err %.4f bad %d\n", rc, maxerr, bad);
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        int rc = 0;
        for (int i = 0; i < 50; ++i) rc |= pa_gemm(&g, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%5d %5d %5d kc%d%d sk%d res%d  rc %d  %.2f us/launch  %.0f TF\n", M, N, K, akc, bkc, sk, res, rc, ms * 1e3 / 50, 2.0 * M * N * K / (ms * 1e-3 / 50) / 1e12);
    }
    return 0;
}
