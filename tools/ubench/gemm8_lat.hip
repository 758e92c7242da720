// Stand-alone build of the eight-wave big-tile GEMM (plankassembly_amd/csrc/gemm8.h): correctness spot check against a CPU
// reference on random data + back-to-back launch latency.  Compiles in ~30 s (gemm.hip takes minutes).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../plankassembly_amd/csrc gemm8_lat.hip -o gemm8_lat
//   gemm8_lat M N K [bk 32|64] [res 0|1] [relu 0|1] [batch] [cfg]     (-DPA_GEMM8_TRACE: per-block cycle stamps; -DPA_G8_ABL=<bits>)
#include "../../plankassembly_amd/csrc/gemm_common.h"
namespace {
#include "../../plankassembly_amd/csrc/gemm8.h"
}
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>

static int g_cfg = -1;      // -1: the product's choice (256 x 256 from N = 1024 on, else 256 x 128); >= 0: probe configurations
template <int WM, int WN, int FM, int FN, int BK, int NSTG, int WPS = 1, int SLOTS = 256>
static int launch_cfg(GemmP pb, hipStream_t st) {
    constexpr int TBM = 32 * FM * WM, TBN = 32 * FN * WN;
    pb.tiles_m = (pb.M + TBM - 1) / TBM; pb.tiles_n = (pb.N + TBN - 1) / TBN;
    pb.plain_order = pb.tiles_m < 8;
    pb.tiles_m_pad = pb.plain_order ? pb.tiles_m : (pb.tiles_m + 7) / 8 * 8;
    pb.units = pb.tiles_m_pad * pb.tiles_n * pb.batch;
    // the XCD interleave pads the row tiles to a multiple of 8; when that padding alone pushes a launch past one round of 256
    // blocks, hand every XCD a contiguous eighth of the tile list instead (gemm8.h, plain_order 2)
    const int tiles = pb.tiles_m * pb.tiles_n, per = (tiles + 7) / 8;
    if (!pb.plain_order && pb.units > SLOTS && 8 * per * pb.batch <= SLOTS) { pb.plain_order = 2; pb.units = 8 * per * pb.batch; }
    const int gb = pb.units < SLOTS ? pb.units : SLOTS;
    hipLaunchKernelGGL((gemm8_kernel<WM, WN, FM, FN, BK, NSTG, WPS>), dim3(gb), dim3(64 * WM * WN), 0, st, pb);
    return (int)hipGetLastError();
}
static int launch8(GemmP pb, int bk, hipStream_t st) {
    switch (g_cfg) {
        case 0: return launch_cfg<2, 4, 4, 2, 64, 2>(pb, st);      // A: 256 x 256, wave 128 x 64
        case 1: return launch_cfg<4, 2, 2, 2, 64, 2>(pb, st);      // B: 256 x 128, wave 64 x 64
        case 2: return launch_cfg<2, 4, 2, 2, 64, 2>(pb, st);      // C: 128 x 256, wave 64 x 64
        case 3: return launch_cfg<4, 2, 2, 4, 64, 2>(pb, st);      // E: 256 x 256, wave 64 x 128
        case 4: return launch_cfg<4, 2, 2, 3, 64, 2>(pb, st);      // F: 256 x 192, wave 64 x 96
        case 5: return launch_cfg<2, 4, 3, 2, 64, 2>(pb, st);      // G: 192 x 256, wave 96 x 64
        // four waves (one per SIMD) stacked along M, whole-width wave tiles: one round of <= 256 tiles at M ~ 8 700
        case 6: return launch_cfg<4, 1, 2, 7, 64, 2>(pb, st);      // H: 256 x 224, wave 64 x 224 (N = 1536: 34 x 7 = 238 tiles)
        case 7: return launch_cfg<4, 1, 2, 5, 64, 2>(pb, st);      // I: 256 x 160, wave 64 x 160 (N = 1024: 34 x 7 = 238 tiles)
        case 8: return launch_cfg<4, 1, 2, 4, 64, 2>(pb, st);      // J: 256 x 128, wave 64 x 128
        // four waves, TWO blocks per CU (512 slots), taller tiles so that 8 704 rows x N 1024 / 1536 stay within one round
        case 9: return launch_cfg<2, 2, 3, 2, 32, 2, 2, 512>(pb, st);     // K: 192 x 128, wave 96 x 64, BK 32
        case 10: return launch_cfg<2, 2, 4, 2, 32, 2, 2, 512>(pb, st);    // L: 256 x 128, wave 128 x 64, BK 32
        case 11: return launch_cfg<2, 2, 2, 2, 32, 3, 2, 512>(pb, st);    // M: 128 x 128, wave 64 x 64, BK 32 (the pair kernel's shape)
        case 12: return launch_cfg<2, 2, 2, 2, 64, 2, 2, 512>(pb, st);    // N: 128 x 128, BK 64, 2 stages (64 KB + 16 KB)
        default: break;
    }
    const bool sq = pb.N >= 1024;
    if (sq) return bk == 64 ? launch_cfg<4, 2, 2, 4, 64, 2>(pb, st) : launch_cfg<2, 4, 4, 2, 32, 4>(pb, st);
    return bk == 64 ? launch_cfg<4, 2, 2, 2, 64, 2>(pb, st) : launch_cfg<4, 2, 2, 2, 32, 5>(pb, st);
}

int main(int argc, char** argv) {
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
    const int bk = argc > 4 ? atoi(argv[4]) : 32, res = argc > 5 ? atoi(argv[5]) : 0, relu = argc > 6 ? atoi(argv[6]) : 0;
    const int batch = argc > 7 ? atoi(argv[7]) : 1;
    g_cfg = argc > 8 ? atoi(argv[8]) : -1;
    void *A, *B, *C, *R; float* bias;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&B, (size_t)batch * N * K * 2); (void)hipMalloc(&C, (size_t)batch * M * N * 2);
    (void)hipMalloc(&R, (size_t)M * N * 2); (void)hipMalloc(&bias, (size_t)batch * N * 4);
    std::vector<uint16_t> ha((size_t)M * K), hb((size_t)batch * N * K), hr((size_t)M * N), hc((size_t)batch * M * N);
    std::vector<float> hbias((size_t)batch * N);
    uint32_t r = 12345u;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; float f = ((r >> 9) & 0xffff) / 65536.0f - 0.5f; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
    auto f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; };
    for (auto& x : ha) x = rnd();
    for (auto& x : hb) x = rnd();
    for (auto& x : hr) x = rnd();
    for (auto& x : hbias) x = f(rnd());
    (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(B, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(R, hr.data(), hr.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(C, 0xff, (size_t)batch * M * N * 2);
    GemmP p; memset((void*)&p, 0, sizeof(p));
    p.A = A; p.B = B; p.C = C; p.bias = bias; p.R = res ? R : nullptr;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = batch * N; p.ldr = N;
    p.batch = batch; p.sA = 0; p.sB = (long long)N * K; p.sC = N; p.sBias = N;       // batch members side by side in C (as the cross K|V launch)
    p.alpha = 1.f; p.relu = relu; p.aux_scale = 1.f; p.drop_scale = 1.f; p.out_dtype = PA_BF16; p.splitk = 1; p.tiles_per_slice = K / 64;
    p.vec_ok = 1;
    int rc = launch8(p, bk, 0);
    hipError_t he = hipDeviceSynchronize();
    (void)hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
    double maxerr = 0; long bad = 0, checked = 0;
    for (int b = 0; b < batch; ++b)
        for (int m = 0; m < M; m += (M > 512 ? 61 : 1)) for (int n = 0; n < N; n += (N > 256 ? 7 : 1)) {
            double acc = 0; for (int k = 0; k < K; ++k) acc += (double)f(ha[(size_t)m * K + k]) * f(hb[((size_t)b * N + n) * K + k]);
            acc += hbias[(size_t)b * N + n];
            if (relu && acc < 0) acc = 0;
            if (res) acc += f(hr[(size_t)m * N + n]);
            const double got = f(hc[(size_t)m * batch * N + (size_t)b * N + n]);
            double e = fabs(acc - got); if (e > maxerr) maxerr = e; if (!(e <= 0.02 + 0.01 * fabs(acc))) ++bad; ++checked;
        }
    // the last rows / columns exhaustively (edge tiles)
    for (int m = (M > 40 ? M - 40 : 0); m < M; ++m) for (int n = 0; n < N; ++n) {
        double acc = 0; for (int k = 0; k < K; ++k) acc += (double)f(ha[(size_t)m * K + k]) * f(hb[(size_t)n * K + k]);
        acc += hbias[n]; if (relu && acc < 0) acc = 0; if (res) acc += f(hr[(size_t)m * N + n]);
        const double got = f(hc[(size_t)m * batch * N + n]);
        double e = fabs(acc - got); if (e > maxerr) maxerr = e; if (!(e <= 0.02 + 0.01 * fabs(acc))) ++bad; ++checked;
    }
    printf("check rc %d sync %d: %ld of %ld bad, max abs err %.4f\n", rc, (int)he, bad, checked, maxerr);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) rc |= launch8(p, bk, 0);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("gemm8 cfg %d bk%d %5d %5d %5d batch %d res %d  rc %d  %.2f us/launch  %.0f TF\n", g_cfg, bk, M, N, K, batch, res, rc, ms * 1e3 / 50,
                             2.0 * batch * M * N * K / (ms * 1e-3 / 50) / 1e12);
    }
#ifdef PA_GEMM8_TRACE
    {
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> tr(512 * 8);
        (void)hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(pa_gemm8_trace), tr.size() * 8);
        int nb = 256;
        double d[4] = {0, 0, 0, 0}; int cnt = 0;
        double tmin = 1e30, tmax = 0, xs0[8], xs1[8]; for (int i = 0; i < 8; ++i) { xs0[i] = 1e30; xs1[i] = 0; }
        for (int b = 0; b < nb; ++b) { const unsigned long long* x = &tr[b * 8]; if (!x[4]) continue; ++cnt; for (int i = 0; i < 4; ++i) d[i] += (double)(x[i + 1] - x[i]);
            const double tot = (double)(x[4] - x[0]); if (tot < tmin) tmin = tot; if (tot > tmax) tmax = tot;
            if ((double)x[0] < xs0[b & 7]) xs0[b & 7] = (double)x[0]; if ((double)x[4] > xs1[b & 7]) xs1[b & 7] = (double)x[4]; }
        if (cnt) { printf("  block total min %.0f max %.0f; per XCD first entry -> last exit:", tmin, tmax); for (int i = 0; i < 8; ++i) printf(" %.0f", xs1[i] - xs0[i]); printf("\n"); }
        if (cnt) printf("  blocks %d: setup %.0f  first item %.0f  K loop %.0f (%.0f / item)  epilogue %.0f  (cycles of the last unit of each block)\n", cnt, d[0] / cnt, d[1] / cnt,
                        d[2] / cnt, d[2] / cnt / (K / bk), d[3] / cnt);
    }
#endif
    return bad ? 1 : 0;
}
