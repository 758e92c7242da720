// Micro-benchmark: issue cost of the VALU instruction classes the attention softmax is made of, alone and next to
// MFMAs, at 1 / 2 / 3 waves per SIMD.  Cycles are s_memtime ticks of one wave (= shader cycles) divided by the
// number of instructions of the class it issued.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
//
// Every body is 16 independent dependency chains so that the numbers are issue rates, not latencies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { K_FMA, K_EXP, K_CVTPK, K_CMPSEL, K_MULLO, K_MAX3, K_MFMA, K_MFMA_FMA4, K_MFMA_FMA8, K_MFMA_FMA12, K_MFMA_EXP2, K_MFMA_MIX, K_N };
static const char* NAMES[K_N] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_cmp+v_cndmask", "v_mul_lo_u32", "v_max3_f32",
                                 "mfma 32x32x16 bf16", "mfma + 4 fma", "mfma + 8 fma", "mfma + 12 fma", "mfma + 2 exp + 4 fma",
                                 "mfma + softmax mix (2 fma 2 exp 2 add 1 cvt 2 cmp 2 sel)"};

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
    float x[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { x[i] = seed + threadIdx.x * 1e-3f + i; u[i] = threadIdx.x * 977u + i; }
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
        } else if (KIND == K_EXP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        } else if (KIND == K_CVTPK) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(x[i]), "v"(x[(i + 1) & 15]));
        } else if (KIND == K_CMPSEL) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_cmp_ge_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[i]) : "v"(u[i]), "v"(u[(i + 3) & 15]) : "vcc");
        } else if (KIND == K_MULLO) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 5) & 15]));
        } else if (KIND == K_MAX3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
        } else {
            // 8 MFMAs on two accumulators, each followed by F fillers
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                constexpr int NF = KIND == K_MFMA_FMA4 ? 4 : KIND == K_MFMA_FMA8 ? 8 : KIND == K_MFMA_FMA12 ? 12 : 0;
#pragma unroll
                for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(m * NF + i) & 15]) : "v"(seed));
                if (KIND == K_MFMA_EXP2) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[(2 * m) & 15]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[(2 * m + 1) & 15]));
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[(m * 4 + i + 8) & 15]) : "v"(seed));
                }
                if (KIND == K_MFMA_MIX) {
                    const int j = (2 * m) & 15, j2 = (2 * m + 1) & 15;
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j]) : "v"(seed));
                    asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[j2]) : "v"(seed));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(x[j2]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(j + 8) & 15]) : "v"(x[j]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(j2 + 8) & 15]) : "v"(x[j2]));
                    asm volatile("v_cmp_ge_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[j]) : "v"(u[j]), "v"(u[j2]) : "vcc");
                    asm volatile("v_cmp_ge_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[j2]) : "v"(u[j2]), "v"(u[j]) : "vcc");
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[(j + 4) & 15]) : "v"(x[j]), "v"(x[j2]));
                }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + (float)u[i] + acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND> void run(int wps, float* o, long long* c) {
    const int iters = 2000;
    // wps waves per SIMD: blocks of 256 threads (one wave per SIMD), wps blocks per CU
    const int grid = 256 * wps;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, o, c, 10, 1.0f);
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, o, c, iters, 1.0f);
    hipDeviceSynchronize();
    static long long h[256 * 4 * 4];
    hipMemcpy(h, c, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < grid * 4; ++i) avg += (double)h[i];
    avg /= grid * 4;
    const int per_iter = KIND >= K_MFMA ? 8 : (KIND == K_CMPSEL ? 32 : 16);
    printf("  %-58s waves/SIMD %d: %8.2f cycles per %s (wave-local), %8.2f SIMD-cycles\n", NAMES[KIND], wps, avg / iters / per_iter,
           KIND >= K_MFMA ? "MFMA group" : "instr", avg / iters / per_iter / wps);
}

int main() {
    float* o; long long* c;
    hipMalloc(&o, 4); hipMalloc(&c, sizeof(long long) * 256 * 4 * 4);
    printf("s_memtime / readcyclecounter ticks; (constant 100 MHz counter if the numbers look 24x too small)\n");
    for (int wps = 1; wps <= 3; ++wps) {
        run<K_FMA>(wps, o, c); run<K_EXP>(wps, o, c); run<K_CVTPK>(wps, o, c); run<K_CMPSEL>(wps, o, c); run<K_MULLO>(wps, o, c);
        run<K_MAX3>(wps, o, c); run<K_MFMA>(wps, o, c); run<K_MFMA_FMA4>(wps, o, c); run<K_MFMA_FMA8>(wps, o, c);
        run<K_MFMA_FMA12>(wps, o, c); run<K_MFMA_EXP2>(wps, o, c); run<K_MFMA_MIX>(wps, o, c);
    }
    return 0;
}
