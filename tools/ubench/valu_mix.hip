// Micro-benchmark, round 3: which softmax instruction mix should the attention kernels issue?  Per MFMA (32x32x16 bf16 =
// the forward's rate of one MFMA per two scores per lane at dh = 64) a wave runs one "pair" of scores through a candidate
// mix; 1..5 waves per SIMD.  Reported: SIMD-cycles per (MFMA + mix) and the MFMA-pipe share that leaves (32 / cycles).
//   hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { M_NONE, M_CUR, M_CUR_NODROP, M_PK, M_PK_NODROP, M_SGPRMASK, M_PKMASK, M_ONLY_VALU_CUR, M_ONLY_VALU_PK, M_PKFMA_ALONE, M_FMA_ALONE,
       M_PKADD_ALONE, M_EXP_ALONE, M_CNDS_ALONE, M_N };
static const char* NAMES[M_N] = {
    "mfma only",
    "mfma + current mix (2 fma 2 exp 2 add 2 mul24 2 cmp 2 sel 1 cvt 1 max3)",
    "mfma + current, no dropout (2 fma 2 exp 2 add 1 cvt 1 max3)",
    "mfma + packed mix (1 pk_fma 2 exp 1 pk_add 2 mul24 2 cmp 2 sel 1 cvt 1 max3)",
    "mfma + packed, no dropout (1 pk_fma 2 exp 1 pk_add 1 cvt 1 max3)",
    "mfma + lane-mask dropout (2 fma 2 exp 2 add 2 cndmask(sgpr) 1 cvt 1 max3)",
    "mfma + packed + lane-mask dropout (1 pk_fma 2 exp 1 pk_add 2 cndmask(sgpr) 1 cvt 1 max3)",
    "current mix alone (no mfma)",
    "packed mix alone (no mfma)",
    "v_pk_fma_f32 alone", "v_fma_f32 alone", "v_pk_add_f32 alone", "v_exp_f32 alone", "v_cndmask(sgpr pair) alone"};

#define FMA(x, s) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s))
#define PKFMA(x, s) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s))
#define PKADD(x, y) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MUL24(d, a, b) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define CMPSEL(x, a, t) asm volatile("v_cmp_ge_u32 vcc, %1, %2\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x) : "v"(a), "v"(t) : "vcc")
#define SELS(x, m) asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(x) : "s"(m))
#define CVT(d, a, b) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define MAX3(x, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b))

template <int KIND, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, long long* cyc, int iters, float seed, unsigned long long m0) {
    f32x2 x[8], l[4];
    uint32_t u[16];
    float mx = seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i][0] = seed + threadIdx.x * 1e-3f + i; x[i][1] = seed - i; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { l[i][0] = 0.f; l[i][1] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = threadIdx.x * 977u + i;
    f32x2 s2 = {seed, seed};
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    unsigned long long lm = m0 | 1ull;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            constexpr bool WITH_MFMA = KIND <= M_PKMASK;
            if (WITH_MFMA) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            }
            f32x2& p = x[m];
            const bool packed = KIND == M_PK || KIND == M_PK_NODROP || KIND == M_PKMASK || KIND == M_ONLY_VALU_PK;
            const bool mix = KIND != M_NONE && KIND < M_PKFMA_ALONE;
            if (mix) {
                if (packed) PKFMA(p, s2); else { FMA(p[0], seed); FMA(p[1], seed); }
                EXP(p[0]); EXP(p[1]);
                if (packed) PKADD(l[m & 3], p); else { ADD(l[m & 3][0], p[0]); ADD(l[m & 3][1], p[1]); }
                if (KIND == M_CUR || KIND == M_PK || KIND == M_ONLY_VALU_CUR || KIND == M_ONLY_VALU_PK) {
                    uint32_t h0, h1;
                    MUL24(h0, u[m], u[m + 8]); MUL24(h1, u[m], u[(m + 9) & 15]);
                    CMPSEL(p[0], h0, u[15]); CMPSEL(p[1], h1, u[15]);
                }
                if (KIND == M_SGPRMASK || KIND == M_PKMASK) { SELS(p[0], lm); SELS(p[1], lm); }
                CVT(u[(m + 4) & 15], p[0], p[1]);
                MAX3(mx, p[0], p[1]);
            }
            if (KIND == M_PKFMA_ALONE) { PKFMA(x[m], s2); PKFMA(l[m & 3], s2); }
            if (KIND == M_FMA_ALONE) { FMA(x[m][0], seed); FMA(x[m][1], seed); }
            if (KIND == M_PKADD_ALONE) { PKADD(x[m], s2); PKADD(l[m & 3], s2); }
            if (KIND == M_EXP_ALONE) { EXP(x[m][0]); EXP(x[m][1]); }
            if (KIND == M_CNDS_ALONE) { SELS(x[m][0], lm); SELS(x[m][1], lm); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = mx;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += l[i][0] + l[i][1];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += (float)u[i] + acc0[i] + acc1[i];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int WPS> void run(float* o, long long* c) {
    const int iters = 1000;
    const int grid = 256 * WPS;
    hipLaunchKernelGGL((k<KIND, WPS>), dim3(grid), dim3(256), 0, 0, o, c, 10, 1.0f, 0xdeadbeefcafef00dull);
    hipLaunchKernelGGL((k<KIND, WPS>), dim3(grid), dim3(256), 0, 0, o, c, iters, 1.0f, 0xdeadbeefcafef00dull);
    hipDeviceSynchronize();
    static long long h[256 * 8 * 4];
    hipMemcpy(h, c, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < grid * 4; ++i) avg += (double)h[i];
    avg /= grid * 4;
    const double per = avg / iters / 8 / WPS;          // SIMD-cycles per group (1 MFMA and/or one 2-score mix)
    printf("  %-86s waves/SIMD %d: %7.2f SIMD-cycles per group%s\n", NAMES[KIND], WPS, per,
           KIND <= M_PKMASK ? "" : " (2 instr / group for the *alone* rows)");
}
template <int WPS> void all(float* o, long long* c) {
    run<M_NONE, WPS>(o, c); run<M_CUR, WPS>(o, c); run<M_CUR_NODROP, WPS>(o, c); run<M_PK, WPS>(o, c); run<M_PK_NODROP, WPS>(o, c);
    run<M_SGPRMASK, WPS>(o, c); run<M_PKMASK, WPS>(o, c); run<M_ONLY_VALU_CUR, WPS>(o, c); run<M_ONLY_VALU_PK, WPS>(o, c);
    run<M_PKFMA_ALONE, WPS>(o, c); run<M_FMA_ALONE, WPS>(o, c); run<M_PKADD_ALONE, WPS>(o, c); run<M_EXP_ALONE, WPS>(o, c);
    run<M_CNDS_ALONE, WPS>(o, c);
}
int main() {
    float* o; long long* c;
    hipMalloc(&o, 4); hipMalloc(&c, sizeof(long long) * 256 * 8 * 4);
    all<1>(o, c); all<2>(o, c); all<3>(o, c); all<4>(o, c);
    return 0;
}
