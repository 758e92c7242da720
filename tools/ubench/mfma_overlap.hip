// Micro-benchmark, round 4: does VALU work overlap with a v_mfma_f32_32x32x16_bf16 on the same SIMD, and does it depend on where
// the MFMA accumulator lives?  gfx950 has one 512-entry register file per lane, split into "arch" VGPRs (v0..) and accumulation
// VGPRs (a0..); hipcc puts MFMA accumulators into arch VGPRs (the _vgprcd form) whenever the kernel may not use more than 256
// registers, i.e. in every kernel that wants 2+ waves per SIMD.  Per loop iteration a wave issues one MFMA followed by N
// independent v_fma_f32 (no operand shared with the MFMA); reported: SIMD cycles per iteration (s_memtime, shader clock) for
// N = 0..16 with the accumulator (a) in arch VGPRs, (b) in AGPRs, at 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define FMA(x, s) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(s))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

template <bool AGPR, int N, int WPS, bool TRANS>
__global__ __launch_bounds__(256, WPS) void k(float* out, long long* cyc, int iters, float seed) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed + threadIdx.x * 1e-3f + i;
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (AGPR) {
                if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            } else {
                if (m & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int i = 0; i < N; ++i) { if (TRANS && (i & 3) == 0) EXP(x[i & 15]); else FMA(x[i & 15], seed); }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i] + acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <bool AGPR, int N, int WPS, bool TRANS> double run(float* out, long long* cyc, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<AGPR, N, WPS, TRANS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((k<AGPR, N, WPS, TRANS>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // cycles of one wave per iteration of (1 MFMA + N VALU); WPS waves share the SIMD, so SIMD cycles per (MFMA + N) = that / WPS
    return s / blocks / iters / 4.0 / WPS;
}

template <int WPS, bool TRANS> void sweep(float* out, long long* cyc) {
    const int blocks = 256 * WPS;
    printf("waves/SIMD %d%s   N VALU per MFMA:      0      2      4      6      8     12     16\n", WPS, TRANS ? " (every 4th a v_exp_f32)" : "");
    printf("  accumulator in arch VGPRs:     %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f   SIMD cycles per (MFMA + N VALU)\n",
           run<false, 0, WPS, TRANS>(out, cyc, blocks), run<false, 2, WPS, TRANS>(out, cyc, blocks), run<false, 4, WPS, TRANS>(out, cyc, blocks), run<false, 6, WPS, TRANS>(out, cyc, blocks),
           run<false, 8, WPS, TRANS>(out, cyc, blocks), run<false, 12, WPS, TRANS>(out, cyc, blocks), run<false, 16, WPS, TRANS>(out, cyc, blocks));
    printf("  accumulator in AGPRs:          %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f\n",
           run<true, 0, WPS, TRANS>(out, cyc, blocks), run<true, 2, WPS, TRANS>(out, cyc, blocks), run<true, 4, WPS, TRANS>(out, cyc, blocks), run<true, 6, WPS, TRANS>(out, cyc, blocks),
           run<true, 8, WPS, TRANS>(out, cyc, blocks), run<true, 12, WPS, TRANS>(out, cyc, blocks), run<true, 16, WPS, TRANS>(out, cyc, blocks));
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(float));
    hipMalloc(&cyc, 1024 * sizeof(long long));
    sweep<1, false>(out, cyc);
    sweep<2, false>(out, cyc);
    sweep<4, false>(out, cyc);
    sweep<2, true>(out, cyc);
    return 0;
}
