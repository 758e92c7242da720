// Micro-benchmark, round 3: what does a launch cost before its blocks do anything, and how long is an s_memtime tick?
//   (a) back-to-back period of dependent launches in one stream (HIP events over N launches): empty 1-block kernel, empty
//       256 x 256-thread kernel, 256 blocks that each read 64 KB and write 64 KB (a "minimal useful" round trip), the same at
//       512 and 2 048 blocks;
//   (b) the same kernels' begin -> end as seen from inside: wall_clock64() (100 MHz constant) of the first block's first
//       instruction and of the last block's last instruction;
//   (c) s_memtime ticks per wall_clock64 tick while an MFMA loop runs on every SIMD (what the block timelines are stamped in).
//   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }

template <int per_thread>
__global__ __launch_bounds__(256) void roundtrip_kernel(const uint4* __restrict__ in, uint4* __restrict__ out,
                                                        unsigned long long* stamps) {
    unsigned long long t0 = wall_clock64();
    const size_t base = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
    uint4 v[per_thread];
#pragma unroll
    for (int i = 0; i < per_thread; ++i) v[i] = in[base + (size_t)i * 256];
#pragma unroll
    for (int i = 0; i < per_thread; ++i) { v[i].x += 1; out[base + (size_t)i * 256] = v[i]; }
    if (stamps) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long t1 = wall_clock64();
        if (threadIdx.x == 0) { atomicMin(&stamps[0], t0); atomicMax(&stamps[1], t1); }
    }
}

__global__ __launch_bounds__(256, 1) void tick_kernel(unsigned long long* out, int iters, float s) {
    f32x16 acc = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(s + i); b[i] = (__bf16)(s - i); }
    unsigned long long w0 = wall_clock64();
    unsigned long long m0 = __builtin_readcyclecounter();
    unsigned long long c0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    unsigned long long c1 = clock64();
    unsigned long long m1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    float r = 0; for (int i = 0; i < 16; ++i) r += acc[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 - w0; out[1] = m1 - m0; out[2] = c1 - c0; out[3] = (unsigned long long)r; }
}

template <typename F> static float period_us(F launch, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms * 1e3f / n;
}

int main() {
    const size_t bytes = (size_t)2048 * 64 * 1024;
    uint4 *in, *out; unsigned long long* st;
    HC(hipMalloc(&in, bytes)); HC(hipMalloc(&out, bytes)); HC(hipMalloc(&st, 64));
    HC(hipMemset(in, 1, bytes));
    const int N = 2000;
    printf("back-to-back period of dependent launches in one stream (us per launch, %d launches)\n", N);
    printf("  empty, 1 block x 64 threads          %7.2f\n", period_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0, nullptr); }, N));
    printf("  empty, 256 blocks x 256 threads      %7.2f\n", period_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, nullptr); }, N));
    printf("  empty, 2048 blocks x 256 threads     %7.2f\n", period_us([&] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, nullptr); }, N));
    {
        hipStream_t cs; HC(hipStreamCreate(&cs));
        hipGraph_t g; hipGraphExec_t ge;
        HC(hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, cs, nullptr);
        HC(hipStreamEndCapture(cs, &g));
        HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        HC(hipGraphLaunch(ge, cs)); HC(hipStreamSynchronize(cs));
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        HC(hipEventRecord(e0, cs)); HC(hipGraphLaunch(ge, cs)); HC(hipEventRecord(e1, cs)); HC(hipEventSynchronize(e1));
        float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1));
        printf("  empty, 256 x 256, 1000 launches replayed from a hipGraph (GPU-side floor)   %7.2f\n", ms);
    }
    const int grids[4] = {256, 512, 1024, 2048};
    const int pts[3] = {1, 4, 16};
    for (int pt : pts)
        for (int g : grids) {
            auto fn = pt == 1 ? roundtrip_kernel<1> : (pt == 4 ? roundtrip_kernel<4> : roundtrip_kernel<16>);
            float us = period_us([&] { hipLaunchKernelGGL(fn, dim3(g), dim3(256), 0, 0, in, out, (unsigned long long*)nullptr); }, N);
            unsigned long long init[2] = {~0ull, 0};
            HC(hipMemcpy(st, init, 16, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fn, dim3(g), dim3(256), 0, 0, in, out, st);
            HC(hipDeviceSynchronize());
            unsigned long long got[2]; HC(hipMemcpy(got, st, 16, hipMemcpyDeviceToHost));
            const double mb = (double)g * 256 * pt * 16 * 2 / 1e6;
            printf("  read+write %5.1f KB per block, %4d blocks (%6.1f MB moved)  period %7.2f us   first instruction -> last store retired %6.2f us\n",
                   256.0 * pt * 16 / 1024, g, mb, us, (double)(got[1] - got[0]) / 100.0);
        }
    unsigned long long* tk; HC(hipMalloc(&tk, 64));
    for (int blocks : {1, 256, 1024}) {
        hipLaunchKernelGGL(tick_kernel, dim3(blocks), dim3(256), 0, 0, tk, 4000, 1.0f);
        HC(hipDeviceSynchronize());
        unsigned long long h[4]; HC(hipMemcpy(h, tk, 32, hipMemcpyDeviceToHost));
        const double us = h[0] / 100.0;
        printf("tick calibration, %4d blocks of 4 waves x 64 000 MFMA 32x32x16: %.1f us wall; s_memtime %.0f MHz, clock64 %.0f MHz; %.2f s_memtime ticks per MFMA\n",
               blocks, us, h[1] / us, h[2] / us, (double)h[1] / 64000.0);
    }
    return 0;
}
