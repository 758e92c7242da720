// Micro-benchmark, round 3: can a fused attention backward accumulate dQ with global f32 atomics?
// Access pattern of a 16-row x 64-column f32 accumulator tile in the MFMA 16x16 output layout (lane: column l % 16 of a
// 16-column group, rows 4 * (l / 16) .. + 3): 16 atomic instructions per tile per wave, each touching 4 rows x 64 B.
// Every wave adds `iters` tiles into a [rows][512] f32 matrix; consecutive iterations of a wave hit the same 16 rows and the
// next 64-column group, different waves start at different rows (as the key blocks of one (sample, head) would).
// Reported: payload GB/s for agent-scope and workgroup-scope atomics and for plain stores of the same pattern.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>   // 0 agent-scope atomic, 1 workgroup-scope atomic, 2 plain store, 3 unsafe-fp-atomics style (no return, agent)
__global__ __launch_bounds__(256) void k(float* dq, int rows, int iters, int same_xcd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // blocks of one "head" share rows: with same_xcd the 8 blocks b, b + 8, ... (one XCD) form a group, else 8 consecutive blocks
    const int grp = same_xcd ? (blockIdx.x % 8) + 8 * (blockIdx.x / 64) : blockIdx.x / 8;
    const int mem = same_xcd ? (blockIdx.x / 8) % 8 : blockIdx.x % 8;
    const int ngrp = gridDim.x / 8;
    const int rows_per_grp = rows / ngrp;                       // rows owned by a group (all 8 x 4 waves add into them)
    float v = 1.0f + lane * 1e-3f;
    for (int it = 0; it < iters; ++it) {
        const int t = it + mem * 7 + wave * 3;                   // tile walk: staggered so that members rarely collide in time
        const int r0 = grp * rows_per_grp + (t * 16) % rows_per_grp;
        const int c0 = ((t * 16) / rows_per_grp % 8) * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* p = dq + (size_t)(r0 + 4 * (lane >> 4) + i) * 512 + c0 + 16 * j + (lane & 15);
                if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else if (MODE == 2) *p = v;
                else unsafeAtomicAdd(p, v);
            }
    }
}

int main() {
    const int rows = 8192, iters = 256;
    float* dq; HC(hipMalloc(&dq, (size_t)rows * 512 * 4));
    HC(hipMemset(dq, 0, (size_t)rows * 512 * 4));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    const char* names[4] = {"atomic add f32, agent scope", "atomic add f32, workgroup scope", "plain store", "unsafeAtomicAdd (agent)"};
    for (int same = 0; same < 2; ++same)
        for (int mode = 0; mode < 4; ++mode)
            for (int grid : {256, 1024}) {
                auto launch = [&] {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, dq, rows, iters, same);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, dq, rows, iters, same);
                    else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, dq, rows, iters, same);
                    else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, dq, rows, iters, same);
                };
                launch(); HC(hipDeviceSynchronize());
                HC(hipEventRecord(e0, 0));
                for (int r = 0; r < 5; ++r) launch();
                HC(hipEventRecord(e1, 0)); HC(hipEventSynchronize(e1));
                float ms = 0; HC(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)grid * 4 * iters * 16 * 64 * 4;
                printf("%-34s  group on %-9s  %4d blocks: %8.1f us per launch, %7.1f GB/s payload (%.0f MB)\n", names[mode],
                       same ? "one XCD" : "8 XCDs", grid, ms / 5 * 1e3, bytes / (ms / 5 * 1e-3) / 1e9, bytes / 1e6);
            }
    return 0;
}
