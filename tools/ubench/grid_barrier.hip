// Micro-benchmark, round 4: what does an in-kernel grid barrier cost on gfx950 when the stages it separates exchange DATA through
// global memory across the 8 XCDs (each XCD has its own L2)?  The decode step's 43 tiny dependent launches (5 us each) could be
// stages of a few persistent "chain" kernels if a barrier + visibility is much cheaper than a launch.
//   stage i: every block writes 4 KB (values derived from i and the block index), barrier, every block reads the 4 KB of the
//   block (b + 37 * (i + 1)) % grid - written on another XCD - and checks the values (a stale line = an error counted).
// Variants:  fence = 0: atomics only (no data guarantees; lower bound)
//   fence = 1: every thread: agent-scope release before arriving, acquire after leaving (buffer_wbl2 sc1 / buffer_inv sc1)
//   fence = 2: the data itself is written and read with agent-scope accesses (global_store / global_load ... sc1: write-through,
//              re-validated on read) - no cache-wide write-back / invalidate at all
//   fence = 3: as 1, but only wave 0 of a block executes the cache-wide operations (they act on the CU's L1 / the XCD's L2, not
//              on a wave), the other waves only drain their stores before the block barrier
//   fence = 4: as 2 with a two-level arrival: blocks of one XCD (blockIdx % 8) count on their own line, the last one of each XCD
//              counts on the global line (8 instead of `grid` atomics on one address)
//   hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// monotone arrival counter: stage s is complete when the counter reaches (s + 1) * gridDim.x
template <int FENCE>
__device__ __forceinline__ void grid_sync(unsigned* bar, unsigned stage, unsigned grid) {
    const unsigned target = (stage + 1) * grid;
    if (FENCE == 1) __threadfence();                // release: this thread's stores are visible at agent scope
    if (FENCE == 2 || FENCE == 3 || FENCE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (FENCE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (FENCE == 4) {
            // blocks of XCD x: bar[64 + 64 x] (own 256-byte line); the last of them adds the XCD's whole count to bar[0]
            const unsigned x = blockIdx.x & 7, mine = (grid - x + 7) / 8;
            const unsigned old = __hip_atomic_fetch_add(bar + 64 + 64 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (stage + 1) * mine) __hip_atomic_fetch_add(bar, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        if (FENCE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (FENCE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // acquire: later loads see the other blocks' stores
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int FENCE> __device__ __forceinline__ void put(uint4* p, uint4 v) {
    if (FENCE == 2 || FENCE == 4) { u32x4 w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory"); }
    else *p = v;
}
template <int FENCE> __device__ __forceinline__ uint4 get(const uint4* p) {
    if (FENCE == 2 || FENCE == 4) {
        u32x4 w; asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
        uint4 v; v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3]; return v;
    }
    return *p;
}

template <int FENCE>
__global__ __launch_bounds__(256, 2) void stages_kernel(uint4* buf, unsigned* bar, int stages, unsigned* errors,
                                                        unsigned long long* stamps) {
    const unsigned long long t0 = wall_clock64();
    const int g = gridDim.x, b = blockIdx.x, t = threadIdx.x;
    unsigned bad = 0;
    for (int s = 0; s < stages; ++s) {
        uint4* mine = buf + ((size_t)(s & 1) * g + b) * 256;
        uint4 v; v.x = s * 1000003u + b; v.y = t; v.z = s; v.w = b ^ t;
        put<FENCE>(mine + t, v);
        grid_sync<FENCE>(bar, (unsigned)s, (unsigned)g);
        const int o = (b + 37 * (s + 1)) % g;
        const uint4 r = get<FENCE>(buf + ((size_t)(s & 1) * g + o) * 256 + t);
        bad += (r.x != s * 1000003u + o) | (r.y != (unsigned)t) | (r.z != (unsigned)s) | (r.w != (unsigned)(o ^ t));
    }
    if (bad) atomicAdd(errors, bad);
    if (t == 0) { atomicMin(&stamps[0], t0); atomicMax(&stamps[1], wall_clock64()); }
}

template <int FENCE> static int run(int grid, int stages, uint4* buf, unsigned* bar, unsigned* err, unsigned long long* stamps) {
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float best = 1e30f; unsigned long long inner = 0; unsigned nerr = 0;
    for (int rep = 0; rep < 5; ++rep) {
        HC(hipMemset(bar, 0, 4096)); HC(hipMemset(err, 0, 4));
        unsigned long long init[2] = {~0ull, 0ull};
        HC(hipMemcpy(stamps, init, 16, hipMemcpyHostToDevice));
        HC(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stages_kernel<FENCE>, dim3(grid), dim3(256), 0, 0, buf, bar, stages, err, stamps);
        HC(hipEventRecord(e1, 0));
        HC(hipDeviceSynchronize());
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        unsigned e; HC(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        unsigned long long st[2]; HC(hipMemcpy(st, stamps, 16, hipMemcpyDeviceToHost));
        if (ms < best) { best = ms; inner = st[1] - st[0]; }
        nerr += e;
    }
    printf("  fence %d  grid %4d x 256 threads, %4d stages: %7.2f us per stage (events)  %7.2f us per stage (inside the kernel)   stale reads: %u\n",
           FENCE, grid, stages, best * 1e3f / stages, inner * 0.01 / stages, nerr);
    return 0;
}

int main() {
    uint4* buf; unsigned *bar, *err; unsigned long long* stamps;
    HC(hipMalloc(&buf, (size_t)2 * 1024 * 256 * 16)); HC(hipMalloc(&bar, 4096)); HC(hipMalloc(&err, 256)); HC(hipMalloc(&stamps, 256));
    printf("stage = write 4 KB per block, grid barrier, read another block's 4 KB (us per stage)\n");
    const int grids[] = {128, 256, 384, 512};
    for (int g : grids) { if (run<0>(g, 1000, buf, bar, err, stamps)) return 1; }
    for (int g : grids) { if (run<1>(g, 1000, buf, bar, err, stamps)) return 1; }
    for (int g : grids) { if (run<3>(g, 1000, buf, bar, err, stamps)) return 1; }
    for (int g : grids) { if (run<2>(g, 1000, buf, bar, err, stamps)) return 1; }
    for (int g : grids) { if (run<4>(g, 1000, buf, bar, err, stamps)) return 1; }
    return 0;
}
