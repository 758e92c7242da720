// Micro-benchmark: per-CU load-path throughput for L2-resident tiles, GEMM-like access (rows of 128 B at a 1 KiB
// stride), via global_load_lds (DMA) or global_load_dwordx4 -> registers -> ds_write, with/without a barrier
// per tile, at 1..4 blocks per CU.   hipcc --offload-arch=gfx950 -O3 ldpath.hip -o ldpath
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BARRIER>
__global__ __launch_bounds__(256) void k(const char* base, size_t region, int iters, int ld, float* out) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // each block owns a 256 KiB window (128 rows x 2 KiB... emulate A tile rows at stride ld bytes)
    const char* win = base + ((size_t)blockIdx.x * 262144) % region;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int k0 = (it & 7) * 128;                       // 8 K tiles of 128 B
        char* buf = smem + (it & 1) * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                        // 2048 chunks of 16 B = 32 KiB per tile
            const int p = tid + i * 256;
            const int row = p >> 3, ch = p & 7;
            const char* src = win + (size_t)(row & 127) * ld + k0 + ch * 16 + (row >> 7) * 131072;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                    (__attribute__((address_space(3))) void*)(buf + (i * 256 + wave * 64) * 16), 16, 0, 0);
            } else {
                u32x4 v = *reinterpret_cast<const u32x4*>(src);
                *reinterpret_cast<u32x4*>(buf + p * 16) = v;
            }
        }
        if (BARRIER) __syncthreads();
        else __builtin_amdgcn_s_waitcnt(0);
        acc += *reinterpret_cast<u32x4*>(buf + tid * 16);
    }
    if (acc[0] == 0x12345) out[0] = 1.f;
}

int main() {
    const size_t region = 64u << 20;
    char* d; float* o;
    hipMalloc(&d, region + (1 << 20)); hipMalloc(&o, 4);
    hipMemset(d, 1, region + (1 << 20));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 400;
    for (int mode = 0; mode < 2; ++mode)
        for (int barrier = 0; barrier < 2; ++barrier)
            for (int bpc = 1; bpc <= 4; ++bpc)
                for (size_t reg : {(size_t)(2u << 20), (size_t)(16u << 20), region}) {
                    int shm = bpc == 1 ? 160 * 1024 : (bpc == 2 ? 80 * 1024 : (bpc == 3 ? 53 * 1024 : 40 * 1024));
                    shm = shm / 256 * 256;
                    auto fn = mode == 0 ? (barrier ? k<0, 1> : k<0, 0>) : (barrier ? k<1, 1> : k<1, 0>);
                    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
                    int grid = 256 * bpc;
                    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), shm, 0, d, reg, 20, 1024, o);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), shm, 0, d, reg, iters, 1024, o);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    double bytes = (double)grid * iters * 32768;
                    double tbs = bytes / (ms * 1e-3) / 1e12;
                    printf("mode=%s barrier=%d blocks/CU=%d region=%3zuMB : %6.2f TB/s  = %5.1f B/clk/CU @2.4GHz  (%.1f us/tile/block)\n",
                           mode == 0 ? "glds" : "regs", barrier, bpc, reg >> 20, tbs, tbs * 1e12 / 256 / 2.4e9, ms * 1e3 / iters);
                }
    return 0;
}
