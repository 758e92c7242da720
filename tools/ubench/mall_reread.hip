// Micro-benchmark, round 4: does the 256 MB Infinity Cache keep a streamed buffer so that a SECOND pass over it runs above the HBM
// rate?  (Question behind it: during the latency-bound launches of a greedy-decode step HBM idles - could they prefetch the next
// cross-attention's K / V?)  Pass 1 reads X MB (plain or non-temporal loads), pass 2 reads the same X MB again; reported: GB/s of
// both passes for X = 16 .. 1024 MB.   hipcc --offload-arch=gfx950 -O3 mall_reread.hip -o mall_reread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void rd(const u32x4* p, long long n, uint32_t* out) {
    uint32_t acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <bool NT> float pass(const u32x4* p, long long n, uint32_t* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(rd<NT>, dim3(2048), dim3(256), 0, 0, p, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main() {
    const size_t maxb = 2048ull << 20;
    char* buf; uint32_t* out;
    hipMalloc(&buf, maxb); hipMalloc(&out, 64);
    hipMemset(buf, 1, maxb);
    printf("   X MB   pass1 plain  pass2 plain  |  pass1 nt  pass2 nt  |  pass1 nt then pass2 plain   (GB/s)\n");
    for (int mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024}) {
        const long long n = ((long long)mb << 20) / 16;
        const u32x4* flush = (const u32x4*)(buf + (1024ull << 20));
        auto gbs = [&](float ms) { return mb / 1024.0 / (ms * 1e-3); };
        pass<false>(flush, (1024ll << 20) / 16, out);                        // evict
        float a1 = pass<false>((const u32x4*)buf, n, out), a2 = pass<false>((const u32x4*)buf, n, out);
        pass<false>(flush, (1024ll << 20) / 16, out);
        float b1 = pass<true>((const u32x4*)buf, n, out), b2 = pass<true>((const u32x4*)buf, n, out);
        pass<false>(flush, (1024ll << 20) / 16, out);
        float c1 = pass<true>((const u32x4*)buf, n, out), c2 = pass<false>((const u32x4*)buf, n, out);
        printf("%7d   %10.0f  %10.0f   | %9.0f %9.0f  | %9.0f %9.0f\n", mb, gbs(a1), gbs(a2), gbs(b1), gbs(b2), gbs(c1), gbs(c2));
    }
    return 0;
}
