// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds u16 element indices; every lane supplies its own
// byte address; print what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const int* addr, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    uint32_t a = (uint32_t)(size_t)lds + addr[threadIdx.x];
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    int* da; uint16_t* dout; hipMalloc(&da, 256); hipMalloc(&dout, 512);
    int h[64]; uint16_t o[256];
    for (int test = 0; test < 3; ++test) {
        const int RS = 256;   // row stride in bytes (128 bf16 per row)
        for (int l = 0; l < 64; ++l) {
            int L = l & 15, g = l >> 4;
            if (test == 0) h[l] = l * 8;                                   // contiguous
            if (test == 1) h[l] = (L >> 2) * RS + (L & 3) * 8 + g * 32;   // 4 rows x 16 cols patch per 16-lane group, groups side by side
            if (test == 2) h[l] = (g * 4 + (L >> 2)) * RS + (L & 3) * 8;  // groups stacked along rows
        }
        hipMemcpy(da, h, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
        hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        printf("test %d (element index = row*128 + col for RS=256B)\n", test);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d addr %5d -> %5d %5d %5d %5d", l, h[l], o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
            if (test) printf("   (r,c): (%d,%d) (%d,%d) (%d,%d) (%d,%d)", o[l*4]/128, o[l*4]%128, o[l*4+1]/128, o[l*4+1]%128, o[l*4+2]/128, o[l*4+2]%128, o[l*4+3]/128, o[l*4+3]%128);
            printf("\n");
        }
    }
    return 0;
}
