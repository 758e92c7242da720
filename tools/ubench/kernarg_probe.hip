// Does a large by-value kernel argument cost an extra __amd_rocclr_copyBuffer dispatch?  Run under rocprofv3 --kernel-trace.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int N> struct Big { int v[N]; };
template <int N> __global__ void kern(Big<N> b, int* out) { if (threadIdx.x == 0) out[0] = b.v[N - 1]; }
template <int N> void run(int* out) { Big<N> b; for (int i = 0; i < N; ++i) b.v[i] = i; for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern<N>, dim3(1), dim3(64), 0, 0, b, out); }
int main() { int* out; (void)hipMalloc(&out, 4); run<32>(out); run<64>(out); run<128>(out); run<256>(out); run<512>(out); run<900>(out); (void)hipDeviceSynchronize(); printf("done\n"); return 0; }
