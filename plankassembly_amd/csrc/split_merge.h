// Included inside the anonymous namespace of attention.hip and decode.hip (after pa_device.h).
//
// Hand-over of partial results between the blocks that share one output tile ("range blocks": each walks a range of the streamed
// side; the LAST one to arrive merges the others' partials and stores).  No block ever waits for another: correct under any
// dispatch order, placement or residency (cdna_hip_programming.md Guideline 16).
#pragma once
#ifdef PA_SPLIT_FENCE
constexpr bool SPLIT_FENCE = true;       // probe build: acquire fence + sc1 loads
#else
constexpr bool SPLIT_FENCE = false;
#endif
// this range block's tiles [lo, hi) of the element's n streamed tiles
__device__ __forceinline__ void split_range(int n, int part, int nparts, int& lo, int& hi) {
    const int per = (n + nparts - 1) / nparts;
    lo = min(n, part * per); hi = min(n, lo + per);
}
// Partial results travel through HBM/L2 between blocks that may sit on different XCDs (private, mutually incoherent L2s):
// cdna_hip_programming.md Guideline 16, form R1 - payload stored WRITE-THROUGH (16-byte sc1 buffer stores), every storing wave
// drains them (s_waitcnt vmcnt(0)), the block meets, ONE lane takes a ticket (relaxed agent-scope atomic); the block that
// draws the last ticket meets again and reads the others' partials with sc1 LOADS (split_get; no acquire fence).
// Tickets are zero between launches: the last arriver puts the word back (the scratch is zeroed once by its owner).
struct SplitOut {
    __amdgpu_buffer_rsrc_t rs;
    __device__ __forceinline__ SplitOut(char* base, int bytes) : rs(__builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000)) {}
    // vector j of this thread ([j][thread] layout: a wave's store covers 1 KB)
    __device__ __forceinline__ void put(int j, int nthreads, f32x4 v) const {
        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), rs, (j * nthreads + (int)threadIdx.x) * 16, 0, 16);   // aux 16 = sc1
    }
};
// (sc1 loads - L1 bypassed, served from the memory side like the sc1 stores that wrote the data: valid WITHOUT an acquire fence
//  when the producer stored sc1, cdna_hip_programming.md Guideline 16; the fence - buffer_inv sc1 - was measured first and made
//  every range-block launch slower than the unsplit one: profiles/r06_attention_launch_shape.txt)
__device__ __forceinline__ f32x4 split_get(const char* base, int j, int nthreads, int bytes) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, bytes, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (j * nthreads + (int)threadIdx.x) * 16, 0, 16);
    return *reinterpret_cast<const f32x4*>(&v);
}
__device__ __forceinline__ bool split_arrive(int* tick, int nparts, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(tick, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == nparts - 1;
        if (last) {
            if (SPLIT_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(tick, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *s_flag = last;
    }
    __syncthreads();
    return *s_flag != 0;
}


