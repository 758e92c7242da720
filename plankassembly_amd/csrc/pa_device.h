// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of the PlankAssembly hot path.
// Everything here is written for MI355X only: 64-lane waves, 32x32 MFMA tiles, 16-byte LDS vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define PA_F32 0
#define PA_BF16 1

// Ordered (run-to-run bit-identical) gradient reductions.  The f32 path is the parity path: its column sums, LayerNorm
// gamma/beta finishes and embedding-table segment sums then have ONE contributor per output element (a single block walks
// all rows in order) instead of several blocks meeting in f32 atomics.  bf16 keeps the faster multi-block form.
// PA_DETERMINISTIC=0/1 forces either for both dtypes.
#include <stdlib.h>
extern "C" int pa_gemm_split_active(void);       // gemm.hip: 1 while the bf16x3 ("split") mode is on
static inline bool pa_ordered_reductions(int dtype) {
    static const int env = getenv("PA_DETERMINISTIC") ? atoi(getenv("PA_DETERMINISTIC")) : -1;
    // exact f32 = the checker: ordered.  bf16x3 (f32 storage, split products) is a throughput mode of the parity path: the faster
    // multi-block form like bf16 (the ordered bias / LayerNorm / embedding sums cost 0.75 ms of its 14.3 ms step)
    return env < 0 ? (dtype == PA_F32 && !pa_gemm_split_active()) : env != 0;
}

// ---------------------------------------------------------------------------------------------
// element traits: EB = elements per 16-byte vector; KC = contraction elements one mma16B() covers
template <typename T> struct ET;
template <> struct ET<float> { static constexpr int EB = 4; static constexpr int KC = 8; static constexpr int DT = PA_F32; };
template <> struct ET<bf16>  { static constexpr int EB = 8; static constexpr int KC = 16; static constexpr int DT = PA_BF16; };

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 (RNE).  As a VECTOR conversion: element-wise casts were sometimes lowered to a conversion per
    // element plus a v_perm_b32, and an inline-asm v_cvt_pk is invisible to hipcc's hazard recogniser (an asm write to a
    // register an in-flight MFMA still reads as its B operand corrupted results - seen with dh = 16 / 32 tiles).
    const f32x2 v = {lo, hi};
    const bf16x2 p = __builtin_convertvector(v, bf16x2);
    return *reinterpret_cast<const uint32_t*>(&p);
}

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16>(const bf16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16>(bf16* p, float v) { *p = (bf16)v; }

// 4 consecutive elements <-> float4 (8 B for bf16, 16 B for f32); pointer must be suitably aligned
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16>(const bf16* p) {
    u32x2 u = *reinterpret_cast<const u32x2*>(p);
    f32x4 r; r[0] = bf16_lo(u[0]); r[1] = bf16_hi(u[0]); r[2] = bf16_lo(u[1]); r[3] = bf16_hi(u[1]);
    return r;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, f32x4 v) {
    u32x2 u; u[0] = pack_bf16(v[0], v[1]); u[1] = pack_bf16(v[2], v[3]);
#ifdef PA_WT_STORES
    // probe build (-DPA_WT_STORES): write-through stores (relaxed agent-scope atomic store = global_store_dwordx2 ... sc1): the row
    // leaves for memory while the kernel is still running instead of sitting dirty in L2 until the end-of-kernel write-back
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), ((unsigned long long)u[1] << 32) | u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *reinterpret_cast<u32x2*>(p) = u;
#endif
}

// bf16x3 image of four consecutive f32 values (gemm.hip "bf16x3 (split) products"; pa_gemm_split_reserve): hi = bf16(x),
// lo = bf16(x - hi) written as the three parts of a [rows][3 cols] image row - pat 0: (hi, hi, lo), pat 1: (hi, lo, hi)
__device__ __forceinline__ void split_store4(bf16* img_row, int cols, int c, f32x4 v, int pat) {
    u32x2 hi, lo;
    hi[0] = pack_bf16(v[0], v[1]); hi[1] = pack_bf16(v[2], v[3]);
    lo[0] = pack_bf16(v[0] - bf16_lo(hi[0]), v[1] - bf16_hi(hi[0]));
    lo[1] = pack_bf16(v[2] - bf16_lo(hi[1]), v[3] - bf16_hi(hi[1]));
    *reinterpret_cast<u32x2*>(img_row + c) = hi;
    *reinterpret_cast<u32x2*>(img_row + cols + c) = pat ? lo : hi;
    *reinterpret_cast<u32x2*>(img_row + 2 * cols + c) = pat ? hi : lo;
}

// ---------------------------------------------------------------------------------------------
// One "16-byte step" of a 32x32 MFMA tile.  Lane l = (i = l & 31, h = l >> 5) supplies for the A
// operand row i and for the B operand column i the EB contraction elements [EB*h, EB*h+EB) of
// the step's KC = 2*EB elements.  bf16: one v_mfma_f32_32x32x16_bf16.  f32: four
// v_mfma_f32_32x32x2_f32 (exact f32; MFMA e pairs element e of both half-waves, which is a
// consistent relabelling of the contraction index for A and B alike).
// C/D layout (dtype independent): col = l & 31, row = (r & 3) + 8*(r >> 2) + 4*(l >> 5), r in [0,16).
template <typename T> __device__ __forceinline__ void mma16B(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16B<bf16>(f32x16& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                  *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16B<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), __uint_as_float(b[e]), acc, 0, 0, 0);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// transpose an EB x EB element block held as EB 16-byte vectors (vector i = row i) in registers
template <typename T> __device__ __forceinline__ void transpose_block(const u32x4* in, u32x4* out);
template <> __device__ __forceinline__ void transpose_block<float>(const u32x4* in, u32x4* out) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        out[e][0] = in[0][e]; out[e][1] = in[1][e]; out[e][2] = in[2][e]; out[e][3] = in[3][e];
    }
}
template <> __device__ __forceinline__ void transpose_block<bf16>(const u32x4* in, u32x4* out) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t lo = in[2 * q][e >> 1], hi = in[2 * q + 1][e >> 1];
            out[e][q] = (e & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// counter-based dropout RNG: 32-bit avalanche hash (of a row / column / key index mixed with the site seed)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// Attention-probability dropout: separable counter-based decisions.  keep(row, key) <=> the low 32 bits of the 24 x 24-bit
// product A[row] * C[key] are >= thr32, with A = a 24-bit avalanche hash of the global query-row index and C = an odd
// 24-bit hash of the key index.  One v_mul_u32_u24 + one compare per score in every kernel layout: the forward / dQ
// kernels (lane = query row) hold A in a register and read C per key tile from LDS, the dK/dV kernel (lane = key) holds
// C and reads A per query tile.  thr32 = p * 2^32, so the drop probability is p to 2^-32 (no quantisation) and the
// survivors' 1/(1-p) is exact.  Statistics (keep rate, row / column variance, pair correlations, 2 x 2 parity, spectrum
// of a 256 x 256 block) checked against the binomial expectation; tests/test_kernels_gpu.py checks rate and determinism.
// Both factors are ODD and have bit 23 SET (values in [2^23, 2^24)): the 48-bit product is then at least 2^46 and its low
// 32 bits have wrapped at least 2^14 times whatever the two hashes are.  Without the forced top bit a small row hash
// (A < 52 at p = 0.2: A * C < thr32 for every key) dropped a whole attention row, and A < ~300 kept only 50-70 % of it
// (ADVICE r2); tests/test_kernels_gpu.py::test_attention_dropout_keep_rate_per_row_and_key checks the smallest hashes.
__device__ __forceinline__ uint32_t drop_row_hash(uint32_t seed, uint32_t row) {
    return (mix32(row * 0x9e3779b9u + seed) & 0xffffffu) | 0x800001u;
}
__device__ __forceinline__ uint32_t drop_key_hash(uint32_t seed, uint32_t key) {
    return (mix32(key * 0x85ebca6bu + (seed ^ 0x5bd1e995u)) & 0xffffffu) | 0x800001u;
}
__device__ __forceinline__ bool drop_keep2(uint32_t a, uint32_t c, uint32_t thr32) { return __umul24(a, c) >= thr32; }
// Dropout on the output of a Linear (GEMM epilogues, the split-K reduce pass, the LayerNorm backward that regenerates the mask):
// the SAME separable decision, keep(row, col) <=> low32(A[row] * C[col]) >= p * 2^32 with row = the output row (batch folded
// in: b * M + m) and col = the output column.  Rounds 1-3 hashed the flat element index with mix32 - two 32-bit multiplies,
// three xor-shifts, an add per ELEMENT plus the index arithmetic, ~14 instructions per element - in epilogues that were as long
// as the K loop at this model's K = 512 (6.9 non-MFMA VALU per MFMA in the two-blocks-per-CU kernel, profiles/r03_gemm_pmc.txt).
// Here a thread hashes its few rows and its four columns once per pass and spends multiply + compare + select per element.
// Round 5: ONE 24 x 24-bit product leaves 22 free bits per row / column hash - among 8 192 rows about eight pairs
// share a row hash and with it their WHOLE mask (correlation 0.9999), column pairs reached 0.85.  The decision is now the SUM of two
// products, keep <=> low32(A1[row] * C1[col] + A2[row] * C2[col]) >= thr: 32 free bits per side, and two rows that agree
// in A1 still differ by (A2 - A2') * C2[col], which changes with the column.  One v_mul_u32_u24 + one v_mad_u32_u24 per element
// (the four hashes are per row / per column and hoisted); tests/dropout_masks.py linear_keep restates it,
// tests/test_dropout_stats.py bounds every row-row and column-column mask correlation.  (The attention-probability dropout keeps
// the single product: its kernels are bound by VALU issue, three instructions per score.)
// (A2 / C2 = the top 24 bits of ONE more xorshift-multiply round on the 32-bit mix A1 / C1 are cut from: 32 free bits per side
// together.  Bits 8..31 of the same mix do NOT do - rows that agree in A1 then differ by a multiple of 2^16 in A2 and stay
// correlated at 0.94.  Cost, A/B in one session against a -DPA_DROP_ONE_PRODUCT build, 200 steps twice: 4.617 vs 4.560 ms per bf16
// train step (+1.2 %; +1.5 % with a second full mix32 per row / column): the extra v_mad per element in every dropout epilogue and in
// the LayerNorm backward that regenerates the masks.)
__device__ __forceinline__ uint32_t drop_mix_row(uint32_t seed, uint32_t row) { return mix32(row * 0x9e3779b9u + seed); }
__device__ __forceinline__ uint32_t drop_mix_key(uint32_t seed, uint32_t key) { return mix32(key * 0x85ebca6bu + (seed ^ 0x5bd1e995u)); }
__device__ __forceinline__ bool drop_keep_rc(uint32_t seed, uint32_t row, uint32_t col, uint32_t thr32) {
#ifdef PA_DROP_ONE_PRODUCT      // ablation build (round 4's decision): step-time A/B only, the dropout parity tests fail with it
    return drop_keep2(drop_row_hash(seed, row), drop_key_hash(seed, col), thr32);
#endif
    const uint32_t mr = drop_mix_row(seed, row), mc = drop_mix_key(seed, col);
    const uint32_t mr2 = (mr ^ (mr >> 13)) * 0x846ca68bu, mc2 = (mc ^ (mc >> 13)) * 0x7feb352du;    // one more mixing round
    const uint32_t h = __umul24((mr & 0xffffffu) | 0x800001u, (mc & 0xffffffu) | 0x800001u) +
                       __umul24((mr2 >> 8) | 0x800001u, (mc2 >> 8) | 0x800001u);
    return h >= thr32;
}

// max(a, b, c) as ONE v_max3_f32.  fmaxf() must quiet signalling NaNs, so hipcc canonicalises every operand that comes out
// of an MFMA first (v_max_f32 x, x, x): 28 instructions for the row maximum of 16 scores instead of 8.  Scores are never
// NaN (finite operands; masked entries are -inf, which v_max3 orders correctly).
__device__ __forceinline__ float max3f(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// hipcc's hazard recogniser does not look inside an asm statement: a v_max3_f32 that reads an MFMA result needs the
// XDL-write -> VALU-read wait states (8-pass 32x32x16: 12 states) like any other VALU reader, and whether a
// compiler-visible reader happens to sit in between is scheduling luck.  settle_mfma() is placed between
// the last MFMA of a score tile and the first max3f on it: the "+v" operands pin the MFMAs before it and every reader
// after it, the s_nops are the wait states (scalar issue slots of this wave only; the SIMD's other waves keep issuing).
__device__ __forceinline__ void settle_mfma(f32x16& a) { asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a)); }
__device__ __forceinline__ void settle_mfma(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Launch + error check.  hipGetLastError() is sticky per thread: a stale error left behind by some unrelated
// earlier runtime call (e.g. the host framework probing devices) must not be blamed on this launch, so it is
// cleared first.
#define PA_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); \
        hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
