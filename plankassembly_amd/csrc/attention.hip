// Flash-style multi-head attention for gfx950 (forward, backward dK/dV, backward dQ).
// See include/plank_hip.h (pa_attn_fwd / pa_attn_bwd).
//
// Formulation.  Every product is computed "transposed" so that per-query softmax statistics are
// lane-local in the 32x32 MFMA C/D layout (col = lane & 31):
//   forward   S^T[key][q] = K Q^T          (A = K rows from LDS, B = this lane's Q row, registers)
//             O^T[d][q]  += V^T P^T        (A = V^T rows from a TRANSPOSED LDS image, B = P^T straight
//                                           from the S^T accumulator registers - no cross-lane moves:
//                                           the 8 contraction slots of a lane are the 8 keys it already
//                                           holds, and V^T is read with the same key labelling)
//   bwd dQ    dP^T = V dO^T, dS^T = P^T o (dP^T - D),  dQ^T += K^T dS^T   (same shapes as forward)
//   bwd dK/dV S = Q K^T, dP = dO V^T (lane = key), dV^T += dO^T P, dK^T += Q^T dS
// Tiles: 128 rows of the "owned" side per block (4 waves x 32), 64 rows of the streamed side per
// step, staged global -> registers -> LDS (double buffered, one barrier per step) as a natural
// [row][dh] image and/or a transposed [dh][row] image (4 x EB register blocks transposed in
// flight).  LDS images are XOR-swizzled in 16-byte chunks.  Scores never touch HBM.
// Softmax runs in the log2 domain (exp2), f32 statistics.
#include <stdio.h>
#include <atomic>
#include <algorithm>
#include "pa_device.h"
#include "../../include/plank_hip.h"

namespace {

constexpr int NTH = 256;
constexpr int BOWN = 128;   // rows owned by a block (32 per wave)
constexpr int BSTR = 64;    // rows streamed per step
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float RESCALE_THR = 8.0f;   // log2 units: running max is only raised when it grows by more than 2^8

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnP;
struct AttnP {
    const void* q; const void* k; const void* v; void* o; float* lse; const uint8_t* kpm;
    int B, H, Lq, Lk;
    int ldq, ldk, ldv, ldo;
    int causal; float scale;
    uint32_t drop_thr; float drop_scale; uint32_t drop_seed;   // drop_thr = p * 2^32 (pa_device.h drop_keep2)
    const void* dout; void* dq; void* dk; void* dv; float* delta;
    int lddo, lddq, lddk, lddv;
    const int32_t* cu_q; const int32_t* cu_k;     // packed (variable-length) row offsets per batch element, or NULL
    const int32_t* order;                         // batch elements in dispatch order (longest first), or NULL
    int balanced;                                 // self-attention over packed rows with `order`: decode_block_balanced
    int ks_min;                                   // in-block key split (KS = 2 kernels): elements with fewer key tiles run unsplit
    int parts_q, parts_kv;                        // bf16x3 backward (attention_x3.h): blocks per owned tile, the streamed side cut in ranges
    // balanced == 2 (round 6, packed self-attention): the STREAMED side of a long element is cut into up to sp_pmax ranges of at
    // most sp_kmax 64-row tiles, one block each (decode_unit_split); the range blocks of an owned tile meet at a ticket and
    // the last one to arrive merges the others' partial results (split_publish / split_arrive)
    int* sp_tick; char* sp_part; int sp_slots, sp_pmax, sp_kmax;
};
constexpr int SP_BYTES = 64 * 1024;               // partial results of one (owned tile, range) block

// Variable-length ("unpadded") batches: with cu_q / cu_k given, batch element b owns rows [cu[b], cu[b+1]) of the
// packed Q resp. K/V matrices; Lq / Lk in the launch parameters are then only the maxima (grid size, layout of the
// per-row statistics lse/delta [B][H][Lq_max] and of the dropout counter).  Returns the per-batch view.
// (off, len >= 0: the element's rows when the caller already knows them - balanced self-attention, cu_q == cu_k.)
__device__ __forceinline__ AttnP batch_view(const AttnP& pin, int b, int& qoff, int& koff, int off = 0, int len = -1) {
    AttnP p = pin;
    qoff = b * pin.Lq; koff = b * pin.Lk;
    if (len >= 0 && pin.cu_q == pin.cu_k) { qoff = koff = off; p.Lq = p.Lk = len; return p; }
    if (pin.cu_q) { qoff = pin.cu_q[b]; p.Lq = pin.cu_q[b + 1] - qoff; }
    if (pin.cu_k) { koff = pin.cu_k[b]; p.Lk = pin.cu_k[b + 1] - koff; }
    return p;
}

template <typename T, int DH> struct AT {
    static constexpr int EB = ET<T>::EB, KC = ET<T>::KC;
    static constexpr int NS = DH / KC;                 // mma16B steps over dh
    static constexpr int NDT = (DH + 31) / 32;         // 32-row d tiles of a transposed product
    static constexpr int NCHR = DH / EB;               // 16-byte chunks per natural row
    static constexpr int RBN = DH * sizeof(T);         // natural row bytes
    static constexpr int RBT = BSTR * sizeof(T);       // transposed row bytes (64 columns)
    static constexpr int NCHT = RBT / 16;
    static constexpr int NAT_BYTES = BSTR * RBN;
    static constexpr int TR_BYTES = DH * RBT;          // == NAT_BYTES
    static constexpr int NSB = 16 * NCHR;              // 4-row sub-blocks per streamed tile
    static constexpr int NITEM = (2 * NSB + NTH - 1) / NTH;   // sub-blocks per thread for two tiles
};

// Swizzle key of an LDS row of RB bytes: its 16-byte chunk c sits at position c ^ tile_key(row).  128-byte rows (bf16 dh = 64 tiles
// of the 32-row-wave kernels, incl. the bf16x3 images) - round 6: bits (r1, r2, r1 ^ r3) instead of (r >> 1) & 7.  The old key is
// conflict-free for the ds_read_b128 fragment reads but 2-way conflicted for every ds_read_b64_tr_b16 (a 32-lane group reads rows
// r0 .. r0 + 3 x four neighbouring chunks: rows r0 and r0 + 2 landed on the same bank quads); the new one is conflict-free for both
// (exhaustive search over the GF(2)-linear keys, tools/ubench/r06/swizzle_search.py) and keeps the two properties the kernels'
// address arithmetic relies on: period 16 rows, key(r + 8) = key(r) ^ 4.  -DPA_KEY5_OLD: round 5's key (A/B builds).
template <int RB> __device__ __forceinline__ int tile_key(int row) {
    constexpr int RPB = (RB >= 256) ? 1 : 256 / RB;
#ifndef PA_KEY5_OLD
    if constexpr (RB == 128) return ((row >> 1) & 3) | ((((row >> 1) ^ (row >> 3)) & 1) << 2);
#endif
    return row / RPB;
}
template <int RB> __device__ __forceinline__ int swz_off(int row, int chunk) {
    constexpr int NCH = RB / 16;
    return row * RB + (((chunk ^ tile_key<RB>(row)) & (NCH - 1)) << 4);
}

// ---- staging: a thread owns 4 consecutive rows x one 16-byte chunk of the streamed tile ----------
template <typename T, int DH>
__device__ __forceinline__ void load_sub(u32x4* regs, const T* base, int ld, int row0, int nrows, int sb) {
    using A = AT<T, DH>;
    const int cb = sb % A::NCHR, rb = sb / A::NCHR;
    // unconditional loads on a clamped row (callers have nrows >= 1), zeroed afterwards: a load inside `if (r < nrows)` becomes a
    // branch with a wait behind it and the four rows cost four dependent memory round trips per tile (round 5: the bf16x3 kernels
    // spent ~7 us per 64-key step here)
    u32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + rb * 4 + i;
        const int rc = r < nrows ? r : nrows - 1;
        v[i] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + cb * A::EB);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool in = row0 + rb * 4 + i < nrows;
        regs[i] = in ? v[i] : u32x4{0u, 0u, 0u, 0u};
    }
}
// The same loads WITHOUT the zeroing (round 6): a select on a freshly loaded value makes hipcc wait for the load on the spot, so a
// "prefetch" through load_sub was a synchronous load in front of the step's MFMAs (tools/isa_loop_waits.py-style reading of the bf16x3
// and exact-f32 attention kernels: loads, s_waitcnt vmcnt(2..0), then the products).  The kernels' gload now takes the raw rows and
// their lstore - at the END of the overlapped step - zeroes the rows past the end (mask_sub) right before the LDS stores.
template <typename T, int DH>
__device__ __forceinline__ void load_sub_raw(u32x4* regs, const T* base, int ld, int row0, int nrows, int sb) {
    using A = AT<T, DH>;
    const int cb = sb % A::NCHR, rb = sb / A::NCHR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + rb * 4 + i;
        const int rc = r < nrows ? r : nrows - 1;
        regs[i] = *reinterpret_cast<const u32x4*>(base + (size_t)rc * ld + cb * A::EB);
    }
}
template <typename T, int DH>
__device__ __forceinline__ void mask_sub(u32x4* regs, int row0, int nrows, int sb) {
    using A = AT<T, DH>;
    const int rb = sb / A::NCHR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool in = row0 + rb * 4 + i < nrows;
        regs[i] = in ? regs[i] : u32x4{0u, 0u, 0u, 0u};
    }
}
template <typename T, int DH>
__device__ __forceinline__ void store_nat(const u32x4* regs, char* lds, int sb) {
    using A = AT<T, DH>;
    const int cb = sb % A::NCHR, rb = sb / A::NCHR;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<u32x4*>(lds + swz_off<A::RBN>(rb * 4 + i, cb)) = regs[i];
}
// transposed image: [d][row]; this thread contributes rows rb*4..+3 for d = cb*EB .. +EB-1
template <typename T, int DH> struct StoreTr;
template <int DH> struct StoreTr<float, DH> {
    static __device__ __forceinline__ void run(const u32x4* regs, char* lds, int sb) {
        using A = AT<float, DH>;
        const int cb = sb % A::NCHR, rb = sb / A::NCHR;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x4 o; o[0] = regs[0][e]; o[1] = regs[1][e]; o[2] = regs[2][e]; o[3] = regs[3][e];
            // row d = cb*4+e, columns rb*4..rb*4+3 = 16 bytes = chunk rb
            *reinterpret_cast<u32x4*>(lds + swz_off<A::RBT>(cb * 4 + e, rb)) = o;
        }
    }
};
template <int DH> struct StoreTr<bf16, DH> {
    static __device__ __forceinline__ void run(const u32x4* regs, char* lds, int sb) {
        using A = AT<bf16, DH>;
        const int cb = sb % A::NCHR, rb = sb / A::NCHR;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            u32x2 o;
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
                const uint32_t lo = regs[2 * qd][e >> 1], hi = regs[2 * qd + 1][e >> 1];
                o[qd] = (e & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
            }
            // row d = cb*8+e, columns rb*4..+3 = 8 bytes at byte rb*8: chunk rb>>1, sub (rb&1)*8
            *reinterpret_cast<u32x2*>(lds + swz_off<A::RBT>(cb * 8 + e, rb >> 1) + (rb & 1) * 8) = o;
        }
    }
};

// ---- MFMA helpers ------------------------------------------------------------------------------------
// acc(32 x 32) += NAT[row0 + (lane&31)][:] (A operand, contraction over dh) x regs (B operand)
template <typename T, int DH>
__device__ __forceinline__ void mma_nat(f32x16& acc, const char* nat, int row0, const u32x4* regs, int lane) {
    using A = AT<T, DH>;
    const int row = row0 + (lane & 31), half = lane >> 5;
#pragma unroll
    for (int s = 0; s < A::NS; ++s) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(nat + swz_off<A::RBN>(row, 2 * s + half));
        mma16B<T>(acc, a, regs[s]);
    }
}
// acc[dt](32 d x 32) += TR[dt*32 + (lane&31)][col0 + slots] (A operand) x P (B operand = this lane's 16
// accumulator values of a 32-slot sub-tile; slot labelling: value r <-> column 8*(r>>2) + 4*half + (r&3))
template <typename T, int DH> struct MmaTr;
template <int DH> struct MmaTr<float, DH> {
    static __device__ __forceinline__ void run(f32x16* acc, const char* tr, int col0, const f32x16& pv, int lane) {
        using A = AT<float, DH>;
        const int half = lane >> 5;
#pragma unroll
        for (int dt = 0; dt < A::NDT; ++dt) {
            const int d = dt * 32 + (lane & 31);
            const bool ok = d < DH;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = col0 + 8 * g + 4 * half;            // 4 consecutive f32 = one 16-byte chunk
                u32x4 a = {0u, 0u, 0u, 0u};
                if (ok) a = *reinterpret_cast<const u32x4*>(tr + swz_off<A::RBT>(d, col >> 2));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[e]), pv[4 * g + e], acc[dt], 0, 0, 0);
            }
        }
    }
};
template <int DH> struct MmaTr<bf16, DH> {
    static __device__ __forceinline__ void run(f32x16* acc, const char* tr, int col0, const f32x16& pv, int lane) {
        using A = AT<bf16, DH>;
        const int half = lane >> 5;
        u32x4 pb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) pb[u][w] = pack_bf16(pv[8 * u + 2 * w], pv[8 * u + 2 * w + 1]);
#pragma unroll
        for (int dt = 0; dt < A::NDT; ++dt) {
            const int d = dt * 32 + (lane & 31);
            const bool ok = d < DH;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = col0 + 16 * u + 4 * half;           // bytes col*2: chunk (col>>3), sub (col&7)*2
                u32x4 a = {0u, 0u, 0u, 0u};
                if (ok) {
                    const u32x2 a0 = *reinterpret_cast<const u32x2*>(tr + swz_off<A::RBT>(d, col >> 3) + (col & 7) * 2);
                    const u32x2 a1 = *reinterpret_cast<const u32x2*>(tr + swz_off<A::RBT>(d, (col + 8) >> 3) + (col & 7) * 2);
                    a[0] = a0[0]; a[1] = a0[1]; a[2] = a1[0]; a[3] = a1[1];
                }
                mma16B<bf16>(acc[dt], a, pb[u]);
            }
        }
    }
};

// this lane's row (q or key) operand for mma_nat's B side: NS 16-byte vectors
template <typename T, int DH>
__device__ __forceinline__ void load_row_regs(u32x4* regs, const T* base, int ld, int row, int nrows, int lane) {
    using A = AT<T, DH>;
    const int half = lane >> 5;
#pragma unroll
    for (int s = 0; s < A::NS; ++s) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + A::KC * s + A::EB * half);
        regs[s] = v;
    }
}

// store a transposed-product accumulator (lane = row, registers = d) as rows of [row][d]
template <typename T, int DH>
__device__ __forceinline__ void store_rows(T* base, int ld, int row, int nrows, const f32x16* acc, float mul, int lane, bool zero = false) {
    using A = AT<T, DH>;
    if (row >= nrows) return;
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = dt * 32 + 8 * g + 4 * (lane >> 5);
            if (d < DH) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = zero ? 0.f : acc[dt][4 * g + e] * mul;
                st4<T>(base + (size_t)row * ld + d, o);
            }
        }
}

template <typename T, int DH> struct Smem {
    using A = AT<T, DH>;
    // [buf][X tile | Y tile | Z tile | W tile | 64 mask bytes + 64 lse + 64 delta]
    static constexpr int AUX_BYTES = 64 + 2 * 64 * 4;
    static constexpr int BUF_FWD = 2 * A::NAT_BYTES + 64;
    static constexpr int BUF_DQ = 3 * A::NAT_BYTES + 64;
    static constexpr int BUF_DKV = 4 * A::NAT_BYTES + 2 * 64 * 4;
};

// =====================================================================================================
// forward
template <typename T, int DH>
__global__ __launch_bounds__(NTH) void attn_fwd_kernel(AttnP pin) {
    using A = AT<T, DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (q0 >= p.Lq) return;
    const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qrow = q0 + wave * 32 + (lane & 31);

    u32x4 qreg[A::NS];
    load_row_regs<T, DH>(qreg, Qp, p.ldq, qrow, p.Lq, lane);

    int nsteps = (p.Lk + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);

    f32x16 oacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl = p.scale * LOG2E;

    u32x4 st[A::NITEM][4];
    uint8_t mreg = 0;
    int st_row0 = 0;                             // first row of the tile held in `st` (its rows past Lk are zeroed in lstore)
    auto gload = [&](int step) {
        const int k0 = step * BSTR;
        st_row0 = k0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<T, DH>(st[j], Kp, p.ldk, k0, p.Lk, sb);
                else load_sub_raw<T, DH>(st[j], Vp, p.ldv, k0, p.Lk, sb);
            }
        }
        // (every thread loads a byte - no guard: a load under `if` is a masked definition hipcc merges with a copy of the loaded
        //  register, i.e. a wait right behind it; without a mask the byte comes from the K rows and is ignored in lstore)
        mreg = (mp ? mp : reinterpret_cast<const uint8_t*>(Kp))[min(k0 + (tid & (BSTR - 1)), p.Lk - 1)];
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * Smem<T, DH>::BUF_FWD;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                mask_sub<T, DH>(st[j], st_row0, p.Lk, sb);
                if (item < A::NSB) store_nat<T, DH>(st[j], base, sb);
                else StoreTr<T, DH>::run(st[j], base + A::NAT_BYTES, sb);
            }
        }
        if (tid < BSTR) reinterpret_cast<uint8_t*>(base + 2 * A::NAT_BYTES)[tid] = (st_row0 + tid >= p.Lk) ? (uint8_t)1 : (mp ? mreg : (uint8_t)0);
    };

    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) gload(step + 1);
        const char* knat = smem + buf * Smem<T, DH>::BUF_FWD;
        const char* vtr = knat + A::NAT_BYTES;
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(knat + 2 * A::NAT_BYTES);
        const int k0 = step * BSTR;

        f32x16 sacc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
            mma_nat<T, DH>(sacc[kt], knat, kt * 32, qreg, lane);
        }
        // scores -> log2 domain, masks
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int koff = kt * 32 + 8 * g + 4 * half;
                const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mk + koff);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + koff + e;
                    float x = sacc[kt][4 * g + e] * sl;
                    const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && key > qrow);
                    x = masked ? -INFINITY : x;
                    sacc[kt][4 * g + e] = x;
                    mx = fmaxf(mx, x);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // Deferred rescale: the running max (any upper-bound-ish reference works for softmax) is raised only
        // when some row of this wave outgrew it by more than 2^RESCALE_THR; then O and l are rescaled once.
        // Scaling by a power-of-two-ish factor does not change floating-point relative error, and l, O and the
        // LSE all use the same reference, so the result is exact up to rounding.
        if (__any(mx > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float ms = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - ms);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            m_run = m_new;
        }
        const float m_safe = (m_run == -INFINITY) ? 0.f : m_run;
        float lsum = 0.f;
        const uint32_t arow = drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow));
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(sacc[kt][4 * g + e] - m_safe);
                    lsum += pe;
                    if (p.drop_thr) {
                        const uint32_t ck = drop_key_hash(p.drop_seed, (uint32_t)(k0 + kt * 32 + 8 * g + 4 * half + e));
                        pe = drop_keep2(arow, ck, p.drop_thr) ? pe * p.drop_scale : 0.f;
                    }
                    sacc[kt][4 * g + e] = pe;
                }
            }
        l_run += lsum;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) MmaTr<T, DH>::run(oacc, vtr, kt * 32, sacc[kt], lane);

        if (step + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    T* Op = reinterpret_cast<T*>(p.o) + (size_t)qoff * p.ldo + h * DH;
    store_rows<T, DH>(Op, p.ldo, qrow, p.Lq, oacc, inv, lane);
    if (half == 0 && qrow < p.Lq && p.lse)
        p.lse[((size_t)b * p.H + h) * pin.Lq + qrow] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : 0.f;
}

// =====================================================================================================
// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnP pin) {
    const int64_t total = (int64_t)pin.B * pin.H * pin.Lq;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % pin.Lq);
    const int h = (int)((i / pin.Lq) % pin.H);
    const int b = (int)(i / ((int64_t)pin.Lq * pin.H));
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if constexpr (sizeof(T) == 4) {
        // range-split bf16x3 backward: its blocks ADD into dq (dk, dv) - zero the rows here, in the launch that runs first anyway
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if (pin.parts_q > 1 && q < p.Lq) {
            float* r = reinterpret_cast<float*>(p.dq) + ((size_t)qoff + q) * p.lddq + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) *reinterpret_cast<f32x4*>(r + c) = z4;
        }
        if (pin.parts_kv > 1 && q < p.Lk) {          // (the host only splits dK / dV when every key row has a thread: Lk <= Lq)
            float* rk = reinterpret_cast<float*>(p.dk) + ((size_t)koff + q) * p.lddk + h * DH;
            float* rv = reinterpret_cast<float*>(p.dv) + ((size_t)koff + q) * p.lddv + h * DH;
#pragma unroll
            for (int c = 0; c < DH; c += 4) { *reinterpret_cast<f32x4*>(rk + c) = z4; *reinterpret_cast<f32x4*>(rv + c) = z4; }
        }
    }
    if (q >= p.Lq) { p.delta[i] = 0.f; return; }
    const T* o = reinterpret_cast<const T*>(p.o) + ((size_t)qoff + q) * p.ldo + h * DH;
    const T* g = reinterpret_cast<const T*>(p.dout) + ((size_t)qoff + q) * p.lddo + h * DH;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
        const f32x4 a = ld4<T>(o + c), d4 = ld4<T>(g + c);
        s += a[0] * d4[0] + a[1] * d4[1] + a[2] * d4[2] + a[3] * d4[3];
    }
    p.delta[i] = s;
}

// =====================================================================================================
// backward, dQ: block owns 128 queries, streams keys
template <typename T, int DH>
__global__ __launch_bounds__(NTH) void attn_bwd_dq_kernel(AttnP pin) {
    using A = AT<T, DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (q0 >= p.Lq) return;
    const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const T* dOp = reinterpret_cast<const T*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qrow = q0 + wave * 32 + (lane & 31);

    u32x4 qreg[A::NS], doreg[A::NS];
    load_row_regs<T, DH>(qreg, Qp, p.ldq, qrow, p.Lq, lane);
    load_row_regs<T, DH>(doreg, dOp, p.lddo, qrow, p.Lq, lane);
    const size_t srow = ((size_t)b * p.H + h) * pin.Lq + qrow;
    const float lse2 = (qrow < p.Lq) ? p.lse[srow] * LOG2E : INFINITY;
    const float dlt = (qrow < p.Lq) ? p.delta[srow] : 0.f;

    int nsteps = (p.Lk + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);

    f32x16 dqacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    const float sl = p.scale * LOG2E;
    const uint32_t arow = drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow));

    u32x4 st[A::NITEM][4];
    uint8_t mreg = 0;
    int st_row0 = 0;                             // first row of the tile held in `st` (its rows past Lk are zeroed in lstore)
    auto gload = [&](int step) {
        const int k0 = step * BSTR;
        st_row0 = k0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<T, DH>(st[j], Kp, p.ldk, k0, p.Lk, sb);
                else load_sub_raw<T, DH>(st[j], Vp, p.ldv, k0, p.Lk, sb);
            }
        }
        // (every thread loads a byte - no guard: a load under `if` is a masked definition hipcc merges with a copy of the loaded
        //  register, i.e. a wait right behind it; without a mask the byte comes from the K rows and is ignored in lstore)
        mreg = (mp ? mp : reinterpret_cast<const uint8_t*>(Kp))[min(k0 + (tid & (BSTR - 1)), p.Lk - 1)];
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * Smem<T, DH>::BUF_DQ;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                mask_sub<T, DH>(st[j], st_row0, p.Lk, sb);
                if (item < A::NSB) {
                    store_nat<T, DH>(st[j], base, sb);                              // K natural
                    StoreTr<T, DH>::run(st[j], base + A::NAT_BYTES, sb);            // K transposed
                } else {
                    store_nat<T, DH>(st[j], base + 2 * A::NAT_BYTES, sb);           // V natural
                }
            }
        }
        if (tid < BSTR) reinterpret_cast<uint8_t*>(base + 3 * A::NAT_BYTES)[tid] = (st_row0 + tid >= p.Lk) ? (uint8_t)1 : (mp ? mreg : (uint8_t)0);
    };

    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        if (step + 1 < nsteps) gload(step + 1);
        const char* knat = smem + buf * Smem<T, DH>::BUF_DQ;
        const char* ktr = knat + A::NAT_BYTES;
        const char* vnat = knat + 2 * A::NAT_BYTES;
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(knat + 3 * A::NAT_BYTES);
        const int k0 = step * BSTR;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            mma_nat<T, DH>(sacc, knat, kt * 32, qreg, lane);
            mma_nat<T, DH>(dpacc, vnat, kt * 32, doreg, lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int koff = kt * 32 + 8 * g + 4 * half;
                const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mk + koff);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = k0 + koff + e;
                    const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && key > qrow);
                    const float pe = masked ? 0.f : fast_exp2(sacc[4 * g + e] * sl - lse2);
                    float dp = dpacc[4 * g + e];
                    if (p.drop_thr)
                        dp = drop_keep2(arow, drop_key_hash(p.drop_seed, (uint32_t)key), p.drop_thr) ? dp * p.drop_scale : 0.f;
                    sacc[4 * g + e] = pe * (dp - dlt) * p.scale;          // dS^T
                }
            }
            MmaTr<T, DH>::run(dqacc, ktr, kt * 32, sacc, lane);
        }
        if (step + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    T* dQp = reinterpret_cast<T*>(p.dq) + (size_t)qoff * p.lddq + h * DH;
    store_rows<T, DH>(dQp, p.lddq, qrow, p.Lq, dqacc, 1.0f, lane);
}

// =====================================================================================================
// backward, dK/dV: block owns 128 keys, streams queries
template <typename T, int DH>
__global__ __launch_bounds__(NTH) void attn_bwd_dkv_kernel(AttnP pin) {
    using A = AT<T, DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y, key0 = blockIdx.x * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (key0 >= p.Lk) return;
    const T* Qp = reinterpret_cast<const T*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const T* Kp = reinterpret_cast<const T*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const T* Vp = reinterpret_cast<const T*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const T* dOp = reinterpret_cast<const T*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const int krow = key0 + wave * 32 + (lane & 31);
    const bool kmasked = (krow >= p.Lk) || (p.kpm && p.kpm[(size_t)b * pin.Lk + krow]);

    u32x4 kreg[A::NS], vreg[A::NS];
    load_row_regs<T, DH>(kreg, Kp, p.ldk, krow, p.Lk, lane);
    load_row_regs<T, DH>(vreg, Vp, p.ldv, krow, p.Lk, lane);

    const int nsteps = (p.Lq + BSTR - 1) / BSTR;
    const int step0 = p.causal ? (key0 / BSTR) : 0;       // queries before the first owned key see none of them

    f32x16 dkacc[A::NDT], dvacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[dt][r] = 0.f; dvacc[dt][r] = 0.f; }
    const float sl = p.scale * LOG2E;

    u32x4 st[A::NITEM][4];
    float lreg = 0.f, dreg = 0.f;
    int st_row0 = 0;                             // first row of the tile held in `st` (rows past Lq are zeroed / neutralised in lstore)
    auto gload = [&](int step) {
        const int r0 = step * BSTR;
        st_row0 = r0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<T, DH>(st[j], Qp, p.ldq, r0, p.Lq, sb);
                else load_sub_raw<T, DH>(st[j], dOp, p.lddo, r0, p.Lq, sb);
            }
        }
        {   // (every thread, no guard - see the forward kernel's mask byte)
            const size_t srow = ((size_t)b * p.H + h) * pin.Lq + min(r0 + (tid & (BSTR - 1)), p.Lq - 1);
            lreg = p.lse[srow];                  // (raw; scaled / replaced for rows past Lq in lstore)
            dreg = p.delta[srow];
        }
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * Smem<T, DH>::BUF_DKV;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                char* t0 = base + (item < A::NSB ? 0 : 2 * A::NAT_BYTES);
                mask_sub<T, DH>(st[j], st_row0, p.Lq, sb);
                store_nat<T, DH>(st[j], t0, sb);
                StoreTr<T, DH>::run(st[j], t0 + A::NAT_BYTES, sb);
            }
        }
        if (tid < BSTR) {
            float* aux = reinterpret_cast<float*>(base + 4 * A::NAT_BYTES);
            const bool in = st_row0 + tid < p.Lq;
            aux[tid] = in ? lreg * LOG2E : INFINITY; aux[64 + tid] = in ? dreg : 0.f;
        }
    };

    if (step0 < nsteps) { gload(step0); lstore(0); }
    __syncthreads();

    for (int step = step0; step < nsteps; ++step) {
        const int buf = (step - step0) & 1;
        if (step + 1 < nsteps) gload(step + 1);
        const char* qnat = smem + buf * Smem<T, DH>::BUF_DKV;
        const char* qtr = qnat + A::NAT_BYTES;
        const char* donat = qnat + 2 * A::NAT_BYTES;
        const char* dotr = qnat + 3 * A::NAT_BYTES;
        const float* aux = reinterpret_cast<const float*>(qnat + 4 * A::NAT_BYTES);
        const int r0 = step * BSTR;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            mma_nat<T, DH>(sacc, qnat, qt * 32, kreg, lane);       // S[q][key]: rows q (regs), col key (lane)
            mma_nat<T, DH>(dpacc, donat, qt * 32, vreg, lane);     // dP[q][key]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qoff = qt * 32 + 8 * g + 4 * half;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(aux + qoff);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(aux + 64 + qoff);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qr = r0 + qoff + e;
                    const bool masked = kmasked || (p.causal && krow > qr);
                    float pe = masked ? 0.f : fast_exp2(sacc[4 * g + e] * sl - l4[e]);
                    float dp = dpacc[4 * g + e];
                    float pd = pe;
                    if (p.drop_thr) {
                        const uint32_t ar = drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qr));
                        const bool keep = drop_keep2(ar, drop_key_hash(p.drop_seed, (uint32_t)krow), p.drop_thr);
                        dp = keep ? dp * p.drop_scale : 0.f;
                        pd = keep ? pe * p.drop_scale : 0.f;
                    }
                    sacc[4 * g + e] = pd;                                   // dropped P  -> dV
                    dpacc[4 * g + e] = pe * (dp - d4[e]) * p.scale;         // dS         -> dK
                }
            }
            MmaTr<T, DH>::run(dvacc, dotr, qt * 32, sacc, lane);
            MmaTr<T, DH>::run(dkacc, qtr, qt * 32, dpacc, lane);
        }
        if (step + 1 < nsteps) lstore(buf ^ 1);
        __syncthreads();
    }
    T* dKp = reinterpret_cast<T*>(p.dk) + (size_t)koff * p.lddk + h * DH;
    T* dVp = reinterpret_cast<T*>(p.dv) + (size_t)koff * p.lddv + h * DH;
    store_rows<T, DH>(dKp, p.lddk, krow, p.Lk, dkacc, 1.0f, lane);
    store_rows<T, DH>(dVp, p.lddv, krow, p.Lk, dvacc, 1.0f, lane);
}

// =====================================================================================================
// bf16 kernels, v2: K / V (resp. Q / dO) tiles are DMA'd (global_load_lds) into LDS in their NATURAL [row][dh]
// layout only; the transposed MFMA operands (V^T for O^T += V^T P^T, K^T for dQ^T, Q^T / dO^T for dK^T / dV^T) are
// formed by ds_read_b64_tr_b16 from that same image (in each 16-lane group lane L supplies row L>>2, column
// 4*(L&3) of a 4 x 16 patch; lane i receives column i).  No staging registers, no in-register transposes, half
// the LDS of v1 for the backward kernels.
typedef short s16x4 __attribute__((ext_vector_type(4)));

// Blocks are launched as a 1-D grid and decoded XCD-aware: the hardware places block L on XCD L % 8, and the blocks
// that stream the same (batch, head) K/V (or Q/dO) panel should share one XCD's L2.  XCD x therefore owns the pairs
// p = x (mod 8) and walks all their tiles; falls back to the plain order when #pairs is not a multiple of 8.
__device__ __forceinline__ void decode_block(int ntiles, int H, int Bn, int& tile, int& h, int& b, int L = blockIdx.x) {
    const int npairs = H * Bn;
    int pair;
    if ((npairs & 7) == 0) {
        const int x = L & 7, j = L >> 3;
        tile = j % ntiles;
        pair = (j / ntiles) * 8 + x;
    } else {
        tile = L % ntiles;
        pair = L / ntiles;
    }
    h = pair % H;
    b = pair / H;
}
// Variable-length batches: all blocks of a launch start together (a few per CU), so the launch lasts as long as the CU
// that drew the most work.  With `order` (batch elements by descending length) consecutive blocks carry descending work
// and the round-robin placement gives every CU one long, one medium and one short block.
__device__ __forceinline__ int dispatch_batch(const int32_t* order, int b) { return order ? order[b] : b; }
// Packed self-attention with H == 8 (XCD x = head x): the (element, tile) blocks of a head, sorted by descending work
// (= element length; `order` + the tile counts from cu), are dealt to the XCD's 32 CUs boustrophedon - row 0 left to
// right, row 1 right to left, ... - so the CU that got the longest block of one row gets the shortest of the next.
// Blocks past the last valid one exit.  Every wave computes the same mapping from B <= 64 lanes (no LDS, no barrier).
// `off` / `len`: the element's first row and row count in `cu`, from the lane that loaded them (saves the caller a second,
// dependent round of loads - ~800 cycles of every block's prologue).
__device__ __forceinline__ bool decode_block_balanced(const AttnP& pin, const int32_t* cu, int& tile, int& h, int& b, int& off, int& len, int L = blockIdx.x, int rows = BOWN) {
    const int x = L & 7, j = L >> 3;
    const int lane = threadIdx.x & 63;
    int o = 0, t = 0, c0 = 0, cl = 0;
    if (lane < pin.B) { o = pin.order[lane]; c0 = cu[o]; cl = cu[o + 1] - c0; t = (cl + rows - 1) / rows; }
    int inc = t;                                             // inclusive prefix of the tile counts over the ranks
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    const int n = __shfl(inc, 63);
    const int row = j >> 5, pos = j & 31, left = n - (row << 5);
    if (left <= 0) return false;
    const int rowlen = left < 32 ? left : 32;
    if (pos >= rowlen) return false;
    const int k = (row << 5) + ((row & 1) ? rowlen - 1 - pos : pos);
    const int rank = __popcll(__ballot(inc <= k));           // elements whose blocks all come before block k
    tile = __builtin_amdgcn_readfirstlane(k - (__shfl(inc, rank) - __shfl(t, rank)));
    b = __builtin_amdgcn_readfirstlane(__shfl(o, rank));
    off = __builtin_amdgcn_readfirstlane(__shfl(c0, rank));
    len = __builtin_amdgcn_readfirstlane(__shfl(cl, rank));
    h = x;
    return true;
}


// ---- balanced == 2: units = (element, owned 128-row tile, range of the streamed side), longest elements first -------------
// Why (profiles/r05_attention_shape_sweep.txt item 4, profiles/r06_attention_launch_shape.txt): every block of the packed launch is resident
// from the start, so the launch lasts as long as the serial chain of its longest block - 15-16 key tiles for a 1 000-row element
// against ~8 on average - while the CUs that drew short blocks idle.  Cutting the long chains makes the launch's duration its
// WORK: an element with kt >= sp_kmax + 1 streamed tiles runs as np = min(sp_pmax, ceil(kt / sp_kmax)) blocks per owned tile.
// Blocks are numbered longest element first (LPT; the hardware hands blocks to CU slots in index order as slots free up), the
// ranges of one owned tile are neighbours (they finish together: the merge rarely waits for data to become visible - it never
// WAITS at all, the last arriver merges).  Every wave computes the mapping from B <= 64 lanes, as decode_block_balanced does.
struct SplitUnit { int tile, h, b, off, len, part, nparts, slot; };
__device__ __forceinline__ bool decode_unit_split(const AttnP& pin, const int32_t* cu, SplitUnit& u, int L = blockIdx.x) {
    const int x = L & 7, j = L >> 3;
    const int lane = threadIdx.x & 63;
    int o = 0, c0 = 0, cl = 0, tq = 0, np = 1;
    if (lane < pin.B) {
        o = pin.order[lane]; c0 = cu[o]; cl = cu[o + 1] - c0; tq = (cl + BOWN - 1) / BOWN;
        const int ks = (cl + BSTR - 1) / BSTR;
        if (ks > pin.sp_kmax) np = min(pin.sp_pmax, (ks + pin.sp_kmax - 1) / pin.sp_kmax);
    }
    int sinc = np > 1 ? tq : 0;                              // inclusive prefix of the split elements' owned tiles: merge slots
    const int spl = sinc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(sinc, d); if (lane >= d) sinc += v; }
    if (sinc > pin.sp_slots) np = 1;                          // no slot left (scratch sized for fewer rows): this element runs unsplit
    const int t = tq * np;
    int inc = t;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(inc, d); if (lane >= d) inc += v; }
    // units in descending work, dealt to the XCD's 32 CUs boustrophedon like decode_block_balanced's blocks (the first rows are
    // all resident from the start - placement IS the balance there; later rows go to whichever slot frees up)
    const int n = __shfl(inc, 63);
    const int row = j >> 5, pos = j & 31, left = n - (row << 5);
    if (left <= 0) return false;
    const int rowlen = left < 32 ? left : 32;
    if (pos >= rowlen) return false;
    const int k = (row << 5) + ((row & 1) ? rowlen - 1 - pos : pos);
    const int rank = __popcll(__ballot(inc <= k));           // elements whose blocks all come before block k
    const int local = k - (__shfl(inc, rank) - __shfl(t, rank));
    const int npr = __shfl(np, rank);
    u.nparts = __builtin_amdgcn_readfirstlane(npr);
    u.tile = __builtin_amdgcn_readfirstlane(local / npr);
    u.part = __builtin_amdgcn_readfirstlane(local % npr);
    u.b = __builtin_amdgcn_readfirstlane(__shfl(o, rank));
    u.off = __builtin_amdgcn_readfirstlane(__shfl(c0, rank));
    u.len = __builtin_amdgcn_readfirstlane(__shfl(cl, rank));
    u.slot = __builtin_amdgcn_readfirstlane(((__shfl(sinc, rank) - __shfl(spl, rank)) + u.tile) * 8 + x);
    u.h = x;
    return true;
}
#include "split_merge.h"


template <int DH> struct BT {
    static constexpr int RBN = DH * 2;                 // natural row bytes
    static constexpr int NCHR = RBN / 16;              // 16-byte chunks per row
    static constexpr int NAT = BSTR * RBN;             // bytes of a 64-row natural tile
    static constexpr int NCHUNK = BSTR * NCHR;         // chunks per tile
    static constexpr int NLD = (NCHUNK + NTH - 1) / NTH;
    static constexpr int NS = DH / 16;
    static constexpr int NDT = (DH + 31) / 32;
};

// DMA one 64-row natural tile (rows row0.. of a [nrows][ld] bf16 matrix, DH columns at column offset 0 of `base`)
template <int DH>
__device__ __forceinline__ void glds_nat(char* lds, const bf16* base, int ld, int row0, int nrows, int tid, int wave) {
    using B = BT<DH>;
    constexpr int RPB = (B::RBN >= 256) ? 1 : 256 / B::RBN;
#pragma unroll
    for (int i = 0; i < B::NLD; ++i) {
        const int p = tid + i * NTH;
        if (B::NCHUNK % NTH == 0 || p < B::NCHUNK) {
            const int row = p / B::NCHR;
            const int ch = ((p % B::NCHR) ^ tile_key<B::RBN>(row)) & (B::NCHR - 1);       // source chunk that belongs at position p
            const int r = min(row0 + row, nrows - 1);                            // clamp: masked rows still read finite data
            const bf16* src = base + (size_t)r * ld + ch * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                (__attribute__((address_space(3))) void*)(lds + (i * NTH + wave * 64) * 16), 16, 0, 0);
        }
    }
}

// The same tile DMA through a buffer resource (v3 kernels): `voff` is the per-thread byte offset of its first chunk inside a
// tile (tile_voff, loop invariant: ONE VGPR for every tile of a matrix), and every round of a tile gets its own resource
// descriptor (four SGPRs, scalar arithmetic only) whose base is the round's first row and whose size ends at the sample's
// last row: rows past the end read as ZEROS through the hardware range check - no clamp, no 64-bit vector address
// arithmetic in the loop.  (The scalar offset operand of buffer loads is excluded from the range check, which is why the
// row offset moves the base instead.)
struct TileSrc {
    uint64_t base; int ld; int64_t bytes;       // first byte of the sample's rows (head offset included), bytes up to the end of its last row
    __amdgpu_buffer_rsrc_t rs;                  // ONE descriptor for the whole sample (round 4, see glds_tile)
};
__device__ __forceinline__ TileSrc tile_src(const bf16* base, int ld, int nrows, int dh) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    TileSrc t;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane(v); };      // (the builtin returns a signed int)
    t.base = ((uint64_t)uni((uint32_t)(a >> 32)) << 32) | (uint64_t)uni((uint32_t)a);
    t.ld = (int)uni((uint32_t)ld);
    const uint64_t nb = nrows > 0 ? ((uint64_t)(nrows - 1) * ld + dh) * 2 : 0;
    t.bytes = (int64_t)(((uint64_t)uni((uint32_t)(nb >> 32)) << 32) | (uint64_t)uni((uint32_t)nb));
    t.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(t.base), 0, (int)(t.bytes < 0x7fffffff ? t.bytes : 0x7fffffff), 0x00020000);
    return t;
}
template <int DH> __device__ __forceinline__ int tile_voff(int ld, int tid) {
    using B = BT<DH>;
    constexpr int RPB = (B::RBN >= 256) ? 1 : 256 / B::RBN;
    const int row = tid / B::NCHR;
    const int ch = ((tid % B::NCHR) ^ tile_key<B::RBN>(row)) & (B::NCHR - 1);           // source chunk that belongs at position tid
    return (row * ld + ch * 8) * 2;
}
// Round 4: the tile's first row goes into the per-lane offset (one scalar multiply + one vector add per DMA instruction) and the
// descriptor is the sample's, built once.  Rounds 1-3 rebuilt a descriptor for every DMA instruction - base advanced, size
// reduced, 64-bit scalar arithmetic, ~18 instructions each - and in kernels that are bound by instruction issue
// (profiles/r04_attn_issue_bound.txt) those were a fifth of all issue slots.  Rows past the sample's end still read as zeros:
// the range check compares the per-lane offset with the descriptor's size.
template <int DH>
__device__ __forceinline__ void glds_tile(char* lds, const TileSrc& ts, int voff, int row0, int wave) {
    using B = BT<DH>;
    static_assert(B::NCHUNK % 64 == 0, "whole waves");
    constexpr int RPI = NTH / B::NCHR;                                        // rows covered by one round of the block
#pragma unroll
    for (int i = 0; i < B::NLD; ++i)
        if (B::NCHUNK % NTH == 0 || i * NTH + wave * 64 < B::NCHUNK) {        // (dh = 16: the tile is two waves' worth)
            const int skip = (row0 + i * RPI) * ts.ld * 2;                    // scalar
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.rs, (__attribute__((address_space(3))) void*)(lds + (i * NTH + wave * 64) * 16), 16,
                                                     voff + skip, 0, 0, 0);
        }
}

// acc[dt] (32 d x 32) += NAT^T[d][rows] (A operand via transposing reads) x P (B operand = this lane's 16 accumulator
// values of a 32-row sub-tile starting at row0; value r <-> row 8*(r>>2) + 4*half + (r&3))
template <int DH>
__device__ __forceinline__ void mma_tr_nat(f32x16* acc, const char* nat, int row0, const f32x16& pv, int lane) {
    using B = BT<DH>;
    const int half = lane >> 5, L = lane & 15, G = lane >> 4;
    u32x4 pb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) pb[u][w] = pack_bf16(pv[8 * u + 2 * w], pv[8 * u + 2 * w + 1]);
#pragma unroll
    for (int dt = 0; dt < B::NDT; ++dt) {
        const int col = dt * 32 + 16 * (G & 1) + 4 * (L & 3);            // dh column this lane fetches for the patch
        const bool ok = col < DH;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x4 a = {0u, 0u, 0u, 0u};
            if (ok) {
                const int r0 = row0 + 16 * u + 4 * half + (L >> 2), r1 = r0 + 8;
                const s16x4 x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(nat + swz_off<B::RBN>(r0, col >> 3) + (col & 7) * 2));
                const s16x4 x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(nat + swz_off<B::RBN>(r1, col >> 3) + (col & 7) * 2));
                const u32x2 a0 = *reinterpret_cast<const u32x2*>(&x0), a1 = *reinterpret_cast<const u32x2*>(&x1);
                a[0] = a0[0]; a[1] = a0[1]; a[2] = a1[0]; a[3] = a1[1];
            }
            mma16B<bf16>(acc[dt], a, pb[u]);
        }
    }
}

// ---- bf16 kernels (v3) -----------------------------------------------------------------------------------------------
// What bounds them: per 64-key step a wave runs 16 (forward) / 24 (dQ) / 32 (dK,dV) MFMAs = 32 cycles each on its SIMD's
// matrix pipe, and one softmax element per lane per half MFMA.  Measured on MI355X (tools/ubench/valu_rate.hip) a single
// wave issues one VALU instruction per ~6.5 cycles whatever the mix; the SIMD reaches its 2.5-3.3 cycles per instruction
// only with >= 3 resident waves.  So the kernels are VALU-issue bound, and the design rules are: (i) as few VALU
// instructions per score as possible - masks only on the tiles that contain a masked key (the key-padding mask is scanned
// once per block for its first masked / last unmasked key; tiles beyond the last unmasked key are skipped altogether),
// dropout as one v_mul_u32_u24 + compare + select per score (pa_device.h drop_keep2; the per-key / per-row hash words
// come from LDS, written once per tile), scale and max folded into one FMA, LDS addresses = loop-invariant base + immediate
// (the two pipeline stages are separate instantiations of the loop body); (ii) four blocks = 16 waves per CU (<= 128
// VGPRs), so that the hardware interleaves the MFMA, VALU and LDS phases of different waves.
// LDS stage: [X tile 64 x DH][Y tile 64 x DH][aux 768 B]; two stages + 32 B for the mask scan.
template <int DH> struct BL {
    static constexpr int NAT = BT<DH>::NAT;
    static constexpr int AUX = 2 * NAT;
    static constexpr int BUF = 2 * NAT + 768;
    static constexpr int SHM = 2 * BUF + 64;
};
template <int V> struct IC { static constexpr int value = V; };
// Block barrier that also publishes this wave's LDS-DMA tiles: hipcc does not count `buffer_load ... lds` among the
// operations __syncthreads() has to wait for (observed: s_waitcnt vmcnt(3) before the s_barrier), so the wait is explicit.
__device__ __forceinline__ void tile_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Key range of a key-padding-mask row: kfirst = first masked key, klast = last unmasked key + 1 (both Lk without a mask).
// Key tiles at or beyond klast contribute exactly zero and are skipped; tiles that end at or before kfirst need no
// per-element mask test.  One pass over the Lk mask bytes by the whole block; contains a barrier.
__device__ __forceinline__ void scan_key_mask(const uint8_t* mp, int Lk, int tid, int* s_scan, int& kfirst, int& klast) {
    int f = Lk, l = 0;
    for (int k = tid; k < Lk; k += NTH) {
        if (mp[k]) f = min(f, k);
        else l = k + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { f = min(f, __shfl_xor(f, o)); l = max(l, __shfl_xor(l, o)); }
    if ((tid & 63) == 0) { s_scan[(tid >> 6) * 2] = f; s_scan[(tid >> 6) * 2 + 1] = l; }
    __syncthreads();
    kfirst = min(min(s_scan[0], s_scan[2]), min(s_scan[4], s_scan[6]));
    klast = max(max(s_scan[1], s_scan[3]), max(s_scan[5], s_scan[7]));
}
// aux slot of key t (0..63) of a tile such that a lane's 32 keys (kt, g, e | half) are 32 consecutive words
__device__ __forceinline__ int key_slot(int t) { return ((t >> 2) & 1) * 32 + (t >> 5) * 16 + ((t >> 3) & 3) * 4 + (t & 3); }

// LDS operand reads of the v3 kernels: hand-placed ds_read instructions on a few loop-invariant per-lane base addresses
// plus immediate offsets (tile, pipeline stage, 32-row sub-tile), so that no address arithmetic and no address registers
// beyond these live in the loop (the compiler's own addressing kept ~50 invariant addresses alive and spilled them).
// hipcc does not track asm-issued LDS reads: every consumer sits behind wait_lds(), which ties the registers through the
// s_waitcnt so that nothing using them can be scheduled above it.
#define PA_DS128(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
#define PA_DSTR(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
template <int N> __device__ __forceinline__ void wait_lds(u32x4 (&f)[N]) {
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]));
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
    else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    else static_assert(N == 1 || N == 2 || N == 4, "fragment count");
}
// Per-lane base addresses (bytes from the start of dynamic LDS, tile offset not included) of a natural [64][DH] bf16 tile:
//   nat[s]  row (lane & 31), 16-byte chunk 2s + half, swizzled             - A operand of S^T = K Q^T style products
//   trA/trB the ds_read_b64_tr_b16 patch of this lane for (dt ^ rsel) = 0 / 1 - A operand of the transposed products
// Every other address of the loop is one of these plus an immediate (32-row sub-tile kt: + kt*32*RBN; 16-row half u of a
// transposed sub-tile: + u*16*RBN; second 8 rows rsel: + 8*RBN; the swizzle term is unaffected by those because
// 32, 16 rows are multiples of the swizzle period, and rsel / dt only flip one chunk bit - which selects trA or trB).
template <int DH> struct LdsBase {
    uint32_t nat[BT<DH>::NS];
    uint32_t trA, trB;
    __device__ __forceinline__ void init(uint32_t smem_base, int lane) {
        using B = BT<DH>;
        const int l = lane & 31, half = lane >> 5, L = lane & 15, G = lane >> 4;
#pragma unroll
        for (int s = 0; s < B::NS; ++s) nat[s] = smem_base + swz_off<B::RBN>(l, 2 * s + half);
        const int col = 16 * (G & 1) + 4 * (L & 3), r = 4 * half + (L >> 2);
        trA = smem_base + r * B::RBN + swz_off<B::RBN>(r, col >> 3) - r * B::RBN + (col & 7) * 2;
        trB = smem_base + r * B::RBN + swz_off<B::RBN>(r + 8, col >> 3) - (r + 8) * B::RBN + (col & 7) * 2;
    }
};
// acc (32 x 32) += TILE[OFF/RBN + (lane & 31)][:] (A operand, contraction over dh) x regs (B operand)
template <int DH, int OFF> __device__ __forceinline__ void mma_nat3(f32x16& acc, const LdsBase<DH>& lb, const u32x4* regs) {
    constexpr int NS = BT<DH>::NS;
    u32x4 a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) PA_DS128(a[s], lb.nat[s], OFF);
    wait_lds(a);
#pragma unroll
    for (int s = 0; s < NS; ++s) mma16B<bf16>(acc, a[s], regs[s]);
}
// acc[dt] (32 d x 32) += TILE^T[d][32 rows at OFF] x P  (P = this lane's 16 accumulator values of the 32-row sub-tile;
// value r <-> row 8*(r>>2) + 4*half + (r&3)); lanes whose dh column lies beyond DH (dh = 16) feed accumulator rows that
// are never stored
template <int DH, int OFF> __device__ __forceinline__ void mma_tr3(f32x16* acc, const LdsBase<DH>& lb, const f32x16& pv) {
    using B = BT<DH>;
    u32x4 pb[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) pb[u][w] = pack_bf16(pv[8 * u + 2 * w], pv[8 * u + 2 * w + 1]);
#pragma unroll
    for (int dt = 0; dt < B::NDT; ++dt) {
        u32x4 a[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            u32x2 x0, x1;
            // rsel = 0 -> chunk bit (dt ^ 0), rsel = 1 -> (dt ^ 1)
            if (dt == 0) { PA_DSTR(x0, lb.trA, OFF + u * 16 * B::RBN); PA_DSTR(x1, lb.trB, OFF + u * 16 * B::RBN + 8 * B::RBN); }
            else { PA_DSTR(x0, lb.trB, OFF + u * 16 * B::RBN); PA_DSTR(x1, lb.trA, OFF + u * 16 * B::RBN + 8 * B::RBN); }
            a[u][0] = x0[0]; a[u][1] = x0[1]; a[u][2] = x1[0]; a[u][3] = x1[1];
        }
        wait_lds(a);
#pragma unroll
        for (int u = 0; u < 2; ++u) mma16B<bf16>(acc[dt], a[u], pb[u]);
    }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(NTH, 4) void attn_fwd_bf16_kernel(AttnP pin) {
    using A = AT<bf16, DH>;
    using B = BT<DH>;
    constexpr int BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, h, b, off_ = 0, len_ = -1;
    if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_q, tile_, h, b, off_, len_)) return; }
    else { decode_block((pin.Lq + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int q0 = tile_ * BOWN;
    if (q0 >= p.Lq) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qrow = q0 + wave * 32 + (lane & 31);

    u32x4 qreg[A::NS];
    load_row_regs<bf16, DH>(qreg, Qp, p.ldq, qrow, p.Lq, lane);

    const TileSrc srcK = tile_src(Kp, p.ldk, p.Lk, DH), srcV = tile_src(Vp, p.ldv, p.Lk, DH);
    const int voffK = tile_voff<DH>(p.ldk, tid), voffV = tile_voff<DH>(p.ldv, tid);
    LdsBase<DH> lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base, lane);
    const uint32_t cbase = smem_base + half * 128;       // this half-wave's 32 key-hash words of a tile (key_slot)
    auto issue = [&](int step, int buf, int kfirst_) {
        char* base = smem + buf * BUF;
        const int k0 = step * BSTR;
        const bool tile_masked = k0 + BSTR > kfirst_;                  // block-uniform
        uint8_t mb = 0;
        if (tile_masked && tid < BSTR) {                               // (loaded before the DMA is issued: its wait must not cover the tiles)
            const int key = k0 + tid;
            mb = (key >= p.Lk) ? 1 : (mp ? mp[key] : 0);
        }
        glds_tile<DH>(base, srcK, voffK, k0, wave);
        glds_tile<DH>(base + B::NAT, srcV, voffV, k0, wave);
        if (tid < BSTR) {
            if (tile_masked) reinterpret_cast<uint8_t*>(base + AUX)[tid] = mb;
            if (DROP) reinterpret_cast<uint32_t*>(base + AUX + 64)[key_slot(tid)] = drop_key_hash(p.drop_seed, (uint32_t)(k0 + tid));
        }
    };
    issue(0, 0, 0);
    int kfirst = p.Lk, klast = p.Lk;
    if (mp) scan_key_mask(mp, p.Lk, tid, reinterpret_cast<int*>(smem + 2 * BUF), kfirst, klast);
    int nsteps = (klast + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);

    f32x16 oacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl = p.scale * LOG2E;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow)) : 0u;
    tile_barrier();

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (step + 1 < nsteps) issue(step + 1, buf ^ 1, kfirst);
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(smem + buf * BUF + AUX);
        // The online softmax advances in 32-key chunks (one 32 x 32 S^T accumulator = 16 registers live instead of 32;
        // what keeps the kernel at four blocks per CU): S^T chunk -> row max -> deferred rescale -> P -> O^T += V^T P^T.
        auto sub = [&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            const int k0 = step * BSTR + kt * 32;
            f32x16 sacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
            mma_nat3<DH, buf * BUF + kt * 32 * B::RBN>(sacc, lb, qreg);
            // Softmax in the log2 domain on the RAW scores (scale folded into one FMA per element); m_run = running max of s * sl.
            const bool key_masked = k0 + 32 > kfirst;                      // (the tile's mask bytes exist: issue() wrote them)
            const bool need_mask = key_masked || (p.causal && (k0 + 31 > q0 + wave * 32));               // wave-uniform
            float mx = -INFINITY;
            if (need_mask) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ko = 8 * g + 4 * half;
                    const uint32_t m4 = key_masked ? *reinterpret_cast<const uint32_t*>(mk + kt * 32 + ko) : 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && k0 + ko + e > qrow);
                        const float x = masked ? -INFINITY : sacc[4 * g + e];
                        sacc[4 * g + e] = x;
                    }
                }
            } else {
                settle_mfma(sacc);        // interior tile: no compiler-visible reader of the MFMA result precedes the asm max3f
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = max3f(mx, sacc[r], sacc[r + 1]);
            mx = fmaxf(mx, __shfl_xor(mx, 32)) * sl;
            // deferred rescale (see the generic kernel): the reference point only moves when some row outgrew it by 2^RESCALE_THR
            if (__any(mx > m_run + RESCALE_THR)) {
                const float m_new = fmaxf(m_run, mx);
                const float ms = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2(m_run - ms);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                m_run = m_new;
            }
            const float nm = (m_run == -INFINITY) ? 0.f : -m_run;
            float lsum[4] = {0.f, 0.f, 0.f, 0.f};                          // four short chains instead of one of 16 adds
            // dropout: survivors are NOT rescaled here - 1/(1-p) is folded into the final normalisation.  The 16 key hashes
            // of this lane's scores stream through two registers: group g + 1 is requested before group g is consumed.
            u32x4 cq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (DROP) PA_DS128(cq[0], cbase, buf * BUF + AUX + 64 + kt * 64);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (DROP) {
                    if (g + 1 < 4) {
                        PA_DS128(cq[(g + 1) & 1], cbase, buf * BUF + AUX + 64 + kt * 64 + (g + 1) * 16);
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cq[g & 1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[g & 1]));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(__builtin_fmaf(sacc[4 * g + e], sl, nm));
                    lsum[e] += pe;
                    if (DROP) pe = drop_keep2(arow, cq[g & 1][e], p.drop_thr) ? pe : 0.f;
                    sacc[4 * g + e] = pe;
                }
            }
            l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
            mma_tr3<DH, buf * BUF + B::NAT + kt * 32 * B::RBN>(oacc, lb, sacc);
        };
        sub(IC<0>{});
        sub(IC<1>{});
        tile_barrier();
    };
    for (int step = 0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
    bf16* Op = reinterpret_cast<bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
    store_rows<bf16, DH>(Op, p.ldo, qrow, p.Lq, oacc, inv, lane);
    if (half == 0 && qrow < p.Lq && p.lse)
        p.lse[((size_t)b * p.H + h) * pin.Lq + qrow] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : 0.f;
}

template <int DH, bool DROP, int OCC>
__global__ __launch_bounds__(NTH, OCC) void attn_bwd_dq_bf16_kernel(AttnP pin) {
    using A = AT<bf16, DH>;
    using B = BT<DH>;
    constexpr int BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, h, b, off_ = 0, len_ = -1;
    if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_q, tile_, h, b, off_, len_)) return; }
    else { decode_block((pin.Lq + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int q0 = tile_ * BOWN;
    if (q0 >= p.Lq) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const bf16* dOp = reinterpret_cast<const bf16*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qrow = q0 + wave * 32 + (lane & 31);

    u32x4 qreg[A::NS], doreg[A::NS];
    load_row_regs<bf16, DH>(qreg, Qp, p.ldq, qrow, p.Lq, lane);
    load_row_regs<bf16, DH>(doreg, dOp, p.lddo, qrow, p.Lq, lane);
    const size_t srow = ((size_t)b * p.H + h) * pin.Lq + qrow;
    const float lse2 = (qrow < p.Lq) ? p.lse[srow] * LOG2E : INFINITY;
    // dS = P o (drop(dP) - delta) * scale with drop(x) = keep ? x / (1-p) : 0  ==  P o ((keep ? dP : 0) - delta * (1-p)) * scale / (1-p):
    // the two constant factors move to the final store, delta is pre-multiplied once
    const float keep_p = DROP ? 1.0f / p.drop_scale : 1.0f;
    // delta[q] = sum_d dO[q][d] * O[q][d] is computed here (each half-wave lane holds half of the row) and published
    // for the dK/dV kernel, which is launched after this one - no separate delta pass
    float dsum = 0.f;
    {
        const bf16* Op = reinterpret_cast<const bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
        u32x4 oreg[A::NS];
        load_row_regs<bf16, DH>(oreg, Op, p.ldo, qrow, p.Lq, lane);
#pragma unroll
        for (int s_ = 0; s_ < A::NS; ++s_)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                dsum += bf16_lo(oreg[s_][w]) * bf16_lo(doreg[s_][w]) + bf16_hi(oreg[s_][w]) * bf16_hi(doreg[s_][w]);
        dsum += __shfl_xor(dsum, 32);
        if (half == 0 && qrow < p.Lq) p.delta[srow] = dsum;
    }
    const float dlt = (qrow < p.Lq) ? dsum * keep_p : 0.f;

    const TileSrc srcK = tile_src(Kp, p.ldk, p.Lk, DH), srcV = tile_src(Vp, p.ldv, p.Lk, DH);
    const int voffK = tile_voff<DH>(p.ldk, tid), voffV = tile_voff<DH>(p.ldv, tid);
    LdsBase<DH> lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base, lane);
    const uint32_t cbase = smem_base + half * 128;       // this half-wave's 32 key-hash words of a tile (key_slot)
    auto issue = [&](int step, int buf, int kfirst_) {
        char* base = smem + buf * BUF;
        const int k0 = step * BSTR;
        const bool tile_masked = k0 + BSTR > kfirst_;
        uint8_t mb = 0;
        if (tile_masked && tid < BSTR) {
            const int key = k0 + tid;
            mb = (key >= p.Lk) ? 1 : (mp ? mp[key] : 0);
        }
        glds_tile<DH>(base, srcK, voffK, k0, wave);
        glds_tile<DH>(base + B::NAT, srcV, voffV, k0, wave);
        if (tid < BSTR) {
            if (tile_masked) reinterpret_cast<uint8_t*>(base + AUX)[tid] = mb;
            if (DROP) reinterpret_cast<uint32_t*>(base + AUX + 64)[key_slot(tid)] = drop_key_hash(p.drop_seed, (uint32_t)(k0 + tid));
        }
    };
    issue(0, 0, 0);
    int kfirst = p.Lk, klast = p.Lk;
    if (mp) scan_key_mask(mp, p.Lk, tid, reinterpret_cast<int*>(smem + 2 * BUF), kfirst, klast);
    int nsteps = (klast + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);

    f32x16 dqacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    const float sl = p.scale * LOG2E;
    const float nl = -lse2;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)srow) : 0u;
    tile_barrier();

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (step + 1 < nsteps) issue(step + 1, buf ^ 1, kfirst);
        const char* knat = smem + buf * BUF;
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(knat + AUX);
        const int k0 = step * BSTR;
        const bool key_masked = k0 + BSTR > kfirst;                       // (the tile's mask bytes exist: issue() wrote them)
        const bool need_mask = key_masked || (p.causal && (k0 + BSTR - 1 > q0 + wave * 32));
        auto sub = [&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            mma_nat3<DH, buf * BUF + kt * 32 * B::RBN>(sacc, lb, qreg);
            mma_nat3<DH, buf * BUF + B::NAT + kt * 32 * B::RBN>(dpacc, lb, doreg);
            u32x4 cq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            if (DROP) PA_DS128(cq[0], cbase, buf * BUF + AUX + 64 + kt * 64);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ko = kt * 32 + 8 * g + 4 * half;
                uint32_t m4 = 0;
                if (key_masked) m4 = *reinterpret_cast<const uint32_t*>(mk + ko);
                if (DROP) {
                    if (g + 1 < 4) {
                        PA_DS128(cq[(g + 1) & 1], cbase, buf * BUF + AUX + 64 + kt * 64 + (g + 1) * 16);
                        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cq[g & 1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[g & 1]));
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(__builtin_fmaf(sacc[4 * g + e], sl, nl));
                    if (need_mask) {
                        const int key = k0 + ko + e;
                        const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && key > qrow);
                        pe = masked ? 0.f : pe;
                    }
                    float dp = dpacc[4 * g + e];
                    if (DROP) dp = drop_keep2(arow, cq[g & 1][e], p.drop_thr) ? dp : 0.f;
                    sacc[4 * g + e] = pe * (dp - dlt);                      // dS^T / (scale / (1-p))
                }
            }
            mma_tr3<DH, buf * BUF + kt * 32 * B::RBN>(dqacc, lb, sacc);      // dQ^T += K^T dS^T
        };
        sub(IC<0>{});
        sub(IC<1>{});
        tile_barrier();
    };
    for (int step = 0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    bf16* dQp = reinterpret_cast<bf16*>(p.dq) + (size_t)qoff * p.lddq + h * DH;
    store_rows<bf16, DH>(dQp, p.lddq, qrow, p.Lq, dqacc, p.scale * (DROP ? p.drop_scale : 1.0f), lane);
}

// (CAUSAL as a template parameter: see attn4_bwd_dkv_kernel.  Not as two copies of the score loop behind a lambda: the
// streamed lse / delta words are written by asynchronous ds_reads into registers the compiler must not move before the wait,
// and by-reference captures let it.)
template <int DH, bool DROP, int OCC, bool CAUSAL>
__global__ __launch_bounds__(NTH, OCC) void attn_bwd_dkv_bf16_kernel(AttnP pin) {
    using A = AT<bf16, DH>;
    using B = BT<DH>;
    constexpr int BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, h, b, off_ = 0, len_ = -1;
    if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_k, tile_, h, b, off_, len_)) return; }
    else { decode_block((pin.Lk + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int key0 = tile_ * BOWN;
    if (key0 >= p.Lk) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const bf16* dOp = reinterpret_cast<const bf16*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const int krow = key0 + wave * 32 + (lane & 31);
    const bool kmasked = (krow >= p.Lk) || (p.kpm && p.kpm[(size_t)b * pin.Lk + krow]);

    u32x4 kreg[A::NS], vreg[A::NS];
    load_row_regs<bf16, DH>(kreg, Kp, p.ldk, krow, p.Lk, lane);
    load_row_regs<bf16, DH>(vreg, Vp, p.ldv, krow, p.Lk, lane);
    const int nsteps = (p.Lq + BSTR - 1) / BSTR;
    const int step0 = CAUSAL ? (key0 / BSTR) : 0;

    f32x16 dkacc[A::NDT], dvacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[dt][r] = 0.f; dvacc[dt][r] = 0.f; }
    const float sl = p.scale * LOG2E;
    const float keep_p = DROP ? 1.0f / p.drop_scale : 1.0f;
    const uint32_t ckey = DROP ? drop_key_hash(p.drop_seed, (uint32_t)krow) : 0u;
    const size_t srow0 = ((size_t)b * p.H + h) * pin.Lq;

    const TileSrc srcQ = tile_src(Qp, p.ldq, p.Lq, DH), srcO = tile_src(dOp, p.lddo, p.Lq, DH);
    const int voffQ = tile_voff<DH>(p.ldq, tid), voffO = tile_voff<DH>(p.lddo, tid);
    LdsBase<DH> lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base, lane);
    const uint32_t abase = smem_base + half * 16;        // this half-wave's 4 query rows of an 8-row group (aux words)
    // (the per-row words are requested with the tile's DMA - buffer loads, scalar resource + the lane's word - and only scaled and
    //  written to LDS at the END of the step that overlaps them, as in attn4_dkv_body: arithmetic next to the loads made hipcc
    //  wait out a memory round trip in front of every step)
    const __amdgpu_buffer_rsrc_t rs_lse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.lse + srow0), 0, p.Lq * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_del = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.delta + srow0), 0, p.Lq * 4, 0x00020000);
    uint32_t lv_raw = 0u, dv_raw = 0u;
    auto issue = [&](int step, int buf) {
        char* base = smem + buf * BUF;
        const int r0 = step * BSTR;
        if (tid < BSTR) {
            lv_raw = __builtin_amdgcn_raw_buffer_load_b32(rs_lse, tid * 4, r0 * 4, 0);
            dv_raw = __builtin_amdgcn_raw_buffer_load_b32(rs_del, tid * 4, r0 * 4, 0);
        }
        glds_tile<DH>(base, srcQ, voffQ, r0, wave);
        glds_tile<DH>(base + B::NAT, srcO, voffO, r0, wave);
    };
    auto issue_aux = [&](int step, int buf) {
        if (tid < BSTR) {
            const int r = step * BSTR + tid;
            const bool in = r < p.Lq;
            float* aux = reinterpret_cast<float*>(smem + buf * BUF + AUX);
            aux[tid] = in ? __uint_as_float(lv_raw) * LOG2E : INFINITY;  // +inf -> p = 0 for rows past Lq
            aux[64 + tid] = in ? __uint_as_float(dv_raw) * keep_p : 0.f;  // delta * (1-p), see the dQ kernel
            if (DROP) reinterpret_cast<uint32_t*>(aux)[128 + tid] = drop_row_hash(p.drop_seed, (uint32_t)(srow0 + r));
        }
    };
    if (step0 < nsteps) { issue(step0, 0); issue_aux(step0, 0); }
    tile_barrier();

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const bool more = step + 1 < nsteps;
        if (more) issue(step + 1, buf ^ 1);
        const int r0 = step * BSTR;
        auto sub = [&](auto qtc) {
            constexpr int qt = decltype(qtc)::value;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            mma_nat3<DH, buf * BUF + qt * 32 * B::RBN>(sacc, lb, kreg);               // S[q][key]
            mma_nat3<DH, buf * BUF + B::NAT + qt * 32 * B::RBN>(dpacc, lb, vreg);     // dP[q][key]
            // A masked KEY (this lane's row) gets its accumulators zeroed at the end instead of per element; the causal
            // test only runs on the q tiles that straddle the diagonal (wave-uniform).
            const bool need_causal = CAUSAL && (key0 + wave * 32 + 31 > r0 + qt * 32);
            // per-query-row words of the tile (lse, delta, dropout row hash) stream through two register sets: the words of
            // group g + 1 are requested before group g is consumed
            u32x4 lq[2], dq_[2], aq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            constexpr int AO = buf * BUF + AUX + qt * 128;
            PA_DS128(lq[0], abase, AO); PA_DS128(dq_[0], abase, AO + 256);
            if (DROP) PA_DS128(aq[0], abase, AO + 512);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qo = qt * 32 + 8 * g + 4 * half;
                if (g + 1 < 4) {
                    PA_DS128(lq[(g + 1) & 1], abase, AO + (g + 1) * 32); PA_DS128(dq_[(g + 1) & 1], abase, AO + 256 + (g + 1) * 32);
                    if (DROP) {
                        PA_DS128(aq[(g + 1) & 1], abase, AO + 512 + (g + 1) * 32);
                        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(lq[g & 1]), "+v"(dq_[g & 1]), "+v"(aq[g & 1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(lq[g & 1]), "+v"(dq_[g & 1]));
                    }
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lq[g & 1]), "+v"(dq_[g & 1]), "+v"(aq[g & 1]));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(__builtin_fmaf(sacc[4 * g + e], sl, -__uint_as_float(lq[g & 1][e])));
                    if (CAUSAL && need_causal) pe = (krow > r0 + qo + e) ? 0.f : pe;
                    float pd = pe;
                    if (DROP) pd = drop_keep2(aq[g & 1][e], ckey, p.drop_thr) ? pe : 0.f;
                    sacc[4 * g + e] = pd;                                  // P_drop * (1-p)
                    // dS * (1-p) / scale = P (keep * dP - delta) = P_drop dP - P delta: one select instead of two
                    dpacc[4 * g + e] = __builtin_fmaf(pd, dpacc[4 * g + e], -(pe * __uint_as_float(dq_[g & 1][e])));
                }
            }
            mma_tr3<DH, buf * BUF + B::NAT + qt * 32 * B::RBN>(dvacc, lb, sacc);      // dV^T += dO^T P
            mma_tr3<DH, buf * BUF + qt * 32 * B::RBN>(dkacc, lb, dpacc);              // dK^T += Q^T dS
        };
        sub(IC<0>{});
        sub(IC<1>{});
        if (more) issue_aux(step + 1, buf ^ 1);
        tile_barrier();
    };
    for (int step = step0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    bf16* dKp = reinterpret_cast<bf16*>(p.dk) + (size_t)koff * p.lddk + h * DH;
    bf16* dVp = reinterpret_cast<bf16*>(p.dv) + (size_t)koff * p.lddv + h * DH;
    const float ds = DROP ? p.drop_scale : 1.0f;
    // masked key: select (not multiply) - its accumulators may hold inf/NaN from exponentials the softmax never saw
    store_rows<bf16, DH>(dKp, p.lddk, krow, p.Lk, dkacc, kmasked ? 0.f : p.scale * ds, lane, kmasked);
    store_rows<bf16, DH>(dVp, p.lddv, krow, p.Lk, dvacc, kmasked ? 0.f : ds, lane, kmasked);
}

// ---- bf16 kernels, v4 (dh = 64): 16 rows of the owned side per WAVE, 8 waves per block ----------------------------------
// Same algorithm, tiles, DMA, masks and dropout as v3; what changes is the MFMA shape: v_mfma_f32_16x16x32_bf16 gives a wave
// 16 query (resp. key) rows instead of 32, so the same 128-row block is 8 waves and a launch has TWICE the waves.  The
// packed encoder step is 7 924 rows x 8 heads = 1 981 32-row waves for 1 024 SIMDs - two per SIMD, where a lone wave pays
// ~6.5 cycles per VALU instruction and every LDS / MFMA latency in full; with 16-row waves the same launch keeps ~4 per
// SIMD and each wave's step is half as long (critical path of the longest sample).  Per-score VALU work and MFMA time are
// unchanged (16 x 16 x 32: 8 passes, two per 32 x 32 x 16); LDS operand traffic per score doubles (13 % -> ~30 % of the
// LDS cycles); registers per lane halve (<= 96 here), so three 512-thread blocks fit a CU.
//   layouts (lane l: i = l & 15, g = l >> 4):  A[i][8g .. 8g+7], B[8g .. 8g+7][i], D[4g + r][i] (r = 0..3)
//   S^T[key][q] = K Q^T   per 16-key block kb: A = K rows (LDS, chunk 4s + g), B = this lane's Q row (registers, s = 0, 1)
//   O^T[d][q] += V^T P^T  per 16-d block db and 32-key block kk: B = the lane's own P values - slot 8g + 4c + r <-> key
//                         32kk + 16c + 4g + r (kb = 2kk + c) - A = V^T through two ds_read_b64_tr_b16 with the same labelling
constexpr int NT4 = 512;
__device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
}
// LDS image of a 64-row tile for the 16-row-wave kernels: row r at r * 128, its 16-byte chunk c at position c ^ key4(r).
// key4(r) = 2 * ((r >> 1) & 3) (round 6).  The 32-row-wave kernels' key (r >> 1) & 7 is conflict-free for the ds_read_b128 fragment
// reads but 2-way conflicted for EVERY ds_read_b64_tr_b16 here: a 32-lane group of the transposing read covers rows 8 m .. 8 m + 7 x two
// neighbouring chunks, and rows r, r + 2 landed on the same bank quads (r >> 1 differs in bit 0 = the chunk pair's own bit).  Measured
// (profiles/r05_attn_pmc_in_step.txt): 26 % of the dK / dV kernel's LDS cycles were conflict cycles with the LDS 47 % busy.  With bits
// 1-2 of the row in bits 1-2 of the key both reads are conflict-free (tools: exhaustive search over the GF(2)-linear keys).
#ifdef PA_KEY4_OLD          // A/B build (round 5's key; tools: PLANK_HIP_LIB=<variant>)
__device__ __forceinline__ int key4(int row) { return (row >> 1) & 7; }
#else
__device__ __forceinline__ int key4(int row) { return ((row >> 1) & 3) << 1; }
#endif
__device__ __forceinline__ int swz4_off(int row, int chunk) { return row * 128 + (((chunk ^ key4(row)) & 7) << 4); }
struct Lds4 {
    uint32_t nat[2];          // natural rows: row (l & 15), chunk 4s + g               (+ kb * 16 * 128)
    uint32_t tr[4];           // transposing patch of d block db: rows 4g + (L >> 2)    (+ (32kk + 16c) * 128)
    __device__ __forceinline__ void init(uint32_t base, int lane) {
        const int i = lane & 15, g = lane >> 4;
#pragma unroll
        for (int s = 0; s < 2; ++s) nat[s] = base + swz4_off(i, 4 * s + g);
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const int r = 4 * g + (i >> 2), col = 16 * db + 4 * (i & 3);
            tr[db] = base + swz4_off(r, col >> 3) + (col & 7) * 2;
        }
    }
};
// sacc[kb] (16 x 16) = TILE rows 16kb + (l & 15) (A, contraction over dh = 64) x regs (B)
template <int OFF> __device__ __forceinline__ void mma_nat4(f32x4 (&acc)[4], const Lds4& lb, const u32x4 (&regs)[2]) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        u32x4 a[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            PA_DS128(a[2 * j], lb.nat[0], OFF + (2 * h2 + j) * 2048);
            PA_DS128(a[2 * j + 1], lb.nat[1], OFF + (2 * h2 + j) * 2048);
        }
        wait_lds(a);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            mma16(z, a[2 * j], regs[0]);
            mma16(z, a[2 * j + 1], regs[1]);
            acc[2 * h2 + j] = z;
        }
    }
}
// The same product started from a per-lane constant c0 instead of zero (all four values of a lane's 16 x 16 accumulator
// belong to ONE column = one query row): with the rows pre-multiplied by scale * log2(e) and c0 = -(the row's softmax
// reference point) the MFMA itself delivers the exponent, and the score loop needs no FMA (one VALU instruction per score
// less in a VALU-bound loop).  c4 = {c0, c0, c0, c0}.
template <int OFF> __device__ __forceinline__ void mma_nat4c(f32x4 (&acc)[4], const Lds4& lb, const u32x4 (&regs)[2], const f32x4& c4) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        u32x4 a[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            PA_DS128(a[2 * j], lb.nat[0], OFF + (2 * h2 + j) * 2048);
            PA_DS128(a[2 * j + 1], lb.nat[1], OFF + (2 * h2 + j) * 2048);
        }
        wait_lds(a);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 z = c4;
            mma16(z, a[2 * j], regs[0]);
            mma16(z, a[2 * j + 1], regs[1]);
            acc[2 * h2 + j] = z;
        }
    }
}
// ... and from a per-ROW constant (dK / dV kernel: a lane's four accumulator values are four consecutive query rows of the tile):
// c[kb] = the lane's four constants of 16-row block kb, loaded from the tile's aux words straight into the MFMA's C operand
template <int OFF> __device__ __forceinline__ void mma_nat4v(f32x4 (&acc)[4], const Lds4& lb, const u32x4 (&regs)[2], const u32x4 (&c)[4]) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        u32x4 a[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            PA_DS128(a[2 * j], lb.nat[0], OFF + (2 * h2 + j) * 2048);
            PA_DS128(a[2 * j + 1], lb.nat[1], OFF + (2 * h2 + j) * 2048);
        }
        wait_lds(a);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 z = *reinterpret_cast<const f32x4*>(&c[2 * h2 + j]);
            mma16(z, a[2 * j], regs[0]);
            mma16(z, a[2 * j + 1], regs[1]);
            acc[2 * h2 + j] = z;
        }
    }
}
// rows held as bf16 fragments *= f (f32 product, one rounding back to bf16)
__device__ __forceinline__ void scale_row4(u32x4 (&regs)[2], float f) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int w = 0; w < 4; ++w) regs[s_][w] = pack_bf16(bf16_lo(regs[s_][w]) * f, bf16_hi(regs[s_][w]) * f);
}
// acc[db] (16 d x 16) += TILE^T[d][64 rows] x P  (P = the lane's 16 values p[kb][r] <-> row 16kb + 4g + r).
// Split into the operand reads of one half (d blocks 2*H2, 2*H2 + 1) and its MFMAs so that a caller can request the
// reads BEFORE the arithmetic that produces P (the forward kernel: the V^T fragments travel while the softmax runs).
template <int OFF, int H2> __device__ __forceinline__ void tr4_reads(u32x4 (&a)[4], const Lds4& lb) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x2 x0, x1;
            PA_DSTR(x0, lb.tr[2 * H2 + j], OFF + (32 * kk) * 128);
            PA_DSTR(x1, lb.tr[2 * H2 + j], OFF + (32 * kk + 16) * 128);
            a[2 * j + kk][0] = x0[0]; a[2 * j + kk][1] = x0[1]; a[2 * j + kk][2] = x1[0]; a[2 * j + kk][3] = x1[1];
        }
}
__device__ __forceinline__ void pack_p4(u32x4 (&pb)[2], const f32x4 (&p)[4]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        pb[kk][0] = pack_bf16(p[2 * kk][0], p[2 * kk][1]); pb[kk][1] = pack_bf16(p[2 * kk][2], p[2 * kk][3]);
        pb[kk][2] = pack_bf16(p[2 * kk + 1][0], p[2 * kk + 1][1]); pb[kk][3] = pack_bf16(p[2 * kk + 1][2], p[2 * kk + 1][3]);
    }
}
template <int H2> __device__ __forceinline__ void tr4_mmas(f32x4 (&acc)[4], u32x4 (&a)[4], const u32x4 (&pb)[2]) {
    wait_lds(a);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        mma16(acc[2 * H2 + j], a[2 * j], pb[0]);
        mma16(acc[2 * H2 + j], a[2 * j + 1], pb[1]);
    }
}
template <int OFF> __device__ __forceinline__ void mma_tr4(f32x4 (&acc)[4], const Lds4& lb, const f32x4 (&p)[4]) {
    u32x4 pb[2], a[4];
    pack_p4(pb, p);
    tr4_reads<OFF, 0>(a, lb); tr4_mmas<0>(acc, a, pb);
    tr4_reads<OFF, 1>(a, lb); tr4_mmas<1>(acc, a, pb);
}
__device__ __forceinline__ void load_row4(u32x4 (&regs)[2], const bf16* base, int ld, int row, int nrows, int lane) {
    const int g = lane >> 4;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)row * ld + 32 * s + 8 * g);
        regs[s] = v;
    }
}
// lane (row = l & 15, g) holds acc[db][r] = X^T[d = 16db + 4g + r][row]: four consecutive d per db
__device__ __forceinline__ void store_rows4(bf16* base, int ld, int row, int nrows, const f32x4 (&acc)[4], float mul, int lane, bool zero = false) {
    if (row >= nrows) return;
    const int g = lane >> 4;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = zero ? 0.f : acc[db][e] * mul;
        st4<bf16>(base + (size_t)row * ld + 16 * db + 4 * g, o);
    }
}
__device__ __forceinline__ int tile_voff4(int ld, int tid) {
    const int row = tid >> 3;
    const int ch = ((tid & 7) ^ key4(row)) & 7;
    return (row * ld + ch * 8) * 2;
}
__device__ __forceinline__ void glds_tile4(char* lds, const TileSrc& ts, int voff, int row0, int wave) {
    const int skip = row0 * ts.ld * 2;                                        // scalar (see glds_tile)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ts.rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, voff + skip, 0, 0, 0);
}
__device__ __forceinline__ void scan_key_mask4(const uint8_t* mp, int Lk, int tid, int* s_scan, int& kfirst, int& klast) {
    int f = Lk, l = 0;
    for (int k = tid; k < Lk; k += NT4) {
        if (mp[k]) f = min(f, k);
        else l = k + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { f = min(f, __shfl_xor(f, o)); l = max(l, __shfl_xor(l, o)); }
    if ((tid & 63) == 0) { s_scan[(tid >> 6) * 2] = f; s_scan[(tid >> 6) * 2 + 1] = l; }
    __syncthreads();
    kfirst = Lk; klast = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) { kfirst = min(kfirst, s_scan[2 * w]); klast = max(klast, s_scan[2 * w + 1]); }
}
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, __shfl_xor(v, 16)); return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float quad_sum(float v) { v += __shfl_xor(v, 16); return v + __shfl_xor(v, 32); }

#ifdef PA_ATTN_TRACE
// debug build only (tools/attn_trace.py): per-block cycle stamps of the forward kernel
__device__ unsigned long long pa_attn_trace[8192 * 8];
#define PA_TR(i) do { if (tid == 0 && blockIdx.x < 8192) pa_attn_trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PA_TR(i) do { } while (0)
#endif
// KS = 2 (in-block key split): the block has 16 waves; waves 8..15 ("key half 1") own the same 128 query rows as waves
// 0..7 but walk the SECOND half of the element's key tiles, from their own pair of LDS stages; the two partial results
// (reference point m, row sum l, O^T) are merged through LDS at the end.  A block's serial chain of key steps - what the
// duration of a launch with few blocks is made of (cross-attention: one 128-row block per (sample, head), up to 16 steps) -
// is halved at the same number of resident waves.  Only launches without a key-padding mask and without causality
// (packed / cross attention: the key range is known before the loop).
template <bool DROP, int KS>
__global__ __launch_bounds__(NT4 * KS, 4) void attn4_fwd_kernel(AttnP pin) {
    constexpr int DH = 64, NAT = BT<DH>::NAT, BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = KS == 2 ? (wave_all >> 3) : 0;            // key half of this wave
    const int wave = wave_all & 7, tid = threadIdx.x & (NT4 - 1);     // position inside the half
    PA_TR(0);
    int tile_, h, b, off_ = 0, len_ = -1;
    if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_q, tile_, h, b, off_, len_)) return; }
    else { decode_block((pin.Lq + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int q0 = tile_ * BOWN;
    if (q0 >= p.Lq) return;
    PA_TR(1);
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qw0 = q0 + wave * 16, qrow = qw0 + (lane & 15);
    const bool wave_on = qw0 < p.Lq;                       // wave-uniform

    u32x4 qreg[2];
    load_row4(qreg, Qp, p.ldq, qrow, p.Lq, lane);
    const float sl = p.scale * LOG2E;
    scale_row4(qreg, sl);                                 // scores come out of the MFMA in log2 units (see mma_nat4c)
    const TileSrc srcK = tile_src(Kp, p.ldk, p.Lk, DH), srcV = tile_src(Vp, p.ldv, p.Lk, DH);
    const int voffK = tile_voff4(p.ldk, tid), voffV = tile_voff4(p.ldv, tid);
    Lds4 lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base + kh * 2 * BUF, lane);
    const uint32_t cbase = smem_base + kh * 2 * BUF + g * 16;   // this lane group's 4 keys of a 16-key block (aux words, natural order)
    // key tiles of this half: [kbase / 64, kbase / 64 + nst_h)
    const int ksteps = (p.Lk + BSTR - 1) / BSTR;
    // an element with few key tiles is not split: its second-half waves leave before the first barrier (their wave
    // slots and registers are free for other blocks at once; s_barrier only counts live waves)
    const bool split = KS == 2 && ksteps >= pin.ks_min;                    // block-uniform
    if (KS == 2 && kh == 1 && !split) return;
    const int s0 = split ? (ksteps + 1) / 2 : ksteps;
    const int kbase = kh * s0 * BSTR, nst_h = KS == 2 ? (kh ? ksteps - s0 : s0) : 0;
    auto issue = [&](int step, int buf, int kfirst_) {
        char* base = smem + kh * 2 * BUF + buf * BUF;
        const int k0 = kbase + step * BSTR;
        const bool tile_masked = k0 + BSTR > kfirst_;
        uint8_t mb = 0;
        if (tile_masked && tid < BSTR) {
            const int key = k0 + tid;
            mb = (key >= p.Lk) ? 1 : (mp ? mp[key] : 0);
        }
        glds_tile4(base, srcK, voffK, k0, wave);
        glds_tile4(base + NAT, srcV, voffV, k0, wave);
        if (tid < BSTR) {
            if (tile_masked) reinterpret_cast<uint8_t*>(base + AUX)[tid] = mb;
            if (DROP) reinterpret_cast<uint32_t*>(base + AUX + 64)[tid] = drop_key_hash(p.drop_seed, (uint32_t)(k0 + tid));
        }
    };
    if (KS == 1 || nst_h > 0) issue(0, 0, 0);
    int kfirst = p.Lk, klast = p.Lk;
    if (KS == 1 && mp) scan_key_mask4(mp, p.Lk, tid, reinterpret_cast<int*>(smem + 2 * BUF), kfirst, klast);
    int nsteps = (klast + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);
    const int my_steps = KS == 2 ? nst_h : nsteps;          // steps this wave computes; the block loops over the longer half
    if (KS == 2) nsteps = s0;

    f32x4 oacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow)) : 0u;
    tile_barrier();
    PA_TR(2);

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (step + 1 < my_steps) issue(step + 1, buf ^ 1, kfirst);
        if (!wave_on || step >= my_steps) { tile_barrier(); return; }      // rows past the element's end / the shorter key half: DMA + barriers only
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(smem + kh * 2 * BUF + buf * BUF + AUX);
        const int k0 = kbase + step * BSTR;
        // S^T - m_run: the accumulator starts at -(this row's reference point), so the exponent of every probability is
        // what the MFMA leaves behind (m_run = running maximum of the log2-domain scores, moved only when a row outgrows
        // it by 2^RESCALE_THR; -inf until the row has seen an unmasked key)
        const float nm = (m_run == -INFINITY) ? 0.f : -m_run;
        const f32x4 c4 = {nm, nm, nm, nm};
        f32x4 sacc[4];
        mma_nat4c<buf * BUF>(sacc, lb, qreg, c4);
        const bool key_masked = k0 + BSTR > kfirst;
        const bool need_mask = key_masked || (p.causal && (k0 + BSTR - 1 > qw0));           // wave-uniform
        float mx = -INFINITY;
        if (need_mask) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int ko = 16 * kb + 4 * g;
                const uint32_t m4 = key_masked ? *reinterpret_cast<const uint32_t*>(mk + ko) : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && k0 + ko + e > qrow);
                    const float x = masked ? -INFINITY : sacc[kb][e];
                    sacc[kb][e] = x;
                }
            }
        } else {
            settle_mfma(sacc[0], sacc[1], sacc[2], sacc[3]);   // interior tile (see the 32-row-wave kernel)
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) mx = max3f(max3f(mx, sacc[kb][0], sacc[kb][1]), sacc[kb][2], sacc[kb][3]);
        // deferred rescale: the reference point moves when some row outgrew it by 2^RESCALE_THR (always on a row's first
        // keys).  The test needs no cross-lane traffic: the four lanes of a row share m_run, so "some LANE's 16 scores
        // outgrew it" is the same wave-wide condition; the row maximum itself (two lane exchanges + their LDS round trips
        // in front of every exponential) is only formed on the steps that do move the reference point.
        const float thr = (m_run == -INFINITY) ? -INFINITY : RESCALE_THR;
        if (__any(mx > thr)) {
            mx = quad_max(mx);                                 // relative to the reference point
            const float m_new = fmaxf(m_run, mx - nm);
            const float ms = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - ms);
            const float shift = ms + nm;                       // new reference - old reference (>= 0)
            l_run *= alpha;
#pragma unroll
            for (int db = 0; db < 4; ++db) oacc[db] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) sacc[kb] -= shift;
            m_run = m_new;
        }
        float lsum[4] = {0.f, 0.f, 0.f, 0.f};
        // (Requesting the V^T fragments here, before the exponentials, so that their LDS round trip runs under the softmax
        // arithmetic, was measured: 33.6 vs 33.4 us on the packed encoder shape, 20.3 vs 20.8 us cross-attention - nothing.
        // With four waves per SIMD the other waves already cover it; profiles/r03_attention_experiments.txt.)
        u32x4 cq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (DROP) PA_DS128(cq[0], cbase, buf * BUF + AUX + 64);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (DROP) {
                if (kb + 1 < 4) {
                    PA_DS128(cq[(kb + 1) & 1], cbase, buf * BUF + AUX + 64 + (kb + 1) * 64);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cq[kb & 1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[kb & 1]));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = fast_exp2(sacc[kb][e]);
                lsum[e] += pe;
                if (DROP) pe = drop_keep2(arow, cq[kb & 1][e], p.drop_thr) ? pe : 0.f;
                sacc[kb][e] = pe;
            }
        }
        l_run += (lsum[0] + lsum[1]) + (lsum[2] + lsum[3]);
        {
            u32x4 pb[2], va1[4];
            pack_p4(pb, sacc);
            tr4_reads<buf * BUF + NAT, 0>(va1, lb); tr4_mmas<0>(oacc, va1, pb);
            tr4_reads<buf * BUF + NAT, 1>(va1, lb); tr4_mmas<1>(oacc, va1, pb);
        }
        tile_barrier();
    };
    for (int step = 0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    PA_TR(3);
    float l_tot = quad_sum(l_run);
    if (KS == 2 && split) {
        // merge the two key halves (every stage is free after the loop's last barrier): half 1 publishes (O^T, m, l) of its 16
        // rows per wave, half 0 rescales both to the common reference point and stores.  [db][lane] f32x4: conflict-free.
        f32x4* xo = reinterpret_cast<f32x4*>(smem) + wave * 4 * 64;
        float* xml = reinterpret_cast<float*>(smem + 8 * 4 * 64 * 16) + wave * 2 * 64;
        if (kh == 1) {
#pragma unroll
            for (int db = 0; db < 4; ++db) xo[db * 64 + lane] = oacc[db];
            xml[lane] = m_run; xml[64 + lane] = l_tot;
        }
        __syncthreads();
        if (kh == 1) return;
        const float m1 = xml[lane], l1 = xml[64 + lane];
        const float m_new = fmaxf(m_run, m1);
        const float ms = (m_new == -INFINITY) ? 0.f : m_new;
        const float a0 = fast_exp2(m_run - ms), a1 = fast_exp2(m1 - ms);
        l_tot = l_tot * a0 + l1 * a1;
#pragma unroll
        for (int db = 0; db < 4; ++db) oacc[db] = oacc[db] * a0 + xo[db * 64 + lane] * a1;
        m_run = m_new;
    }
    const float inv = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
    bf16* Op = reinterpret_cast<bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
    store_rows4(Op, p.ldo, qrow, p.Lq, oacc, inv, lane);
    if (g == 0 && qrow < p.Lq && p.lse)
        p.lse[((size_t)b * p.H + h) * pin.Lq + qrow] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : 0.f;
#ifdef PA_ATTN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PA_TR(4);
    if (tid == 0 && blockIdx.x < 8192) { pa_attn_trace[blockIdx.x * 8 + 5] = nsteps; pa_attn_trace[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }
#endif
}

template <bool DROP, int KS>          // KS = 2: in-block key split, see attn4_fwd_kernel (the partial dQ^T of the two halves are summed)
__device__ __forceinline__ void attn4_dq_body(const AttnP& pin, int bid, char* smem) {
    constexpr int DH = 64, NAT = BT<DH>::NAT, BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    const int lane = threadIdx.x & 63, g = lane >> 4;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kh = KS == 2 ? (wave_all >> 3) : 0;
    const int wave = wave_all & 7, tid = threadIdx.x & (NT4 - 1);
    int tile_, h, b, off_ = 0, len_ = -1;
    int part = 0, nparts = 1, slot = 0;                      // balanced == 2 (KS == 1 launches): this block's range of the key tiles
    if (KS == 1 && pin.balanced == 2) {
        SplitUnit su;
        if (!decode_unit_split(pin, pin.cu_q, su, bid)) return;
        tile_ = su.tile; h = su.h; b = su.b; off_ = su.off; len_ = su.len; part = su.part; nparts = su.nparts; slot = su.slot;
    }
    else if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_q, tile_, h, b, off_, len_, bid)) return; }
    else { decode_block((pin.Lq + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b, bid); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int q0 = tile_ * BOWN;
    if (q0 >= p.Lq) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const bf16* dOp = reinterpret_cast<const bf16*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qw0 = q0 + wave * 16, qrow = qw0 + (lane & 15);
    const bool wave_on = qw0 < p.Lq;                       // wave-uniform

    u32x4 qreg[2], doreg[2];
    load_row4(qreg, Qp, p.ldq, qrow, p.Lq, lane);
    load_row4(doreg, dOp, p.lddo, qrow, p.Lq, lane);
    // Instruction diet (round 6; the loop is bound by instruction issue, profiles/r04_attn_issue_bound.txt): the query rows are
    // pre-multiplied by scale * log2(e) - the forward kernels' own rounding, bf16(q * scale * log2 e) - and both score products start
    // from a per-row constant (all four values of a lane's 16 x 16 accumulator belong to ONE query row): S^T from -lse, dP^T from
    // -delta.  The MFMAs then deliver the exponent and dP - delta: no FMA and no subtraction per score (7.5 -> 5.5 VALU per score).
    scale_row4(qreg, p.scale * LOG2E);
    const size_t srow = ((size_t)b * p.H + h) * pin.Lq + qrow;
    const float lse2 = (qrow < p.Lq) ? p.lse[srow] * LOG2E : INFINITY;
    const float keep_p = DROP ? 1.0f / p.drop_scale : 1.0f;
    float dsum = 0.f;
    {
        const bf16* Op = reinterpret_cast<const bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
        u32x4 oreg[2];
        load_row4(oreg, Op, p.ldo, qrow, p.Lq, lane);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int w = 0; w < 4; ++w)
                dsum += bf16_lo(oreg[s_][w]) * bf16_lo(doreg[s_][w]) + bf16_hi(oreg[s_][w]) * bf16_hi(doreg[s_][w]);
        dsum = quad_sum(dsum);
        if (g == 0 && qrow < p.Lq && kh == 0 && part == 0) p.delta[srow] = dsum;
    }
    const float dlt = (qrow < p.Lq) ? dsum * keep_p : 0.f;
    const TileSrc srcK = tile_src(Kp, p.ldk, p.Lk, DH), srcV = tile_src(Vp, p.ldv, p.Lk, DH);
    const int voffK = tile_voff4(p.ldk, tid), voffV = tile_voff4(p.ldv, tid);
    Lds4 lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base + kh * 2 * BUF, lane);
    const uint32_t cbase = smem_base + kh * 2 * BUF + g * 16;
    const int ksteps = (p.Lk + BSTR - 1) / BSTR;
    // an element with few key tiles is not split: its second-half waves leave before the first barrier (their wave
    // slots and registers are free for other blocks at once; s_barrier only counts live waves)
    const bool split = KS == 2 && ksteps >= pin.ks_min;                    // block-uniform
    if (KS == 2 && kh == 1 && !split) return;
    const int s0 = split ? (ksteps + 1) / 2 : ksteps;
    int r_lo = 0, r_hi = ksteps;
    if (KS == 1 && nparts > 1) split_range(ksteps, part, nparts, r_lo, r_hi);
    const int kbase = KS == 2 ? kh * s0 * BSTR : r_lo * BSTR, nst_h = KS == 2 ? (kh ? ksteps - s0 : s0) : 0;
    auto issue = [&](int step, int buf, int kfirst_) {
        char* base = smem + kh * 2 * BUF + buf * BUF;
        const int k0 = kbase + step * BSTR;
        const bool tile_masked = k0 + BSTR > kfirst_;
        uint8_t mb = 0;
        if (tile_masked && tid < BSTR) {
            const int key = k0 + tid;
            mb = (key >= p.Lk) ? 1 : (mp ? mp[key] : 0);
        }
        glds_tile4(base, srcK, voffK, k0, wave);
        glds_tile4(base + NAT, srcV, voffV, k0, wave);
        if (tid < BSTR) {
            if (tile_masked) reinterpret_cast<uint8_t*>(base + AUX)[tid] = mb;
            if (DROP) reinterpret_cast<uint32_t*>(base + AUX + 64)[tid] = drop_key_hash(p.drop_seed, (uint32_t)(k0 + tid));
        }
    };
    if (KS == 1 ? r_hi > r_lo : nst_h > 0) issue(0, 0, 0);
    int kfirst = p.Lk, klast = p.Lk;
    if (KS == 1 && mp) scan_key_mask4(mp, p.Lk, tid, reinterpret_cast<int*>(smem + 2 * BUF), kfirst, klast);
    int nsteps = (klast + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);
    if (KS == 1) nsteps = max(0, min(nsteps, r_hi) - r_lo);          // steps of this block's range (all of them when unsplit)
    const int my_steps = KS == 2 ? nst_h : nsteps;
    if (KS == 2) nsteps = s0;

    f32x4 dqacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) dqacc[db] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float nl = -lse2, ndlt = -dlt;
    const f32x4 c_nl = {nl, nl, nl, nl}, c_ndlt = {ndlt, ndlt, ndlt, ndlt};
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)srow) : 0u;
    tile_barrier();

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        if (step + 1 < my_steps) issue(step + 1, buf ^ 1, kfirst);
        if (!wave_on || step >= my_steps) { tile_barrier(); return; }      // rows past the element's end / the shorter key half: DMA + barriers only
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(smem + kh * 2 * BUF + buf * BUF + AUX);
        const int k0 = kbase + step * BSTR;
        const bool key_masked = k0 + BSTR > kfirst;
        const bool need_mask = key_masked || (p.causal && (k0 + BSTR - 1 > qw0));
        f32x4 sacc[4], dpacc[4];
        mma_nat4c<buf * BUF>(sacc, lb, qreg, c_nl);                    // log2-domain scores - lse: the exponent
        mma_nat4c<buf * BUF + NAT>(dpacc, lb, doreg, c_ndlt);          // dP - delta
        u32x4 cq[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (DROP) PA_DS128(cq[0], cbase, buf * BUF + AUX + 64);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int ko = 16 * kb + 4 * g;
            uint32_t m4 = 0;
            if (key_masked) m4 = *reinterpret_cast<const uint32_t*>(mk + ko);
            if (DROP) {
                if (kb + 1 < 4) {
                    PA_DS128(cq[(kb + 1) & 1], cbase, buf * BUF + AUX + 64 + (kb + 1) * 64);
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(cq[kb & 1]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cq[kb & 1]));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = fast_exp2(sacc[kb][e]);
                if (need_mask) {
                    const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && k0 + ko + e > qrow);
                    pe = masked ? 0.f : pe;
                }
                float dp = dpacc[kb][e];                               // dP - delta;  a dropped probability: 0 - delta
                if (DROP) dp = drop_keep2(arow, cq[kb & 1][e], p.drop_thr) ? dp : ndlt;
                sacc[kb][e] = pe * dp;                                 // dS^T / (scale / (1-p))
            }
        }
        mma_tr4<buf * BUF>(dqacc, lb, sacc);                           // dQ^T += K^T dS^T
        tile_barrier();
    };
    for (int step = 0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    if (KS == 2 && split) {
        f32x4* xo = reinterpret_cast<f32x4*>(smem) + wave * 4 * 64;
        if (kh == 1) {
#pragma unroll
            for (int db = 0; db < 4; ++db) xo[db * 64 + lane] = dqacc[db];
        }
        __syncthreads();
        if (kh == 1) return;
#pragma unroll
        for (int db = 0; db < 4; ++db) dqacc[db] += xo[db * 64 + lane];
    }
    if (KS == 1 && nparts > 1) {
        // range block: publish the partial dQ^T; the tile's last range block to arrive adds the others to its own and stores
        char* const pbase = pin.sp_part + (size_t)slot * pin.sp_pmax * SP_BYTES;
        const SplitOut so(pbase + (size_t)part * SP_BYTES, SP_BYTES);
#pragma unroll
        for (int db = 0; db < 4; ++db) so.put(db, NT4, dqacc[db]);
        if (!split_arrive(pin.sp_tick + slot, nparts, reinterpret_cast<int*>(smem + 2 * BUF))) return;
        for (int pp = 0; pp < nparts; ++pp) {
            if (pp == part) continue;
#pragma unroll
            for (int db = 0; db < 4; ++db) dqacc[db] += split_get(pbase + (size_t)pp * SP_BYTES, db, NT4, SP_BYTES);
        }
    }
    bf16* dQp = reinterpret_cast<bf16*>(p.dq) + (size_t)qoff * p.lddq + h * DH;
    store_rows4(dQp, p.lddq, qrow, p.Lq, dqacc, p.scale * (DROP ? p.drop_scale : 1.0f), lane);
}

template <bool DROP, int KS>
__global__ __launch_bounds__(NT4 * KS, 4) void attn4_bwd_dq_kernel(AttnP pin) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    attn4_dq_body<DROP, KS>(pin, blockIdx.x, smem);
}

// CAUSAL is a template parameter here: as a run-time flag the per-score causal select (index, compare, select) was executed
// on every tile of the non-causal encoder / cross attention, and two copies of the score loop in one kernel spill at 128 VGPRs.
// SELF_DELTA: the block computes delta[q] = sum_d dO[q][d] O[q][d] of its element's query rows itself (into LDS, Lq <= MERGE_MAX_LQ)
// instead of reading what the dQ kernel published - what lets the dQ and dK/dV blocks of a single-query-tile attention run
// in ONE launch (attn4_bwd_merged_kernel).
template <bool DROP, bool CAUSAL, bool SELF_DELTA>
__device__ __forceinline__ void attn4_dkv_body(const AttnP& pin, int bid, char* smem) {
    constexpr int DH = 64, NAT = BT<DH>::NAT, BUF = BL<DH>::BUF, AUX = BL<DH>::AUX;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, h, b, off_ = 0, len_ = -1;
    int part = 0, nparts = 1, slot = 0;                      // balanced == 2: this block's range of the query tiles
    if (!SELF_DELTA && pin.balanced == 2) {
        SplitUnit su;
        if (!decode_unit_split(pin, pin.cu_k, su, bid)) return;
        tile_ = su.tile; h = su.h; b = su.b; off_ = su.off; len_ = su.len; part = su.part; nparts = su.nparts; slot = su.slot;
    }
    else if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_k, tile_, h, b, off_, len_, bid)) return; }
    else { decode_block((pin.Lk + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b, bid); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int key0 = tile_ * BOWN;
    if (key0 >= p.Lk) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const bf16* dOp = reinterpret_cast<const bf16*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const int kw0 = key0 + wave * 16, krow = kw0 + (lane & 15);
    const bool wave_on = kw0 < p.Lk;                       // wave-uniform
    const bool kmasked = (krow >= p.Lk) || (p.kpm && p.kpm[(size_t)b * pin.Lk + krow]);

    u32x4 kreg[2], vreg[2];
    load_row4(kreg, Kp, p.ldk, krow, p.Lk, lane);
    load_row4(vreg, Vp, p.ldv, krow, p.Lk, lane);
    // Instruction diet (round 6, as attn4_dq_body): the key rows are pre-multiplied by scale * log2(e) and both score products start
    // from the tile's per-query-row aux words, stored NEGATED - S from -lse, dP from -delta: the MFMAs deliver the exponent and
    // dP - delta (8 -> 6 VALU per score with dropout, 4 -> 2 without).
    scale_row4(kreg, p.scale * LOG2E);
    int nsteps = (p.Lq + BSTR - 1) / BSTR;
    int step0 = CAUSAL ? (key0 / BSTR) : 0;
    if (nparts > 1) split_range(nsteps, part, nparts, step0, nsteps);

    f32x4 dkacc[4], dvacc[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) { dkacc[db] = f32x4{0.f, 0.f, 0.f, 0.f}; dvacc[db] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const float sl = p.scale * LOG2E;
    const float keep_p = DROP ? 1.0f / p.drop_scale : 1.0f;
    const uint32_t ckey = DROP ? drop_key_hash(p.drop_seed, (uint32_t)krow) : 0u;
    const size_t srow0 = ((size_t)b * p.H + h) * pin.Lq;
    const TileSrc srcQ = tile_src(Qp, p.ldq, p.Lq, DH), srcO = tile_src(dOp, p.lddo, p.Lq, DH);
    const int voffQ = tile_voff4(p.ldq, tid), voffO = tile_voff4(p.lddo, tid);
    Lds4 lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base, lane);
    const uint32_t abase = smem_base + g * 16;            // this lane group's 4 query rows of a 16-row block (aux words)
    float* sdelta = reinterpret_cast<float*>(smem + 2 * BUF + 64);       // [Lq] (SELF_DELTA; the merged launch allocates it)
    // A tile's per-row words (lse, delta) are REQUESTED with the tile's DMA (issue) and only touched - scaled, negated, written to
    // LDS - at the END of the step that overlaps them (issue_aux, in front of the step's barrier).  Round 6: with the arithmetic next
    // to the loads hipcc waited for them (`s_waitcnt vmcnt(0)`) before the first wave had even issued its share of the DMA - a
    // memory round trip in front of every step (tools/isa_loop_waits.py).
    // (buffer loads: the element's lse / delta rows as a scalar resource + the lane's word offset - no 64-bit lane addresses to
    //  keep in registers; the kernel sits at its 128-register budget.  Rows past Lq read 0 and are replaced in issue_aux.)
    const __amdgpu_buffer_rsrc_t rs_lse = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.lse + srow0), 0, p.Lq * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_del = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.delta + srow0), 0, SELF_DELTA ? 0 : p.Lq * 4, 0x00020000);
    uint32_t lv_raw = 0u, dv_raw = 0u;
    auto issue = [&](int step, int buf) {
        char* base = smem + buf * BUF;
        const int r0 = step * BSTR;
        if (tid < BSTR) {
            lv_raw = __builtin_amdgcn_raw_buffer_load_b32(rs_lse, tid * 4, r0 * 4, 0);
            if (!SELF_DELTA) dv_raw = __builtin_amdgcn_raw_buffer_load_b32(rs_del, tid * 4, r0 * 4, 0);
        }
        glds_tile4(base, srcQ, voffQ, r0, wave);
        glds_tile4(base + NAT, srcO, voffO, r0, wave);
    };
    auto issue_aux = [&](int step, int buf) {
        if (tid < BSTR) {
            const int r = step * BSTR + tid;
            const bool in = r < p.Lq;
            float* aux = reinterpret_cast<float*>(smem + buf * BUF + AUX);
            const float dlt = SELF_DELTA ? sdelta[min(r, p.Lq - 1)] : __uint_as_float(dv_raw);
            aux[tid] = in ? -(__uint_as_float(lv_raw) * LOG2E) : -INFINITY;      // (negated: accumulator start values of the score products)
            aux[64 + tid] = in ? -(dlt * keep_p) : 0.f;
            if (DROP) reinterpret_cast<uint32_t*>(aux)[128 + tid] = drop_row_hash(p.drop_seed, (uint32_t)(srow0 + r));
        }
    };
    if constexpr (SELF_DELTA) {
        const bf16* Op = reinterpret_cast<const bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
        const int ch = tid & 7;                              // eight lanes per row, 8 elements each
        // four rows per thread in flight (unconditional clamped loads, surplus rows discarded): the element's Lq rows cost
        // Lq / 256 memory round trips per block
        const int r_first = CAUSAL ? step0 * BSTR : 0;       // causal: query rows before this block's keys are never visited
        for (int r0_ = r_first + (tid >> 3); r0_ < p.Lq; r0_ += 4 * (NT4 / 8)) {
            u32x4 o4[4], d4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = min(r0_ + j * (NT4 / 8), p.Lq - 1);
                o4[j] = *reinterpret_cast<const u32x4*>(Op + (size_t)r * p.ldo + ch * 8);
                d4[j] = *reinterpret_cast<const u32x4*>(dOp + (size_t)r * p.lddo + ch * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) a += bf16_lo(o4[j][w]) * bf16_lo(d4[j][w]) + bf16_hi(o4[j][w]) * bf16_hi(d4[j][w]);
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
                const int r = r0_ + j * (NT4 / 8);
                if (ch == 0 && r < p.Lq) sdelta[r] = a;
            }
        }
        __syncthreads();
    }
    if (step0 < nsteps) { issue(step0, 0); issue_aux(step0, 0); }
    tile_barrier();

    auto body = [&](int step, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const bool more = step + 1 < nsteps;
        if (more) issue(step + 1, buf ^ 1);
        if (!wave_on) { if (more) issue_aux(step + 1, buf ^ 1); tile_barrier(); return; }      // this wave's 16 keys lie past the element's last key
        const int r0 = step * BSTR;
        constexpr int AO = buf * BUF + AUX;
        u32x4 nl4[4], nd4[4];                                          // -lse * log2 e and -delta * (1 - p) of this lane's query rows
        // (the two sets one after the other: requested together, the 32 start values pushed the kernel past 128 registers)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) PA_DS128(nl4[qb], abase, AO + qb * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nl4[0]), "+v"(nl4[1]), "+v"(nl4[2]), "+v"(nl4[3]));
        f32x4 sacc[4], dpacc[4];
        mma_nat4v<buf * BUF>(sacc, lb, kreg, nl4);                     // S[q][key] - lse[q] (log2 domain): rows q = 16qb + 4g + r, col key = l & 15
        __builtin_amdgcn_sched_barrier(0);                             // (keeps hipcc from hoisting the second set above the first product)
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) PA_DS128(nd4[qb], abase, AO + 256 + qb * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nd4[0]), "+v"(nd4[1]), "+v"(nd4[2]), "+v"(nd4[3]));
        mma_nat4v<buf * BUF + NAT>(dpacc, lb, vreg, nd4);              // dP[q][key] - delta[q]
        const bool need_causal = CAUSAL && (kw0 + 15 > r0);
#pragma unroll
        for (int qb = 0; qb < 4; ++qb) {
            const int qo = 16 * qb + 4 * g;
            u32x4 aq = {0u, 0u, 0u, 0u};
            if (DROP) { PA_DS128(aq, abase, AO + 512 + qb * 64); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aq)); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = fast_exp2(sacc[qb][e]);
                if (CAUSAL && need_causal) pe = (krow > r0 + qo + e) ? 0.f : pe;
                float pd = pe, x = dpacc[qb][e];
                if (DROP) {
                    const bool keep = drop_keep2(aq[e], ckey, p.drop_thr);
                    pd = keep ? pe : 0.f;                                           // P_drop * (1-p)
                    x = keep ? x : __uint_as_float(nd4[qb][e]);                     // a dropped probability: 0 - delta
                }
                sacc[qb][e] = pd;
                dpacc[qb][e] = pe * x;                                              // dS * (1-p) / scale = P (keep * dP - delta)
            }
        }
        mma_tr4<buf * BUF + NAT>(dvacc, lb, sacc);                     // dV^T += dO^T P
        mma_tr4<buf * BUF>(dkacc, lb, dpacc);                          // dK^T += Q^T dS
        if (more) issue_aux(step + 1, buf ^ 1);
        tile_barrier();
    };
    for (int step = step0; step < nsteps; step += 2) {
        body(step, IC<0>{});
        if (step + 1 < nsteps) body(step + 1, IC<1>{});
    }
    if (nparts > 1) {
        // range block: publish the partial dK^T / dV^T; the tile's last range block to arrive adds the others to its own and stores
        char* const pbase = pin.sp_part + (size_t)slot * pin.sp_pmax * SP_BYTES;
        const SplitOut so(pbase + (size_t)part * SP_BYTES, SP_BYTES);
#pragma unroll
        for (int db = 0; db < 4; ++db) { so.put(db, NT4, dkacc[db]); so.put(4 + db, NT4, dvacc[db]); }
        if (!split_arrive(pin.sp_tick + slot, nparts, reinterpret_cast<int*>(smem + 2 * BUF))) return;
        for (int pp = 0; pp < nparts; ++pp) {
            if (pp == part) continue;
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                dkacc[db] += split_get(pbase + (size_t)pp * SP_BYTES, db, NT4, SP_BYTES);
                dvacc[db] += split_get(pbase + (size_t)pp * SP_BYTES, 4 + db, NT4, SP_BYTES);
            }
        }
    }
    bf16* dKp = reinterpret_cast<bf16*>(p.dk) + (size_t)koff * p.lddk + h * DH;
    bf16* dVp = reinterpret_cast<bf16*>(p.dv) + (size_t)koff * p.lddv + h * DH;
    const float ds = DROP ? p.drop_scale : 1.0f;
    store_rows4(dKp, p.lddk, krow, p.Lk, dkacc, kmasked ? 0.f : p.scale * ds, lane, kmasked);
    store_rows4(dVp, p.lddv, krow, p.Lk, dvacc, kmasked ? 0.f : ds, lane, kmasked);
}

template <bool DROP, bool CAUSAL>
__global__ __launch_bounds__(NT4, 4) void attn4_bwd_dkv_kernel(AttnP pin) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    attn4_dkv_body<DROP, CAUSAL, false>(pin, blockIdx.x, smem);
}
// dQ blocks [0, gq) and dK/dV blocks [gq, ...) of one attention backward in ONE launch, for launches whose queries are a
// single 128-row tile per (sample, head) - decoder self-attention and cross-attention: a kernel boundary less per layer, and
// the 128 dQ blocks (half of the CUs) run beside the dK/dV blocks instead of ahead of them.
template <bool DROP, bool CAUSAL>
__global__ __launch_bounds__(NT4, 4) void attn4_bwd_merged_kernel(AttnP pin, int gq) {
    extern __shared__ __attribute__((aligned(256))) char smem[];
    if ((int)blockIdx.x < gq) attn4_dq_body<DROP, 1>(pin, blockIdx.x, smem);
    else attn4_dkv_body<DROP, CAUSAL, true>(pin, (int)blockIdx.x - gq, smem);
}

#include "attention5.h"
#include "attention_x3.h"

// =====================================================================================================
static int split_kmax() { static const int v = getenv("PA_ATTN_SPLIT_KMAX") ? atoi(getenv("PA_ATTN_SPLIT_KMAX")) : 8; return v < 1 ? 1 : v; }
static int split_pmax() { static const int v = getenv("PA_ATTN_SPLIT_PMAX") ? atoi(getenv("PA_ATTN_SPLIT_PMAX")) : 2; return v < 1 ? 1 : v > 8 ? 8 : v; }
AttnP make_params(const pa_attn_args* a) {
    AttnP p;
    p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o; p.lse = a->lse; p.kpm = a->kpm;
    p.B = a->B; p.H = a->H; p.Lq = a->Lq; p.Lk = a->Lk;
    p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
    p.causal = a->causal; p.scale = a->scale;
    p.drop_thr = (uint32_t)((double)a->drop_p * 4294967296.0);     // keep <=> 32-bit product >= thr (pa_device.h drop_keep2)
    p.drop_scale = (float)(1.0 / (1.0 - (double)p.drop_thr / 4294967296.0));
    p.drop_seed = a->drop_seed;
    p.dout = a->dout; p.dq = a->dq; p.dk = a->dk; p.dv = a->dv; p.delta = a->delta;
    p.lddo = a->lddo; p.lddq = a->lddq; p.lddk = a->lddk; p.lddv = a->lddv;
    p.cu_q = a->cu_q; p.cu_k = a->cu_k; p.order = a->order;
    static const bool bal_env = !(getenv("PA_ATTN_BALANCED") && atoi(getenv("PA_ATTN_BALANCED")) == 0);
    p.ks_min = 4; p.parts_q = 1; p.parts_kv = 1;
    p.balanced = (bal_env && a->order && a->cu_q && a->cu_k && a->H == 8 && a->B <= 64) ? 1 : 0;
    // range blocks (balanced == 2, set by the launchers whose kernels know it): packed self-attention with scratch from the caller.
    // PA_ATTN_SPLIT=1 enables (default off); PA_ATTN_SPLIT_KMAX (8) longest unsplit chain in 64-row tiles; PA_ATTN_SPLIT_PMAX (2) most ranges.
    p.sp_tick = nullptr; p.sp_part = nullptr; p.sp_slots = 0; p.sp_pmax = split_pmax(); p.sp_kmax = split_kmax();
    static const bool sp_env = getenv("PA_ATTN_SPLIT") && atoi(getenv("PA_ATTN_SPLIT")) != 0;     // opt-in: measured slower (profiles/r06_attention_launch_shape.txt)
    if (sp_env && p.balanced && a->ws && a->cu_q == a->cu_k && !a->kpm && !a->causal && a->dtype == PA_BF16 && a->dh == 64 &&
        (reinterpret_cast<uintptr_t>(a->ws) & 255) == 0 && p.sp_pmax > 1) {
        const int64_t per = (int64_t)p.sp_pmax * SP_BYTES + 64;          // one (owned tile, head): its range blocks' partials + ticket
        const int64_t total = a->ws_bytes / per;
        p.sp_slots = (int)(total / 8 < (1 << 20) ? total / 8 : (1 << 20));
        p.sp_tick = static_cast<int*>(a->ws);
        p.sp_part = static_cast<char*>(a->ws) + ((int64_t)p.sp_slots * 8 * 4 + 255) / 256 * 256;
        if ((int64_t)p.sp_slots * 8 * per > a->ws_bytes || p.sp_slots < 1) { p.sp_slots = 0; p.sp_tick = nullptr; }
    }
    return p;
}
// blocks of a balanced == 2 launch: an upper bound of the units (the host does not know the elements' lengths; surplus blocks exit)
static unsigned split_grid(const AttnP& p, int owned_max, int streamed_max) {
    const int ks = (streamed_max + BSTR - 1) / BSTR;
    const int np = ks > p.sp_kmax ? std::min(p.sp_pmax, (ks + p.sp_kmax - 1) / p.sp_kmax) : 1;
    return (unsigned)(8 * p.B * ((owned_max + BOWN - 1) / BOWN) * np);
}

template <typename K> int set_lds(K kern, int bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

// 16-row waves (v4, dh = 64) for launches that cannot fill the SIMDs with 32-row waves: variable-length (packed) batches and
// launches with fewer than four 32-row waves per SIMD.  Measured (tools/attn_bench.py): packed self-attention forward 35.9 ->
// 32.9 us, cross-attention forward / backward 28.9 -> 23.5 / 57.1 -> 52.9 us, but the dense padded S = 1024 launch (4 096
// 32-row waves, VALU-bound) 67.9 -> 75.0 us - twice the LDS operand reads and barriers for no extra occupancy.
// PA_ATTN_V4=0 never, 2 always.
static bool use_v4(const AttnP& p, int rows_owned) {
    static const int mode = getenv("PA_ATTN_V4") ? atoi(getenv("PA_ATTN_V4")) : 1;
    if (mode == 0) return false;
    if (mode >= 2) return true;
    const long long waves32 = (long long)p.B * p.H * ((rows_owned + 31) / 32);
    return p.cu_q != nullptr || p.cu_k != nullptr || waves32 < 4 * 1024;
}
// In-block key split of the 16-row-wave forward / dQ kernels (attn4_fwd_kernel): for launches whose duration is one block's
// serial chain of key steps - few blocks, many keys.  PA_ATTN_KSPLIT: 0 never, 1 (default) cross-attention-like launches (at
// most one block per CU and at least 4 key tiles), 2 every eligible launch (no key-padding mask, not causal).
static bool use_ksplit(AttnP& p, unsigned blocks) {
    static const int mode = getenv("PA_ATTN_KSPLIT") ? atoi(getenv("PA_ATTN_KSPLIT")) : 1;
    static const int min_few = getenv("PA_ATTN_KSPLIT_MIN") ? atoi(getenv("PA_ATTN_KSPLIT_MIN")) : 4;
    static const int min_many = getenv("PA_ATTN_KSPLIT_MIN2") ? atoi(getenv("PA_ATTN_KSPLIT_MIN2")) : 10;
    if (mode == 0 || p.kpm || p.causal || p.Lk < 4 * BSTR) return false;
    // at most one block per CU (cross-attention): every element with >= 4 key tiles; more blocks than CUs (packed
    // self-attention, mode 2): only the long elements, whose chain of key steps sets the duration of the launch
    p.ks_min = blocks <= 256 ? min_few : min_many;
    return mode >= 2 || blocks <= 256;
}
// v5 forward (attention5.h): PA_ATTN_V5 = 0 never, 1 (default) self-attention-like launches (not causal, at least two 128-row
// query tiles per element), 2 every non-causal dh = 64 launch.
static bool use_v5(const AttnP& p) {
    static const int mode = getenv("PA_ATTN_V5") ? atoi(getenv("PA_ATTN_V5")) : 1;
    if (mode == 0 || p.causal) return false;
    return mode >= 2 || p.Lq > BOWN;
}
template <int DH> int run_fwd_bf16(AttnP p, hipStream_t st) {
    const int shm = BL<DH>::SHM;
    dim3 grid(((p.Lq + BOWN - 1) / BOWN) * p.H * p.B);
    if constexpr (DH == 64) {
        if (use_v5(p)) {
            // PA_ATTN_V5_OCC: 3 (default) three blocks per CU with a 3-stage K/V ring, 2: two blocks per CU, 4 stages
            static const int occ = getenv("PA_ATTN_V5_OCC") ? atoi(getenv("PA_ATTN_V5_OCC")) : 3;
            static const int rc4 = set_lds(attn5_fwd_kernel<true, 4, 2>, L5<4>::SHM) | set_lds(attn5_fwd_kernel<false, 4, 2>, L5<4>::SHM);
            if (rc4) return rc4;
            if (p.balanced && p.sp_tick) { p.balanced = 2; grid = dim3(split_grid(p, p.Lq, p.Lk)); }
            if (occ == 2) {
                if (p.drop_thr) PA_LAUNCH((attn5_fwd_kernel<true, 4, 2>), grid, dim3(NTH), L5<4>::SHM, st, p);
                else PA_LAUNCH((attn5_fwd_kernel<false, 4, 2>), grid, dim3(NTH), L5<4>::SHM, st, p);
            } else {
                if (p.drop_thr) PA_LAUNCH((attn5_fwd_kernel<true, 3, 3>), grid, dim3(NTH), L5<3>::SHM, st, p);
                else PA_LAUNCH((attn5_fwd_kernel<false, 3, 3>), grid, dim3(NTH), L5<3>::SHM, st, p);
            }
            return 0;
        }
        if (use_v4(p, p.Lq)) {
            if (use_ksplit(p, grid.x)) {
                constexpr int shm2 = 4 * BL<DH>::BUF + 64;                 // two pairs of stages: > 64 KiB, opt in once
                static const int rc_d = set_lds(attn4_fwd_kernel<true, 2>, shm2), rc_n = set_lds(attn4_fwd_kernel<false, 2>, shm2);
                if (rc_d || rc_n) return rc_d ? rc_d : rc_n;
                if (p.drop_thr) PA_LAUNCH((attn4_fwd_kernel<true, 2>), grid, dim3(2 * NT4), shm2, st, p);
                else PA_LAUNCH((attn4_fwd_kernel<false, 2>), grid, dim3(2 * NT4), shm2, st, p);
                return 0;
            }
            if (p.drop_thr) PA_LAUNCH((attn4_fwd_kernel<true, 1>), grid, dim3(NT4), shm, st, p);
            else PA_LAUNCH((attn4_fwd_kernel<false, 1>), grid, dim3(NT4), shm, st, p);
            return 0;
        }
    }
    if (p.drop_thr) PA_LAUNCH((attn_fwd_bf16_kernel<DH, true>), grid, dim3(NTH), shm, st, p);
    else PA_LAUNCH((attn_fwd_bf16_kernel<DH, false>), grid, dim3(NTH), shm, st, p);
    return 0;
}
template <int DH> int run_bwd_bf16(AttnP p, hipStream_t st) {
    const int shm = BL<DH>::SHM;
    const dim3 gq0(((p.Lq + BOWN - 1) / BOWN) * p.H * p.B), gk0(((p.Lk + BOWN - 1) / BOWN) * p.H * p.B);
    if constexpr (DH == 64) {
        if (use_v4(p, p.Lq > p.Lk ? p.Lq : p.Lk)) {
            dim3 gq = gq0, gk = gk0;
            // single query tile per (sample, head): dQ and dK/dV blocks in one launch (PA_ATTN_BWD_MERGE=0: two launches)
            static const bool merge_env = !(getenv("PA_ATTN_BWD_MERGE") && atoi(getenv("PA_ATTN_BWD_MERGE")) == 0);
            static const int merge_max = getenv("PA_ATTN_BWD_MERGE_MAX") ? atoi(getenv("PA_ATTN_BWD_MERGE_MAX")) : 128;
            if (merge_env && p.Lq <= merge_max && p.Lq <= 2048 && (gq.x & 7) == 0) {
                const int shm3 = shm + (p.Lq + 15) / 16 * 64 + 64;
                const dim3 gm(gq.x + gk.x);
                if (p.drop_thr) {
                    if (p.causal) PA_LAUNCH((attn4_bwd_merged_kernel<true, true>), gm, dim3(NT4), shm3, st, p, (int)gq.x);
                    else PA_LAUNCH((attn4_bwd_merged_kernel<true, false>), gm, dim3(NT4), shm3, st, p, (int)gq.x);
                } else {
                    if (p.causal) PA_LAUNCH((attn4_bwd_merged_kernel<false, true>), gm, dim3(NT4), shm3, st, p, (int)gq.x);
                    else PA_LAUNCH((attn4_bwd_merged_kernel<false, false>), gm, dim3(NT4), shm3, st, p, (int)gq.x);
                }
                return 0;
            }
            bool ks = use_ksplit(p, gq.x);
            constexpr int shm2 = 4 * BL<DH>::BUF + 64;
            if (p.balanced && p.sp_tick && !ks) { p.balanced = 2; gq = dim3(split_grid(p, p.Lq, p.Lk)); gk = dim3(split_grid(p, p.Lk, p.Lq)); }
            if (ks) {
                static const int rc_d = set_lds(attn4_bwd_dq_kernel<true, 2>, shm2), rc_n = set_lds(attn4_bwd_dq_kernel<false, 2>, shm2);
                if (rc_d || rc_n) return rc_d ? rc_d : rc_n;
            }
            if (p.drop_thr) {
                if (ks) PA_LAUNCH((attn4_bwd_dq_kernel<true, 2>), gq, dim3(2 * NT4), shm2, st, p);
                else PA_LAUNCH((attn4_bwd_dq_kernel<true, 1>), gq, dim3(NT4), shm, st, p);
                if (p.causal) PA_LAUNCH((attn4_bwd_dkv_kernel<true, true>), gk, dim3(NT4), shm, st, p);
                else PA_LAUNCH((attn4_bwd_dkv_kernel<true, false>), gk, dim3(NT4), shm, st, p);
            } else {
                if (ks) PA_LAUNCH((attn4_bwd_dq_kernel<false, 2>), gq, dim3(2 * NT4), shm2, st, p);
                else PA_LAUNCH((attn4_bwd_dq_kernel<false, 1>), gq, dim3(NT4), shm, st, p);
                if (p.causal) PA_LAUNCH((attn4_bwd_dkv_kernel<false, true>), gk, dim3(NT4), shm, st, p);
                else PA_LAUNCH((attn4_bwd_dkv_kernel<false, false>), gk, dim3(NT4), shm, st, p);
            }
            return 0;
        }
    }
    const dim3 gq = gq0, gk = gk0;
    // blocks per CU the register allocation is made for (experiment knob PA_ATTN_OCC="<dq><dkv>", e.g. "43")
    static const int occ_env = getenv("PA_ATTN_OCC") ? atoi(getenv("PA_ATTN_OCC")) : 0;
    const int oq = occ_env ? occ_env / 10 : 3, ok = occ_env ? occ_env % 10 : 2;
    if (p.drop_thr) {
        if (oq >= 4) PA_LAUNCH((attn_bwd_dq_bf16_kernel<DH, true, 4>), gq, dim3(NTH), shm, st, p);      // also writes delta
        else PA_LAUNCH((attn_bwd_dq_bf16_kernel<DH, true, 3>), gq, dim3(NTH), shm, st, p);
        if (ok >= 3) { if (p.causal) PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, true, 3, true>), gk, dim3(NTH), shm, st, p); else PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, true, 3, false>), gk, dim3(NTH), shm, st, p); }
        else { if (p.causal) PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, true, 2, true>), gk, dim3(NTH), shm, st, p); else PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, true, 2, false>), gk, dim3(NTH), shm, st, p); }
    } else {
        if (oq >= 4) PA_LAUNCH((attn_bwd_dq_bf16_kernel<DH, false, 4>), gq, dim3(NTH), shm, st, p);
        else PA_LAUNCH((attn_bwd_dq_bf16_kernel<DH, false, 3>), gq, dim3(NTH), shm, st, p);
        if (ok >= 3) { if (p.causal) PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, false, 3, true>), gk, dim3(NTH), shm, st, p); else PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, false, 3, false>), gk, dim3(NTH), shm, st, p); }
        else { if (p.causal) PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, false, 2, true>), gk, dim3(NTH), shm, st, p); else PA_LAUNCH((attn_bwd_dkv_bf16_kernel<DH, false, 2, false>), gk, dim3(NTH), shm, st, p); }
    }
    return 0;
}

// bf16x3 attention (attention_x3.h) for f32 launches with dh = 64 while pa_attn_split_config(1) is in force
extern "C" int pa_split_attn_set(int32_t on);       // gemm.hip: the flag lives in the current bf16x3 context (pa_split_ctx_*)
extern "C" int pa_split_attn_active(void);
std::atomic<long long> g_attn_x3_taken{0};
template <typename T, int DH> int run_fwd(const AttnP& p, hipStream_t st) {
    if constexpr (sizeof(T) == 2) return run_fwd_bf16<DH>(p, st);
    if constexpr (sizeof(T) == 4 && DH == 64) {
        if (pa_split_attn_active()) {
            const int shm = X3L<DH>::SHM;
            static const int rc_ = set_lds(attnx_fwd_kernel<DH, true>, shm) | set_lds(attnx_fwd_kernel<DH, false>, shm);
            if (rc_) return rc_;
            const dim3 gq((p.Lq + BOWN - 1) / BOWN, p.H, p.B);
            if (p.drop_thr) PA_LAUNCH((attnx_fwd_kernel<DH, true>), gq, dim3(NTH), shm, st, p);
            else PA_LAUNCH((attnx_fwd_kernel<DH, false>), gq, dim3(NTH), shm, st, p);
            g_attn_x3_taken.fetch_add(1);
            return 0;
        }
    }
    const int shm = 2 * Smem<T, DH>::BUF_FWD;
    int rc = set_lds(attn_fwd_kernel<T, DH>, shm);
    if (rc) return rc;
    dim3 grid((p.Lq + BOWN - 1) / BOWN, p.H, p.B);
    PA_LAUNCH((attn_fwd_kernel<T, DH>), grid, dim3(NTH), shm, st, p);
    return 0;
}
template <typename T, int DH> int run_bwd(const AttnP& p_in, hipStream_t st) {
    if constexpr (sizeof(T) == 2) return run_bwd_bf16<DH>(p_in, st);
    AttnP p = p_in;
    const int64_t total = (int64_t)p.B * p.H * p.Lq;
    bool x3 = false;
    if constexpr (sizeof(T) == 4 && DH == 64) {
        x3 = pa_split_attn_active() != 0;
        if (x3) {
            // PA_X3_PARTS=n (n >= 2): range-split backward, n blocks per owned tile adding their partial dQ / dK / dV with f32
            // atomics (attention_x3.h x3_add_rows).  OFF by default - MEASURED on MI355X, round 5, x3 train step: 13.0 ms unsplit,
            // 16.4 ms with 2 parts, 20.2 ms with 4 (packed encoder backward 360 -> 843 us): the atomic adds of 128 x 64 f32 tiles
            // at a 6 KB row stride cost several times the chain they shorten.  Results are equal to rounding either way
            // (tests/test_kernels_gpu.py::test_attention_x3_* pass with PA_X3_PARTS=4).
            static const int parts_env = getenv("PA_X3_PARTS") ? atoi(getenv("PA_X3_PARTS")) : 0;
            static const int parts_min = getenv("PA_X3_PARTS_MIN") ? atoi(getenv("PA_X3_PARTS_MIN")) : 512;
            const int n = parts_env;
            if (n >= 2) {
                if (p.Lk >= parts_min) p.parts_q = n;
                if (p.Lq >= parts_min && p.Lk <= p.Lq && (!p.cu_q || p.cu_q == p.cu_k)) p.parts_kv = n;     // self-attention shapes only
            }
        }
    }
    PA_LAUNCH((attn_delta_kernel<T, DH>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    if constexpr (sizeof(T) == 4 && DH == 64) {
        if (x3) {
            const int shk = X3L<DH>::SHM, shq = X3L<DH>::SHM;
            static const int rc_ = set_lds(attnx_bwd_dkv_kernel<DH, true, 2>, shk) | set_lds(attnx_bwd_dq_kernel<DH, true>, shq) |
                                   set_lds(attnx_bwd_dkv_kernel<DH, false, 2>, shk) | set_lds(attnx_bwd_dq_kernel<DH, false>, shq) |
                                   set_lds(attnx_bwd_dkv_kernel<DH, true, 1>, shk) | set_lds(attnx_bwd_dkv_kernel<DH, false, 1>, shk);
            if (rc_) return rc_;
            // dK / dV at two blocks per CU sits at the 256-register limit and spills 160 bytes per lane; allocated for ONE block per CU
            // (no spills, one wave per SIMD) the x3 step measures 12.72 against 12.82 ms (A/B twice in one session): default 1
            static const int dkv_occ = getenv("PA_X3_DKV_OCC") ? atoi(getenv("PA_X3_DKV_OCC")) : 1;
            const dim3 gk(((p.Lk + BOWN - 1) / BOWN) * p.parts_kv, p.H, p.B), gq(((p.Lq + BOWN - 1) / BOWN) * p.parts_q, p.H, p.B);
            if (p.drop_thr) {
                if (dkv_occ == 1) PA_LAUNCH((attnx_bwd_dkv_kernel<DH, true, 1>), gk, dim3(NTH), shk, st, p);
                else PA_LAUNCH((attnx_bwd_dkv_kernel<DH, true, 2>), gk, dim3(NTH), shk, st, p);
                PA_LAUNCH((attnx_bwd_dq_kernel<DH, true>), gq, dim3(NTH), shq, st, p);
            } else {
                if (dkv_occ == 1) PA_LAUNCH((attnx_bwd_dkv_kernel<DH, false, 1>), gk, dim3(NTH), shk, st, p);
                else PA_LAUNCH((attnx_bwd_dkv_kernel<DH, false, 2>), gk, dim3(NTH), shk, st, p);
                PA_LAUNCH((attnx_bwd_dq_kernel<DH, false>), gq, dim3(NTH), shq, st, p);
            }
            g_attn_x3_taken.fetch_add(1);
            return 0;
        }
    }
    int shm = 2 * Smem<T, DH>::BUF_DKV;
    int rc = set_lds(attn_bwd_dkv_kernel<T, DH>, shm);
    if (rc) return rc;
    PA_LAUNCH((attn_bwd_dkv_kernel<T, DH>), dim3((p.Lk + BOWN - 1) / BOWN, p.H, p.B), dim3(NTH), shm, st, p);
    shm = 2 * Smem<T, DH>::BUF_DQ;
    rc = set_lds(attn_bwd_dq_kernel<T, DH>, shm);
    if (rc) return rc;
    PA_LAUNCH((attn_bwd_dq_kernel<T, DH>), dim3((p.Lq + BOWN - 1) / BOWN, p.H, p.B), dim3(NTH), shm, st, p);
    return 0;
}

template <typename T> int dispatch(const pa_attn_args* a, bool bwd, hipStream_t st) {
    const AttnP p = make_params(a);
    switch (a->dh) {
        case 16: return bwd ? run_bwd<T, 16>(p, st) : run_fwd<T, 16>(p, st);
        case 32: return bwd ? run_bwd<T, 32>(p, st) : run_fwd<T, 32>(p, st);
        case 64: return bwd ? run_bwd<T, 64>(p, st) : run_fwd<T, 64>(p, st);
        default: return PA_ESHAPE;
    }
}

int check_args(const pa_attn_args* a, bool bwd) {
    if (!a || !a->q || !a->k || !a->v || !a->o || !a->lse) return PA_EINVAL;
    if (a->B <= 0 || a->H <= 0 || a->Lq <= 0 || a->Lk <= 0) return PA_EINVAL;
    if (a->dtype != PA_F32 && a->dtype != PA_BF16) return PA_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return PA_EINVAL;
    const int EB = a->dtype == PA_BF16 ? 8 : 4;
    auto al = [&](const void* ptr, int ld) { return (reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && ld % EB == 0; };
    if (!al(a->q, a->ldq) || !al(a->k, a->ldk) || !al(a->v, a->ldv) || !al(a->o, a->ldo)) return PA_EALIGN;
    if (bwd) {
        if (!a->dout || !a->dq || !a->dk || !a->dv || !a->delta) return PA_EINVAL;
        if (!al(a->dout, a->lddo) || !al(a->dq, a->lddq) || !al(a->dk, a->lddk) || !al(a->dv, a->lddv)) return PA_EALIGN;
    }
    return 0;
}

}  // namespace

extern "C" int pa_attn_split_config(int32_t on) { return pa_split_attn_set(on); }
extern "C" int64_t pa_attn_split_taken(int32_t reset) {
    const long long v = g_attn_x3_taken.load();
    if (reset) g_attn_x3_taken.store(0);
    return v;
}

extern "C" int64_t pa_attn_ws_bytes(int32_t rows_total, int32_t B, int32_t H, int32_t L_max) {
    static const bool sp_env = getenv("PA_ATTN_SPLIT") && atoi(getenv("PA_ATTN_SPLIT")) != 0;     // opt-in: measured slower (profiles/r06_attention_launch_shape.txt)
    if (!sp_env || split_pmax() < 2 || H != 8 || rows_total <= 0 || B <= 0 || B > 64) return 0;
    if ((L_max + BSTR - 1) / BSTR <= split_kmax()) return 0;
    const int64_t slots = (int64_t)(rows_total / BOWN + B);             // owned tiles of all elements, an upper bound
    return slots * 8 * ((int64_t)split_pmax() * SP_BYTES + 64) + 256;
}
extern "C" int64_t pa_attn_ws_ticket_bytes(int64_t ws_bytes) {
    const int64_t per = (int64_t)split_pmax() * SP_BYTES + 64;
    return (ws_bytes / per / 8 * 8 * 4 + 255) / 256 * 256;
}

extern "C" int pa_attn_fwd(const pa_attn_args* a, void* stream) {
    int rc = check_args(a, false);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return a->dtype == PA_BF16 ? dispatch<bf16>(a, false, st) : dispatch<float>(a, false, st);
}

extern "C" int pa_attn_bwd(const pa_attn_args* a, void* stream) {
    int rc = check_args(a, true);
    if (rc) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    return a->dtype == PA_BF16 ? dispatch<bf16>(a, true, st) : dispatch<float>(a, true, st);
}

#ifdef PA_ATTN_TRACE
extern "C" int pa_attn_trace_read(unsigned long long* out, int32_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pa_attn_trace), sizeof(unsigned long long) * (size_t)n);
}
#endif
