// MFMA GEMM with fused epilogue for gfx950.  See include/plank_hip.h (pa_gemm).
//
// Block = 256 threads = 4 waves (2 x 2), block tile 128 x 128, each wave 64 x 64 = 2 x 2 MFMA
// 32x32 tiles (64 accumulator registers).  K tile: 128 bytes of contraction per row for bf16
// (BK = 64), 64 bytes for f32 (BK = 16).  Operand tiles are staged global -> registers -> LDS
// as [row][k] images (k contiguous, 16-byte chunks XOR-swizzled by row so that the 32 rows a
// half-wave reads with ds_read_b128 spread over all banks).  Operands whose contraction index
// is NOT contiguous in memory (dX = dY W, dW = dY^T X) are transposed in registers
// (EB x EB element blocks) on their way into LDS, so the MFMA side is identical for all layouts.
// Double-buffered LDS, one barrier per K tile; the global loads of tile t+1 are in flight
// while tile t is multiplied.
#include "common.cuh"
#include "../../include/plank_hip.h"

namespace {

struct GemmP {
    const void* A; const void* B; void* C;
    const float* bias; const void* R; const void* aux;
    int M, N, K;
    int lda, ldb, ldc, ldr, ldaux;
    long long sA, sB, sC, sR, sAux;
    int batch;
    float alpha; int relu; float aux_scale;
    uint32_t drop_thr; float drop_scale; uint32_t drop_seed;
    int out_dtype;
    int splitk, tiles_per_slice;   // split-K: C is the f32 slab workspace, plain store
    int tiles_n;
};

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T> struct Tile {
    static constexpr int EB = ET<T>::EB;
    static constexpr int BK = (sizeof(T) == 2) ? 64 : 16;
    static constexpr int RB = BK * sizeof(T);        // bytes per LDS row (128 / 64)
    static constexpr int NCH = RB / 16;              // 16-byte chunks per row (8 / 4)
    static constexpr int RPB = 256 / RB;             // rows per 256-byte bank row (2 / 4)
    static constexpr int STEPS = BK / ET<T>::KC;     // mma16B steps per K tile (4 / 2)
    static constexpr int NLD = BM * NCH / NT;        // 16-byte loads per thread, k-contiguous (4 / 2)
    static constexpr int TILE_BYTES = BM * RB;       // 16 KiB / 8 KiB per operand
};

template <typename T> __device__ __forceinline__ int lds_off(int row, int chunk) {
    using TL = Tile<T>;
    return row * TL::RB + (((chunk ^ (row / TL::RPB)) & (TL::NCH - 1)) << 4);
}

// ---- global -> register staging ---------------------------------------------------------------
// k-contiguous operand: element (r, k) at base[r * ld + k]; rows r0.., contraction k0..
template <typename T, bool ALIGNED>
__device__ __forceinline__ void load_kc(u32x4* regs, const T* base, int ld, int r0, int nrows, int k0, int K, int tid) {
    using TL = Tile<T>;
#pragma unroll
    for (int i = 0; i < TL::NLD; ++i) {
        int c = tid + i * NT;
        int row = c / TL::NCH, ch = c % TL::NCH;
        int r = r0 + row, k = k0 + ch * TL::EB;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < nrows) {
            const T* p = base + (size_t)r * ld + k;
            if (ALIGNED) {
                if (k < K) v = *reinterpret_cast<const u32x4*>(p);
            } else {
                T tmp[TL::EB];
#pragma unroll
                for (int e = 0; e < TL::EB; ++e) tmp[e] = (k + e < K) ? p[e] : (T)0.0f;
                v = *reinterpret_cast<u32x4*>(tmp);
            }
        }
        regs[i] = v;
    }
}
template <typename T>
__device__ __forceinline__ void store_kc(const u32x4* regs, char* lds, int tid) {
    using TL = Tile<T>;
#pragma unroll
    for (int i = 0; i < TL::NLD; ++i) {
        int c = tid + i * NT;
        int row = c / TL::NCH, ch = c % TL::NCH;
        *reinterpret_cast<u32x4*>(lds + lds_off<T>(row, ch)) = regs[i];
    }
}
// transposed operand: element (r, k) at base[k * ld + r]; one EB x EB block per thread (128 threads)
template <typename T, bool ALIGNED>
__device__ __forceinline__ void load_tr(u32x4* regs, const T* base, int ld, int r0, int nrows, int k0, int K, int t128) {
    using TL = Tile<T>;
    constexpr int RBLK = BM / TL::EB;                // row blocks per tile (16 / 32)
    int rb = t128 % RBLK, kb = t128 / RBLK;
    int r = r0 + rb * TL::EB;
#pragma unroll
    for (int i = 0; i < TL::EB; ++i) {
        int k = k0 + kb * TL::EB + i;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (k < K) {
            const T* p = base + (size_t)k * ld + r;
            if (ALIGNED) {
                if (r < nrows) v = *reinterpret_cast<const u32x4*>(p);   // nrows % EB == 0 when ALIGNED
            } else {
                T tmp[TL::EB];
#pragma unroll
                for (int e = 0; e < TL::EB; ++e) tmp[e] = (r + e < nrows) ? p[e] : (T)0.0f;
                v = *reinterpret_cast<u32x4*>(tmp);
            }
        }
        regs[i] = v;
    }
}
template <typename T>
__device__ __forceinline__ void store_tr(const u32x4* regs, char* lds, int t128) {
    using TL = Tile<T>;
    constexpr int RBLK = BM / TL::EB;
    int rb = t128 % RBLK, kb = t128 / RBLK;
    u32x4 tr[TL::EB];
    transpose_block<T>(regs, tr);
#pragma unroll
    for (int e = 0; e < TL::EB; ++e)
        *reinterpret_cast<u32x4*>(lds + lds_off<T>(rb * TL::EB + e, kb)) = tr[e];
}

template <typename T, bool A_KC, bool B_KC, bool ALIGNED>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
    using TL = Tile<T>;
    __shared__ __attribute__((aligned(16))) char smem[4 * TL::TILE_BYTES];   // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int z = blockIdx.y;
    const int b = z / p.splitk, slice = z % p.splitk;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const T* A = reinterpret_cast<const T*>(p.A) + (size_t)b * p.sA;
    const T* B = reinterpret_cast<const T*>(p.B) + (size_t)b * p.sB;

    const int nt_total = (p.K + TL::BK - 1) / TL::BK;
    const int t_begin = slice * p.tiles_per_slice;
    const int t_end = min(nt_total, t_begin + p.tiles_per_slice);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NRA = A_KC ? TL::NLD : TL::EB;
    constexpr int NRB = B_KC ? TL::NLD : TL::EB;
    u32x4 ra[NRA], rb[NRB];
    // thread roles for transposed operands: A blocks on threads 0..127, B blocks on 128..255
    const bool a_role = A_KC || (tid < 128);
    const bool b_role = B_KC || (tid >= 128);
    const int t128 = tid & 127;

    auto gload = [&](int t) {
        int k0 = t * TL::BK;
        if constexpr (A_KC) load_kc<T, ALIGNED>(ra, A, p.lda, m0, p.M, k0, p.K, tid);
        else { if (a_role) load_tr<T, ALIGNED>(ra, A, p.lda, m0, p.M, k0, p.K, t128); }
        if constexpr (B_KC) load_kc<T, ALIGNED>(rb, B, p.ldb, n0, p.N, k0, p.K, tid);
        else { if (b_role) load_tr<T, ALIGNED>(rb, B, p.ldb, n0, p.N, k0, p.K, t128); }
    };
    auto lstore = [&](int buf) {
        char* la = smem + buf * 2 * TL::TILE_BYTES;
        char* lb = la + TL::TILE_BYTES;
        if constexpr (A_KC) store_kc<T>(ra, la, tid);
        else { if (a_role) store_tr<T>(ra, la, t128); }
        if constexpr (B_KC) store_kc<T>(rb, lb, tid);
        else { if (b_role) store_tr<T>(rb, lb, t128); }
    };

    if (t_begin < t_end) {
        gload(t_begin);
        lstore(0);
    }
    __syncthreads();

    const int arow = wm * 64 + (lane & 31), brow = wn * 64 + (lane & 31), half = lane >> 5;
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) gload(t + 1);
        const char* la = smem + buf * 2 * TL::TILE_BYTES;
        const char* lb = la + TL::TILE_BYTES;
#pragma unroll
        for (int s = 0; s < TL::STEPS; ++s) {
            u32x4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const u32x4*>(la + lds_off<T>(arow + i * 32, 2 * s + half));
                fb[i] = *reinterpret_cast<const u32x4*>(lb + lds_off<T>(brow + i * 32, 2 * s + half));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma16B<T>(acc[i][j], fa[i], fb[j]);
        }
        if (t + 1 < t_end) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------
    const bool slab = p.splitk > 1;
    const size_t cbase = slab ? (size_t)z * p.M * p.ldc : (size_t)b * p.sC;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (n >= p.N) continue;
        const float bias = (!slab && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (m >= p.M) continue;
                float v = acc[i][j][r];
                if (slab) {
                    reinterpret_cast<float*>(p.C)[cbase + (size_t)m * p.ldc + n] = v;
                    continue;
                }
                v = v * p.alpha + bias;
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.aux) {
                    float g = ld1(reinterpret_cast<const T*>(p.aux) + (size_t)b * p.sAux + (size_t)m * p.ldaux + n);
                    v = g > 0.f ? v * p.aux_scale : 0.f;
                }
                if (p.drop_thr) {
                    uint32_t idx = (uint32_t)(((size_t)b * p.M + m) * p.N + n);
                    v = drop_keep(p.drop_seed, idx, p.drop_thr) ? v * p.drop_scale : 0.f;
                }
                const size_t co = cbase + (size_t)m * p.ldc + n;
                if (p.out_dtype == PA_F32) {
                    if (p.R) v += reinterpret_cast<const float*>(p.R)[(size_t)b * p.sR + (size_t)m * p.ldr + n];
                    reinterpret_cast<float*>(p.C)[co] = v;
                } else {
                    if (p.R) v += (float)reinterpret_cast<const bf16*>(p.R)[(size_t)b * p.sR + (size_t)m * p.ldr + n];
                    reinterpret_cast<bf16*>(p.C)[co] = (bf16)v;
                }
            }
        }
    }
}

// split-K second pass: sum the slabs and apply the epilogue
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, const float* ws) {
    const size_t total = (size_t)p.batch * p.M * p.N;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(e % p.N);
        const size_t bm = e / p.N;
        const int m = (int)(bm % p.M), b = (int)(bm / p.M);
        float v = 0.f;
        for (int s = 0; s < p.splitk; ++s) v += ws[(((size_t)b * p.splitk + s) * p.M + m) * p.N + n];
        v = v * p.alpha + (p.bias ? p.bias[n] : 0.f);
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.aux) {
            float g = ld1(reinterpret_cast<const T*>(p.aux) + (size_t)b * p.sAux + (size_t)m * p.ldaux + n);
            v = g > 0.f ? v * p.aux_scale : 0.f;
        }
        if (p.drop_thr) v = drop_keep(p.drop_seed, (uint32_t)e, p.drop_thr) ? v * p.drop_scale : 0.f;
        const size_t co = (size_t)b * p.sC + (size_t)m * p.ldc + n;
        if (p.out_dtype == PA_F32) {
            if (p.R) v += reinterpret_cast<const float*>(p.R)[(size_t)b * p.sR + (size_t)m * p.ldr + n];
            reinterpret_cast<float*>(p.C)[co] = v;
        } else {
            if (p.R) v += (float)reinterpret_cast<const bf16*>(p.R)[(size_t)b * p.sR + (size_t)m * p.ldr + n];
            reinterpret_cast<bf16*>(p.C)[co] = (bf16)v;
        }
    }
}

template <typename T, bool A_KC, bool B_KC>
int launch_t(const GemmP& p, bool aligned, dim3 grid, hipStream_t st) {
    if (aligned) PA_LAUNCH((gemm_kernel<T, A_KC, B_KC, true>), grid, dim3(NT), 0, st, p);
    else PA_LAUNCH((gemm_kernel<T, A_KC, B_KC, false>), grid, dim3(NT), 0, st, p);
    return 0;
}
template <typename T>
int launch_layout(const GemmP& p, bool akc, bool bkc, bool aligned, dim3 grid, hipStream_t st) {
    if (akc && bkc) return launch_t<T, true, true>(p, aligned, grid, st);
    if (akc && !bkc) return launch_t<T, true, false>(p, aligned, grid, st);
    if (!akc && bkc) return launch_t<T, false, true>(p, aligned, grid, st);
    return launch_t<T, false, false>(p, aligned, grid, st);
}

template <typename T> bool is_aligned(const pa_gemm_args* a) {
    constexpr int EB = ET<T>::EB;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool ok = al16(a->A) && al16(a->B) && (a->lda % EB == 0) && (a->ldb % EB == 0) &&
              (a->sA % EB == 0) && (a->sB % EB == 0);
    // vectors run along K for k-contiguous operands, along the row index for transposed ones
    ok = ok && (a->a_kcontig ? (a->K % EB == 0) : (a->M % EB == 0));
    ok = ok && (a->b_kcontig ? (a->K % EB == 0) : (a->N % EB == 0));
    return ok;
}

}  // namespace

extern "C" int pa_gemm(const pa_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->C) return PA_EINVAL;
    if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return PA_EINVAL;
    if (a->in_dtype != PA_F32 && a->in_dtype != PA_BF16) return PA_EINVAL;
    if (a->out_dtype != PA_F32 && a->out_dtype != PA_BF16) return PA_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return PA_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    GemmP p;
    p.A = a->A; p.B = a->B; p.C = a->C; p.bias = a->bias; p.R = a->R; p.aux = a->aux;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.lda = a->lda; p.ldb = a->ldb; p.ldc = a->ldc; p.ldr = a->ldr; p.ldaux = a->ldaux;
    p.sA = a->sA; p.sB = a->sB; p.sC = a->sC; p.sR = a->sR; p.sAux = a->sAux;
    p.batch = a->batch;
    p.alpha = a->alpha; p.relu = a->relu; p.aux_scale = a->aux_scale;
    p.drop_thr = (uint32_t)(a->drop_p * 65536.0f + 0.5f);
    p.drop_scale = 1.0f / (1.0f - a->drop_p);
    p.drop_seed = a->drop_seed;
    p.out_dtype = a->out_dtype;
    const int BK = a->in_dtype == PA_BF16 ? Tile<bf16>::BK : Tile<float>::BK;
    const int nt = (a->K + BK - 1) / BK;
    int splitk = a->splitk > 1 ? a->splitk : 1;
    if (splitk > nt) splitk = nt;
    if (splitk > 1 && !a->ws) return PA_EINVAL;
    p.splitk = splitk;
    p.tiles_per_slice = (nt + splitk - 1) / splitk;
    const int tiles_m = (a->M + BM - 1) / BM;
    p.tiles_n = (a->N + BN - 1) / BN;
    dim3 grid(tiles_m * p.tiles_n, a->batch * splitk);
    GemmP pk = p;
    if (splitk > 1) { pk.C = a->ws; pk.ldc = a->N; }
    int rc;
    if (a->in_dtype == PA_BF16)
        rc = launch_layout<bf16>(pk, a->a_kcontig, a->b_kcontig, is_aligned<bf16>(a), grid, st);
    else
        rc = launch_layout<float>(pk, a->a_kcontig, a->b_kcontig, is_aligned<float>(a), grid, st);
    if (rc) return rc;
    if (splitk > 1) {
        size_t total = (size_t)a->batch * a->M * a->N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (a->in_dtype == PA_BF16)
            PA_LAUNCH(splitk_reduce_kernel<bf16>, dim3(blocks), dim3(256), 0, st, p, (const float*)a->ws);
        else
            PA_LAUNCH(splitk_reduce_kernel<float>, dim3(blocks), dim3(256), 0, st, p, (const float*)a->ws);
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// column sums (bias gradients)
namespace {
constexpr int CS_ROWS = 256;   // rows per block
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* X, int M, int N, int ldx, float* partial) {
    // block (x = column group of 256, y = row group); thread = one column; coalesced along n
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * CS_ROWS;
    const int r1 = min(M, r0 + CS_ROWS);
    if (n >= N) return;
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += ld1(X + (size_t)r * ldx + n);
    partial[(size_t)blockIdx.y * N + n] = s;
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* partial, int nparts, int N, float* out, int accumulate) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int i = 0; i < nparts; ++i) s += partial[(size_t)i * N + n];
    out[n] = accumulate ? out[n] + s : s;
}
}  // namespace

extern "C" int64_t pa_colsum_ws_floats(int32_t M, int32_t N) {
    return (int64_t)((M + CS_ROWS - 1) / CS_ROWS) * N;
}
extern "C" int pa_colsum(const void* X, int32_t dtype, int32_t M, int32_t N, int32_t ldx, float* out,
                         int32_t accumulate, float* partial, void* stream) {
    if (!X || !out || !partial || M <= 0 || N <= 0) return PA_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nparts = (M + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((N + 255) / 256, nparts);
    if (dtype == PA_BF16)
        PA_LAUNCH(colsum_partial_kernel<bf16>, grid, dim3(256), 0, st, (const bf16*)X, M, N, ldx, partial);
    else
        PA_LAUNCH(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)X, M, N, ldx, partial);
    PA_LAUNCH(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, partial, nparts, N, out, accumulate);
    return 0;
}
