// HBM-bound row kernels of the PlankAssembly hot path for gfx950: embedding gathers (+ scatter-add
// backward), LayerNorm fwd/bwd, switch head, fused mixture-NLL fwd/bwd, Adam, casts.
// One 64-lane wave per row with 16-byte (4 x f32 / 4 x bf16 = 8-byte) vector accesses and
// shuffle reductions; parameter-gradient column sums go through per-block partials (deterministic).
#include <type_traits>
#include "pa_device.h"
#include "../../include/plank_hip.h"

namespace {

constexpr int MAXV = 8;   // vectors of 4 per lane per row -> d <= 2048

// ================================================================================ embeddings
struct EmbTabs { const float* t[5]; const int64_t* idx[5]; int n; const int32_t* rowmap; };
struct EmbGrads { float* t[5]; const int64_t* idx[5]; int rows[5]; int n; const int32_t* rowmap; };

template <typename T>
__global__ __launch_bounds__(256) void embed_input_fwd_kernel(T* out, EmbTabs tb, int64_t n_tok, int d) {
    const int vec_per_row = d >> 2;
    const int64_t total = n_tok * vec_per_row;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tok = e / vec_per_row;
        const int c = (int)(e % vec_per_row) << 2;
        const int64_t st = tb.rowmap ? (int64_t)tb.rowmap[tok] : tok;     // packed row -> position in the [B*S] id tensors
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k < tb.n && tb.idx[k]) {
                const int64_t r = tb.idx[k][st];
                acc += *reinterpret_cast<const f32x4*>(tb.t[k] + r * d + c);
            }
        }
        st4<T>(out + tok * d + c, acc);
    }
}

constexpr int EMB_TOK_PER_BLOCK = 64;
constexpr int EMB_SMALL_ROWS = 8;      // tables with <= 8 rows (coord, view, type) are reduced per block in LDS

// Scatter-add of d_out into the table gradients.  Tables with a handful of rows would serialise ~n_tok atomics
// per address in L2; they are accumulated per block in LDS (ds_add_f32) and flushed with one global atomic per
// (row, column) per block.  Large tables (value, pos) use global atomics directly; all-zero gradient vectors
// (padded positions: no gradient ever reaches them) are skipped.
template <typename T>
__global__ __launch_bounds__(256) void embed_input_bwd_kernel(const T* dout, EmbGrads tb, int64_t n_tok, int d) {
    extern __shared__ __attribute__((aligned(16))) float acc[];        // [n_small * EMB_SMALL_ROWS][d]
    const int nvec = d >> 2;
    const int lanes = 256 / nvec > 0 ? 256 / nvec : 1;                  // tokens processed concurrently
    const int vec = threadIdx.x % nvec, tl = threadIdx.x / nvec;
    int slot[5], nsmall = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) slot[k] = (k < tb.n && tb.idx[k] && tb.rows[k] <= EMB_SMALL_ROWS) ? nsmall++ : -1;
    for (int e = threadIdx.x; e < nsmall * EMB_SMALL_ROWS * d; e += 256) acc[e] = 0.f;
    __syncthreads();
    const int64_t t0 = (int64_t)blockIdx.x * EMB_TOK_PER_BLOCK;
    if (tl < lanes) {
        for (int v = vec; v < nvec; v += (nvec > 256 ? 256 : nvec)) {
            const int c = v << 2;
            for (int i = tl; i < EMB_TOK_PER_BLOCK; i += lanes) {
                const int64_t tok = t0 + i;
                if (tok >= n_tok) break;
                const f32x4 g = ld4<T>(dout + tok * d + c);
                if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f && g[3] == 0.f) continue;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    if (!(k < tb.n && tb.idx[k])) continue;
                    const int64_t r = tb.idx[k][tb.rowmap ? (int64_t)tb.rowmap[tok] : tok];
                    if (slot[k] >= 0) {
                        float* dst = acc + ((size_t)(slot[k] * EMB_SMALL_ROWS + r)) * d + c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) atomicAdd(dst + j, g[j]);      // LDS atomic
                    } else {
                        float* dst = tb.t[k] + r * d + c;
#pragma unroll
                        for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dst + j, g[j]);
                    }
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (slot[k] < 0) continue;
        for (int e = threadIdx.x; e < tb.rows[k] * d; e += 256) {
            const float v = acc[(size_t)slot[k] * EMB_SMALL_ROWS * d + e];
            if (v != 0.f) unsafeAtomicAdd(tb.t[k] + e, v);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_output_fwd_kernel(T* out, const float* value, const float* coord,
                                                               const float* pos, const int64_t* tok, int tok_ld,
                                                               int B, int Tn, int d, int dof) {
    const int vec_per_row = d >> 2;
    const int64_t total = (int64_t)B * Tn * vec_per_row;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / vec_per_row;
        const int c = (int)(e % vec_per_row) << 2;
        const int b = (int)(row / Tn), t = (int)(row % Tn);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            const int64_t v = tok[(int64_t)b * tok_ld + (t - 1)];
            acc = *reinterpret_cast<const f32x4*>(value + v * d + c);
            acc += *reinterpret_cast<const f32x4*>(coord + (int64_t)((t - 1) % dof) * d + c);
            acc += *reinterpret_cast<const f32x4*>(pos + (int64_t)((t - 1) / dof) * d + c);
        }
        st4<T>(out + row * d + c, acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_output_bwd_kernel(const T* dout, float* dvalue, float* dcoord, float* dpos,
                                                               const int64_t* tok, int tok_ld, int B, int Tn, int d, int dof) {
    const int vec_per_row = d >> 2;
    const int64_t total = (int64_t)B * Tn * vec_per_row;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / vec_per_row;
        const int c = (int)(e % vec_per_row) << 2;
        const int b = (int)(row / Tn), t = (int)(row % Tn);
        if (t == 0) continue;
        const f32x4 g = ld4<T>(dout + row * d + c);
        const int64_t v = tok[(int64_t)b * tok_ld + (t - 1)];
        float* p0 = dvalue + v * d + c;
        float* p1 = dcoord + (int64_t)((t - 1) % dof) * d + c;
        float* p2 = dpos + (int64_t)((t - 1) / dof) * d + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) { unsafeAtomicAdd(p0 + j, g[j]); unsafeAtomicAdd(p1 + j, g[j]); unsafeAtomicAdd(p2 + j, g[j]); }
    }
}

// ================================================================================ row packing
// "Unpadding": padded encoder positions never influence the loss, so the encoder runs on the valid rows only.
// cu[b] = number of valid positions before batch element b (cu[B] = total), rowmap[packed row] = b*S + s.
__global__ __launch_bounds__(256) void pack_count_kernel(const uint8_t* mask, int S, int32_t* cnt) {
    __shared__ int red[4];
    const int b = blockIdx.x;
    int c = 0;
    for (int s = threadIdx.x; s < S; s += 256) c += mask[(size_t)b * S + s] ? 0 : 1;
    c = (int)wave_sum((float)c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[b] = red[0] + red[1] + red[2] + red[3];
}
__global__ void pack_scan_kernel(const int32_t* cnt, int B, int32_t* cu) {
    if (threadIdx.x == 0) { int a = 0; for (int b = 0; b < B; ++b) { cu[b] = a; a += cnt[b]; } cu[B] = a; }
}
__global__ void pack_iota_kernel(int32_t* out, int B) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < B) out[i] = i; }
// cnt[B] (rows per batch element) -> in place: the batch elements by descending row count (ties: batch order)
__global__ __launch_bounds__(256) void pack_order_kernel(int32_t* cnt, int B) {
    extern __shared__ int len_s[];
    for (int i = threadIdx.x; i < B; i += 256) len_s[i] = cnt[i];
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += 256) {
        const int li = len_s[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (len_s[j] > li || (len_s[j] == li && j < i)) ? 1 : 0;
        cnt[rank] = i;
    }
}
__global__ __launch_bounds__(256) void pack_fill_kernel(const uint8_t* mask, int S, const int32_t* cu, int32_t* rowmap) {
    __shared__ int wsum[4];
    __shared__ int base;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = cu[b];
    __syncthreads();
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int s = s0 + threadIdx.x;
        const bool v = s < S && !mask[(size_t)b * S + s];
        const unsigned long long bal = __ballot(v);
        const int rank = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (v) rowmap[off + rank] = b * S + s;
        __syncthreads();
        if (threadIdx.x == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

// ---- batch preparation in one launch each -------------------------------------------------------------------------------
// pack_rows_fused_kernel: everything pa_pack_rows produces (valid-row counts, their prefix sums, the packed row -> position map
// and the batch elements by descending row count) from ONE block of 16 waves; wave w owns the batch rows w, w + 16, ...
__global__ __launch_bounds__(1024) void pack_rows_fused_kernel(const uint8_t* mask, int B, int S, int32_t* cu, int32_t* rowmap) {
    extern __shared__ int pk_s[];                    // [B] counts | [B + 1] offsets
    int* cnt = pk_s;
    int* off = pk_s + B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // mask bytes of a batch row: PK_CH x 64 columns per pass, ALL loads of a pass issued before the first ballot (clamped index,
    // surplus discarded): a load inside `s < S && !mask[..]` is a branch with a wait behind it - 16 dependent round trips per
    // row and pass at S = 1024, 28 us for this kernel (round 5: on the step's main stream since the prefetcher change)
    constexpr int PK_CH = 20;
    for (int b = wave; b < B; b += 16) {
        int c = 0;
        for (int c0 = 0; c0 < S; c0 += 64 * PK_CH) {
            uint8_t mv[PK_CH];
#pragma unroll
            for (int i = 0; i < PK_CH; ++i) { const int s = c0 + i * 64 + lane; mv[i] = mask[(size_t)b * S + (s < S ? s : S - 1)]; }
#pragma unroll
            for (int i = 0; i < PK_CH; ++i) { const int s = c0 + i * 64 + lane; c += __popcll(__ballot(s < S && !mv[i])); }
        }
        if (lane == 0) cnt[b] = c;
    }
    __syncthreads();
    if (wave == 0) {                                 // exclusive scan over the B counts, 64 at a time
        int carry = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const int v = b < B ? cnt[b] : 0;
            int inc = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
            if (b < B) off[b] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (lane == 0) off[B] = carry;
    }
    __syncthreads();
    for (int b = threadIdx.x; b <= B; b += 1024) cu[b] = off[b];
    for (int b = wave; b < B; b += 16) {
        int base = off[b];
        for (int c0 = 0; c0 < S; c0 += 64 * PK_CH) {
            uint8_t mv[PK_CH];
#pragma unroll
            for (int i = 0; i < PK_CH; ++i) { const int s = c0 + i * 64 + lane; mv[i] = mask[(size_t)b * S + (s < S ? s : S - 1)]; }
#pragma unroll
            for (int i = 0; i < PK_CH; ++i) {
                const int s = c0 + i * 64 + lane;
                const bool v = s < S && !mv[i];
                const unsigned long long bal = __ballot(v);
                if (v) rowmap[base + __popcll(bal & ((1ull << lane) - 1ull))] = b * S + s;
                base += __popcll(bal);
            }
        }
    }
    for (int i = threadIdx.x; i < B; i += 1024) {    // dispatch order: descending count, ties in batch order
        const int li = cnt[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) rank += (cnt[j] > li || (cnt[j] == li && j < i)) ? 1 : 0;
        cu[B + 1 + rank] = i;
    }
}

// group_rows_kernel: token rows grouped by embedding-table row for pa_embed_segment_bwd, one block per table - a STABLE
// counting sort (ties keep token order, so the segment sums add in a fixed order and a training run is reproducible):
// every wave owns a contiguous range of the tokens; per-(wave, id) counts by LDS atomics, an exclusive scan over the ids
// gives seg[], per-wave cursors give each wave its slots inside every segment, and a second pass places the tokens in
// order - the lanes of a 64-token batch that share an id are ranked with ballots.
struct GroupTab { pa_group_desc d[PA_MAX_GROUP_TABLES]; };
__device__ __forceinline__ int group_id(const pa_group_desc& g, int i, int& row) {
    if (g.kind == 0) {
        row = i;
        return (int)g.idx[g.rowmap ? g.rowmap[i] : i];
    }
    const int b = i / (g.T - 1), t1 = i - b * (g.T - 1);            // decoder row (b, t1 + 1) embeds token t1
    row = b * g.T + t1 + 1;
    if (g.kind == 1) return (int)g.idx[(size_t)b * g.tok_ld + t1];
    return g.kind == 2 ? t1 % g.dof : t1 / g.dof;
}
__global__ __launch_bounds__(1024) void group_rows_kernel(GroupTab tab) {
    extern __shared__ int gr_s[];                                   // [16][R] per-wave counts / cursors | [R + 1] seg | [16] wave sums
    const pa_group_desc& g = tab.d[blockIdx.x];
    const int R = g.rows, n = g.n;
    int* hist = gr_s;
    int* seg = gr_s + 16 * R;
    int* wsum = seg + R + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * R; i += 1024) hist[i] = 0;
    __syncthreads();
    const int chunk = ((n + 15) / 16 + 63) / 64 * 64;
    const int lo = wave * chunk, hi = min(n, lo + chunk);
    // The ids of this lane's tokens are fetched ONCE, all loads in flight together (idx[rowmap[i]] is two dependent round trips;
    // the two passes below then run from registers).  Up to GR_NIT batches per wave (16 * 64 * GR_NIT = 24 576 tokens per table:
    // B * S of every shipped config); longer tables re-fetch.  (Neutral for the launch's duration - 91.5 us on an idle GPU, most of
    // it the per-distinct-id loop the second pass used to run; 41.8 us with the per-bit ballots below.)
    constexpr int GR_NIT = 24;
    int idr[GR_NIT];
#pragma unroll
    for (int it = 0; it < GR_NIT; ++it) {
        const int i = lo + it * 64 + lane;
        int row;
        idr[it] = (i < hi) ? group_id(g, i, row) : -1;
    }
    auto id_of = [&](int it, int i, int& row) -> int {
        if (it < GR_NIT) {                      // (row is pure arithmetic on i; only the id needs memory)
            if (g.kind == 0) row = i;
            else { const int b = i / (g.T - 1); row = b * g.T + (i - b * (g.T - 1)) + 1; }
            int v = -1;
#pragma unroll
            for (int k = 0; k < GR_NIT; ++k) v = (k == it) ? idr[k] : v;
            return v;
        }
        return i < hi ? group_id(g, i, row) : -1;
    };
    for (int i0 = lo, it = 0; i0 < hi; i0 += 64, ++it) {
        const int i = i0 + lane;
        int row;
        const int id = id_of(it, i, row);
        if (id >= 0 && id < R) atomicAdd(&hist[wave * R + id], 1);
    }
    __syncthreads();
    // exclusive scan of the per-id totals: thread t handles ids t, t + 1024, ... (R <= 1024 in practice: one round)
    int carry = 0;
    for (int r0 = 0; r0 < R; r0 += 1024) {
        const int r = r0 + threadIdx.x;
        int tot = 0;
        if (r < R) for (int w = 0; w < 16; ++w) tot += hist[w * R + r];
        int inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int wbase = carry;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        if (r < R) {
            int run = wbase + inc - tot;                            // segment start
            seg[r] = run;
            for (int w = 0; w < 16; ++w) { const int c = hist[w * R + r]; hist[w * R + r] = run; run += c; }   // wave cursors
        }
        int total = 0;
        for (int w = 0; w < 16; ++w) total += wsum[w];
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) seg[R] = carry;
    __syncthreads();
    for (int r = threadIdx.x; r <= R; r += 1024) g.seg[r] = seg[r];
    const int idbits = R > 1 ? 32 - __clz(R - 1) : 0;               // ids are 0 .. R - 1
    for (int i0 = lo, it = 0; i0 < hi; i0 += 64, ++it) {
        const int i = i0 + lane;
        int row = 0;
        int id = id_of(it, i, row);
        if (i >= hi || id >= R) id = -1;
        // lanes of the batch that carry MY id, without a loop over the distinct ids (that loop - ~64 dependent rounds of shuffle,
        // ballot and LDS read per batch for value tokens - was most of this launch's 91 us): one ballot per id bit, each lane
        // keeps the lanes that agree with it on that bit.  The rank among them in lane order keeps the sort stable.
        unsigned long long eq = __ballot(id >= 0);
        for (int b = 0; b < idbits; ++b) {
            const bool bit = (id >> b) & 1;
            const unsigned long long bb = __ballot(bit);
            eq &= bit ? bb : ~bb;
        }
        if (id >= 0) {
            const int cur = hist[wave * R + id];                 // every lane reads the cursor before the leader below moves it
            const int rank = __popcll(eq & ((1ull << lane) - 1ull));
            g.order[cur + rank] = row;
            if (rank == 0) hist[wave * R + id] = cur + __popcll(eq);
        }
    }
}

// ================================================================================ LayerNorm
template <typename T, int NV>   // NV = ceil(d / 256): 4-wide vectors per lane per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(T* y, const T* z, const float* gamma, const float* beta,
                                                            float* mean, float* rstd, int64_t rows, int d, float eps,
                                                            bf16* img, int img_pat) {
    const int lane = threadIdx.x & 63;
    const int64_t row_ = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // (a wave past the last row works on the last row and stores nothing: an early `return` made the row count a lone
    //  kernel-argument load in front of all others - two scalar-memory round trips before the first row load)
    const bool live = row_ < rows;
    const int64_t row = live ? row_ : rows - 1;
    const T* zr = z + row * d;
    // every load of the row (and of gamma / beta) is issued up front, unconditionally (column clamped; lanes past d are
    // masked in the arithmetic)
    f32x4 v[NV], g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = min((lane + i * 64) << 2, d - 4);
        v[i] = ld4<T>(zr + c);
        g[i] = *reinterpret_cast<const f32x4*>(gamma + c);
        b[i] = *reinterpret_cast<const f32x4*>(beta + c);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (((lane + i * 64) << 2) < d) s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    const float mu = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (((lane + i * 64) << 2) < d) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float t = v[i][j] - mu; q += t * t; }
        }
    const float rs = 1.0f / sqrtf(wave_sum(q) / d + eps);
    if (!live) return;
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    T* yr = y + row * d;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * 64) << 2;
        if (c < d) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (v[i][j] - mu) * rs * g[i][j] + b[i][j];
            st4<T>(yr + c, o);
            if (std::is_same<T, float>::value && img) split_store4(img + row * 3 * d, d, c, o, img_pat);   // (bf16x3 mode: the consumer's cut)
        }
    }
}

#ifndef PA_LNB_ROWS
#define PA_LNB_ROWS 8
#endif
// (A/B builds, round 3, train step: 4 rows 5.17 ms, 8 rows 5.10, 16 rows 5.32 vs 5.26 on a slower box, 32 rows 5.43 vs 5.26)
constexpr int LNB_ROWS = PA_LNB_ROWS;    // rows per block in backward: one pair of rows per wave, ~1000 blocks at 7 940 rows (4 per CU in flight)

// dz = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-block partial column sums of dy * xhat (dgamma),
// dy (dbeta) and the (dropped) dz (dzsum) go to partial[block][3][d]; partial_finish_kernel adds them to the outputs.
// (Device-scope f32 atomics from every block measured 9x slower here: ~5 atomics/ns across the 8 XCDs.)
template <typename T, int NV>   // NV = ceil(d / 256): 4-wide vectors per lane per row
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(T* dz, T* ddrop, const T* dy, const T* z, const float* gamma,
                                                            const float* mean, const float* rstd, float* partial,
                                                            int64_t rows, int d, uint32_t drop_thr, float drop_scale,
                                                            uint32_t drop_seed, int want_dzsum) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][3][d]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[NV], ab[NV], as[NV];
    f32x4 gm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = ag[i]; as[i] = ag[i];
        const int c = (lane + i * 64) << 2;
        if (c < d) gm[i] = *reinterpret_cast<const f32x4*>(gamma + c);
    }
    const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
    constexpr int U = NV <= 2 ? 2 : 1;   // rows in flight per wave (register budget)
    for (int rr = wave; rr < LNB_ROWS; rr += 4 * U) {
        // U rows (rr, rr + 4, ..) per iteration: all loads of all rows are issued before the first reduction
        f32x4 zz[U][NV], dd[U][NV];
        float mu[U], rs[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mu[u] = 0.f; rs[u] = 0.f;
            const int64_t row = r0 + rr + 4 * u;
            ok[u] = (rr + 4 * u < LNB_ROWS) && row < rows;
            // unconditional loads (row and column clamped; the surplus is discarded through ok / c < d below): a branch
            // around them would put a wait between the loads of the two rows
            const int64_t rowc = row < rows ? row : rows - 1;
            mu[u] = mean[rowc]; rs[u] = rstd[rowc];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = min((lane + i * 64) << 2, d - 4);
                zz[u][i] = ld4<T>(z + rowc * d + c); dd[u][i] = ld4<T>(dy + rowc * d + c);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int64_t row = r0 + rr + 4 * u;
            f32x4 xh[NV], g[NV];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + i * 64) << 2;
                if (c < d) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        xh[i][j] = (zz[u][i][j] - mu[u]) * rs[u];
                        g[i][j] = dd[u][i][j] * gm[i][j];
                        s1 += g[i][j];
                        s2 += g[i][j] * xh[i][j];
                        ag[i][j] += dd[u][i][j] * xh[i][j];
                        ab[i][j] += dd[u][i][j];
                    }
                }
            }
            s1 = wave_sum(s1) / d;
            s2 = wave_sum(s2) / d;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (lane + i * 64) << 2;
                if (c < d) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = rs[u] * (g[i][j] - s1 - xh[i][j] * s2);
                    st4<T>(dz + row * d + c, o);
                    if (drop_thr) {
                        f32x4 od;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            od[j] = drop_keep_rc(drop_seed, (uint32_t)row, (uint32_t)(c + j), drop_thr) ? o[j] * drop_scale : 0.f;
                        st4<T>(ddrop + row * d + c, od);
                        as[i] += od;
                    } else {
                        as[i] += o;
                    }
                }
            }
        }
    }
    // cross-wave reduction of the column sums -> this block's partial row
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * 64) << 2;
        if (c < d) {
            *reinterpret_cast<f32x4*>(red + (wave * 3 + 0) * d + c) = ag[i];
            *reinterpret_cast<f32x4*>(red + (wave * 3 + 1) * d + c) = ab[i];
            *reinterpret_cast<f32x4*>(red + (wave * 3 + 2) * d + c) = as[i];
        }
    }
    __syncthreads();
    const int nq = want_dzsum ? 3 : 2;
    float* po = partial + (size_t)blockIdx.x * 3 * d;
    for (int e = threadIdx.x; e < nq * d; e += 256) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) sum += red[w * 3 * d + e];
        po[e] = sum;
    }
}

// d = 512 (the benchmarked width): every row of a wave's share is requested before the first one is reduced - a lane holds 8
// columns of a row (bf16: ONE 16-byte load per tensor and row, columns 8 lane ..; f32: two, columns 4 lane .. and 256 + 4 lane ..),
// LNB_ROWS / 4 rows per wave - so the stores of a row leave while the loads of the following rows are still arriving
// (the generic kernel above keeps two rows in flight and walks the rest one pair after the other).
template <typename T> struct LnRaw;
template <> struct LnRaw<bf16> {
    u32x4 v;
    __device__ __forceinline__ void load(const bf16* row, int lane) { v = *reinterpret_cast<const u32x4*>(row + lane * 8); }
    __device__ __forceinline__ float get(int j) const { return (j & 1) ? bf16_hi(v[j >> 1]) : bf16_lo(v[j >> 1]); }
    static __device__ __forceinline__ int col(int lane, int j) { return lane * 8 + j; }
    static __device__ __forceinline__ void store(bf16* row, int lane, const float* o) {
        u32x4 u;
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = pack_bf16(o[2 * k], o[2 * k + 1]);
        *reinterpret_cast<u32x4*>(row + lane * 8) = u;
    }
};
template <> struct LnRaw<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* row, int lane) {
        a = *reinterpret_cast<const f32x4*>(row + lane * 4); b = *reinterpret_cast<const f32x4*>(row + 256 + lane * 4);
    }
    __device__ __forceinline__ float get(int j) const { return j < 4 ? a[j] : b[j - 4]; }
    static __device__ __forceinline__ int col(int lane, int j) { return j < 4 ? lane * 4 + j : 256 + lane * 4 + (j - 4); }
    static __device__ __forceinline__ void store(float* row, int lane, const float* o) {
        *reinterpret_cast<f32x4*>(row + lane * 4) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(row + 256 + lane * 4) = f32x4{o[4], o[5], o[6], o[7]};
    }
};
template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd512_kernel(T* dz, T* ddrop, const T* dy, const T* z, const float* gamma,
                                                               const float* mean, const float* rstd, float* partial,
                                                               int64_t rows, uint32_t drop_thr, float drop_scale,
                                                               uint32_t drop_seed, int want_dzsum, bf16* img, int img_pat) {
    constexpr int D = 512, RW = LNB_ROWS / 4;                     // rows per wave
    static_assert(LNB_ROWS % 4 == 0, "whole rows per wave");
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
    LnRaw<T> zz[RW], dd[RW];
    float mu[RW], rs[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const int64_t row = r0 + wave + 4 * u, rowc = row < rows ? row : rows - 1;        // (clamped: no branch between the loads)
        zz[u].load(z + rowc * D, lane); dd[u].load(dy + rowc * D, lane);
        mu[u] = mean[rowc]; rs[u] = rstd[rowc];
    }
    float gm[8], ag[8], ab[8], as[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { gm[j] = gamma[LnRaw<T>::col(lane, j)]; ag[j] = 0.f; ab[j] = 0.f; as[j] = 0.f; }
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const int64_t row = r0 + wave + 4 * u;
        if (row >= rows) continue;                                   // (wave-uniform)
        float xh[8], g[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d_ = dd[u].get(j);
            xh[j] = (zz[u].get(j) - mu[u]) * rs[u];
            g[j] = d_ * gm[j];
            s1 += g[j]; s2 += g[j] * xh[j];
            ag[j] += d_ * xh[j]; ab[j] += d_;
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rs[u] * (g[j] - s1 - xh[j] * s2);
        LnRaw<T>::store(dz + row * D, lane, o);
        if (drop_thr) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                o[j] = drop_keep_rc(drop_seed, (uint32_t)row, (uint32_t)LnRaw<T>::col(lane, j), drop_thr) ? o[j] * drop_scale : 0.f;
            LnRaw<T>::store(ddrop + row * D, lane, o);
        }
        if (std::is_same<T, float>::value && img) {                  // bf16x3 mode: the cut image of the dX GEMM's operand (ddrop, or dz without dropout)
            split_store4(img + row * 3 * D, D, lane * 4, f32x4{o[0], o[1], o[2], o[3]}, img_pat);
            split_store4(img + row * 3 * D, D, 256 + lane * 4, f32x4{o[4], o[5], o[6], o[7]}, img_pat);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) as[j] += o[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = LnRaw<T>::col(lane, j);
        red[(wave * 3 + 0) * D + c] = ag[j]; red[(wave * 3 + 1) * D + c] = ab[j]; red[(wave * 3 + 2) * D + c] = as[j];
    }
    __syncthreads();
    const int nq = want_dzsum ? 3 : 2;
    float* po = partial + (size_t)blockIdx.x * 3 * D;
    for (int e = threadIdx.x * 4; e < nq * D; e += 1024) {
        f32x4 sum = *reinterpret_cast<const f32x4*>(red + e);
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const f32x4*>(red + w * 3 * D + e);
        *reinterpret_cast<f32x4*>(po + e) = sum;
    }
}

// sums `nparts` partial rows and ACCUMULATES into up to three outputs.  Block = 64 columns x 4 part lanes
// (each lane strides over the parts with independent, coalesced loads), LDS combine.
__global__ __launch_bounds__(256) void partial_finish_kernel(const float* partial, int nparts, int part_stride, int q_stride,
                                                             int ncols, float* o0, float* o1, float* o2) {
    __shared__ float red[256];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    const int qn = blockIdx.y;
    float* o = qn == 0 ? o0 : (qn == 1 ? o1 : o2);
    float s = 0.f;
    if (c < ncols && o) {
        const float* p = partial + (size_t)qn * q_stride + c;
#pragma unroll 8
        for (int i = pl; i < nparts; i += 4) s += p[(size_t)i * part_stride];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c < ncols && o) o[c] += (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// the same for several LayerNorm backward passes at once (blockIdx.z = descriptor)
struct LnFinishTab { pa_ln_finish_desc d[PA_MAX_LN_FINISH]; int n; };
__global__ __launch_bounds__(256) void partial_finish_many_kernel(LnFinishTab t, int ncols) {
    __shared__ float red[256];
    const pa_ln_finish_desc d = t.d[blockIdx.z];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
    const int qn = blockIdx.y;
    float* o = qn == 0 ? d.dgamma : (qn == 1 ? d.dbeta : d.dzsum);
    float s = 0.f;
    if (c < ncols && o) {
        const float* p = d.partial + (size_t)qn * ncols + c;
#pragma unroll 8
        for (int i = pl; i < d.nparts; i += 4) s += p[(size_t)i * 3 * ncols];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c < ncols && o) o[c] += (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// ================================================================================ switch head
template <typename T>
__global__ __launch_bounds__(256) void switch_fwd_kernel(float* s, const T* h, const float* w, const float* b,
                                                         int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float acc = 0.f;
    for (int c = lane << 2; c < d; c += 256) {
        const f32x4 hv = ld4<T>(h + row * d + c);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
        acc += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
    }
    acc = wave_sum(acc);
    if (lane == 0) s[row] = acc + b[0];
}

// dh (+)= ds * w ; partial[blk][0][d] = sum_rows ds*h ; partial[blk][1][0] = sum_rows ds
template <typename T>
__global__ __launch_bounds__(256) void switch_bwd_kernel(T* dh, int accumulate, const float* ds, const T* h,
                                                         const float* w, float* partial, int64_t rows, int d) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [4][d] + [4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 aw[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) aw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float adb = 0.f;
    const int64_t r0 = (int64_t)blockIdx.x * LNB_ROWS;
    for (int rr = wave; rr < LNB_ROWS; rr += 4) {
        const int64_t row = r0 + rr;
        if (row >= rows) break;
        const float g = ds[row];
        adb += g;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (lane + i * 64) << 2;
            if (c < d) {
                const f32x4 hv = ld4<T>(h + row * d + c);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) { aw[i][j] += g * hv[j]; o[j] = g * wv[j]; }
                if (accumulate) o += ld4<T>(dh + row * d + c);
                st4<T>(dh + row * d + c, o);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (lane + i * 64) << 2;
        if (c < d) *reinterpret_cast<f32x4*>(red + wave * d + c) = aw[i];
    }
    if (lane == 0) red[4 * d + wave] = adb;
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256)
        partial[((size_t)blockIdx.x * 2 + 0) * d + c] = red[c] + red[d + c] + red[2 * d + c] + red[3 * d + c];
    if (threadIdx.x == 0)
        partial[((size_t)blockIdx.x * 2 + 1) * d] = red[4 * d] + red[4 * d + 1] + red[4 * d + 2] + red[4 * d + 3];
}

// ================================================================================ mixture NLL
// One wave per (b, i) row.  Row = [vocab logits (V)] ++ [pointer logits (T), j >= i replaced by 1e-6].
struct RowStat { float m, s; int arg; };
__device__ __forceinline__ void online(RowStat& st, float x, int idx) {
    if (x > st.m) { st.s = st.s * expf(st.m - x) + 1.f; st.m = x; st.arg = idx; }
    else st.s += expf(x - st.m);
}
__device__ __forceinline__ RowStat wave_merge(RowStat a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(a.m, o), s2 = __shfl_xor(a.s, o);
        const int g2 = __shfl_xor(a.arg, o);
        const float mn = fmaxf(a.m, m2);
        const float sa = (a.m == -INFINITY) ? 0.f : a.s * expf(a.m - mn);
        const float sb = (m2 == -INFINITY) ? 0.f : s2 * expf(m2 - mn);
        // first-max tie break: smaller index wins on equal value
        if (m2 > a.m || (m2 == a.m && g2 < a.arg)) a.arg = g2;
        a.m = mn; a.s = sa + sb;
    }
    return a;
}

constexpr int NLL_ROWS = 4;      // rows per block (one per wave); the three statistics leave a block as ONE atomic each
constexpr int NLL_VSLOTS = 16;   // a lane holds up to 16 vocabulary logits (V <= 1024) / 4 pointer logits (T <= 256) of its row:
constexpr int NLL_PSLOTS = 4;    // all loads are issued up front, unconditionally (clamped index), then reduced in order
__global__ __launch_bounds__(256) void mixture_nll_fwd_kernel(float* stats, float* row_lse, const float* vocab, int ldv,
                                                              const float* ptr, const float* sw, const int64_t* label,
                                                              int B, int Tn, int V, int pad) {
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a_nll = 0.f, a_cnt = 0.f, a_hit = 0.f;
    for (int rr = wave; rr < NLL_ROWS; rr += 4) {
        const int64_t row = (int64_t)blockIdx.x * NLL_ROWS + rr;
        if (row >= (int64_t)B * Tn) break;
        const int i = (int)(row % Tn);
        const float* vr = vocab + row * ldv;
        const float* pr = ptr + row * Tn;
        RowStat sv{-INFINITY, 0.f, 0x7fffffff}, sp{-INFINITY, 0.f, 0x7fffffff};
        if (V <= 64 * NLL_VSLOTS && Tn <= 64 * NLL_PSLOTS) {
            float xv[NLL_VSLOTS], xp[NLL_PSLOTS];
#pragma unroll
            for (int q = 0; q < NLL_VSLOTS; ++q) xv[q] = vr[min(lane + 64 * q, V - 1)];
#pragma unroll
            for (int q = 0; q < NLL_PSLOTS; ++q) xp[q] = pr[min(lane + 64 * q, Tn - 1)];
#pragma unroll
            for (int q = 0; q < NLL_VSLOTS; ++q) { const int k = lane + 64 * q; if (k < V) online(sv, xv[q], k); }
#pragma unroll
            for (int q = 0; q < NLL_PSLOTS; ++q) { const int j = lane + 64 * q; if (j < Tn) online(sp, (j >= i) ? 1e-6f : xp[q], j); }
        } else {
            for (int k = lane; k < V; k += 64) online(sv, vr[k], k);
            for (int j = lane; j < Tn; j += 64) online(sp, (j >= i) ? 1e-6f : pr[j], j);
        }
        sv = wave_merge(sv);
        sp = wave_merge(sp);
        const float lse_v = sv.m + logf(sv.s), lse_p = sp.m + logf(sp.s);
        const float prob = 1.0f / (1.0f + expf(-sw[row]));
        const float lv = logf(fmaxf(1.0f - prob, 1e-6f)), lp = logf(fmaxf(prob, 1e-6f));
        if (lane == 0) {
            row_lse[row * 2] = lse_v; row_lse[row * 2 + 1] = lse_p;
            const int64_t lab = label[row];
            if (lab != pad) {
                float logp;
                if (lab < V) logp = vr[lab] - lse_v + lv;
                else { const int j = (int)(lab - V); logp = ((j >= i) ? 1e-6f : pr[j]) - lse_p + lp; }
                const float best_v = sv.m - lse_v + lv, best_p = sp.m - lse_p + lp;
                const int64_t pred = (best_p > best_v) ? (int64_t)V + sp.arg : (int64_t)sv.arg;
                a_nll -= logp; a_cnt += 1.0f;
                if (pred == lab) a_hit += 1.0f;
            }
        }
    }
    if (lane == 0) { red[wave][0] = a_nll; red[wave][1] = a_cnt; red[wave][2] = a_hit; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (v != 0.f) atomicAdd(stats + threadIdx.x, v);
    }
}
// stats8 form (pa_mixture_nll_fwd_fin): loss = nll / count, accuracy = hits / (count + 1e-10) (reference models.py:226-231) and the
// upstream gradient's default 1.0 in ONE one-wave launch behind the forward kernel - what five tiny torch launches and a memset did
// on the step's serial chain before.  Two in-kernel forms were measured first and lost: the last block (ticket)
// finishing the sums behind __threadfence() took the forward kernel from 11.7 to 38.6 us, and with returning device-scope atomics
// instead of the fence to 33.8 us (512 blocks x 4 returning atomics on four words) - more than the launches they replaced.
__global__ void mixture_nll_finish_kernel(float* stats) {
    if (threadIdx.x == 0) {
        const float s0 = stats[0], s1 = stats[1], s2 = stats[2];
        stats[4] = s0 / s1; stats[5] = s2 / (s1 + 1e-10f); stats[3] = 1.0f;
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void mixture_nll_bwd_kernel(TO* dvocab, TO* dptr, float* dsw, const float* stats,
                                                              const float* row_lse, const float* vocab, int ldv,
                                                              const float* ptr, const float* sw, const int64_t* label,
                                                              int B, int Tn, int V, int pad, float gscale, const float* upstream) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (int64_t)B * Tn) return;
    const int i = (int)(row % Tn);
    const int64_t lab = label[row];
    TO* dv = dvocab + row * ldv;
    TO* dp = dptr + row * Tn;
    if (lab == pad) {
        for (int k = lane; k < ldv; k += 64) st1<TO>(dv + k, 0.f);
        for (int j = lane; j < Tn; j += 64) st1<TO>(dp + j, 0.f);
        if (lane == 0) dsw[row] = 0.f;
        return;
    }
    const float g = gscale * (upstream ? *upstream : stats[3]) / stats[1];      // upstream d(loss), device resident (stats[3] unless given)
    const float prob = 1.0f / (1.0f + expf(-sw[row]));
    if (lab < V) {
        const float lse_v = row_lse[row * 2];
        const float* vr = vocab + row * ldv;
        for (int k = lane; k < ldv; k += 64)
            st1<TO>(dv + k, (k < V) ? g * (expf(vr[k] - lse_v) - (k == lab ? 1.f : 0.f)) : 0.f);
        for (int j = lane; j < Tn; j += 64) st1<TO>(dp + j, 0.f);
        if (lane == 0) dsw[row] = (1.0f - prob >= 1e-6f) ? g * prob : 0.f;
    } else {
        const float lse_p = row_lse[row * 2 + 1];
        const float* pr = ptr + row * Tn;
        const int js = (int)(lab - V);
        for (int k = lane; k < ldv; k += 64) st1<TO>(dv + k, 0.f);
        for (int j = lane; j < Tn; j += 64)
            st1<TO>(dp + j, (j < i) ? g * (expf(pr[j] - lse_p) - (j == js ? 1.f : 0.f)) : 0.f);
        if (lane == 0) dsw[row] = (prob >= 1e-6f) ? -g * (1.0f - prob) : 0.f;
    }
}

// ================================================================================ Adam / cast
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, bf16* pb, int64_t n,
                                                   float step_size, float b1, float b2, float eps, float inv_sqrt_bc2,
                                                   float gscale) {
    const int64_t n4 = n >> 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp = *reinterpret_cast<f32x4*>(p + e * 4);
        f32x4 gg = *reinterpret_cast<const f32x4*>(g + e * 4);
        f32x4 mm = *reinterpret_cast<f32x4*>(m + e * 4);
        f32x4 vv = *reinterpret_cast<f32x4*>(v + e * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = gg[j] * gscale;
            mm[j] = b1 * mm[j] + (1.f - b1) * gj;
            vv[j] = b2 * vv[j] + (1.f - b2) * gj * gj;
            pp[j] -= step_size * mm[j] / (sqrtf(vv[j]) * inv_sqrt_bc2 + eps);
        }
        *reinterpret_cast<f32x4*>(p + e * 4) = pp;
        *reinterpret_cast<f32x4*>(m + e * 4) = mm;
        *reinterpret_cast<f32x4*>(v + e * 4) = vv;
        if (pb) st4<bf16>(pb + e * 4, pp);
    }
    // tail (n % 4)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const float gj = g[e] * gscale;
        const float mj = b1 * m[e] + (1.f - b1) * gj;
        const float vj = b2 * v[e] + (1.f - b2) * gj * gj;
        m[e] = mj; v[e] = vj;
        p[e] -= step_size * mj / (sqrtf(vj) * inv_sqrt_bc2 + eps);
        if (pb) pb[e] = (bf16)p[e];
    }
}

template <typename TD, typename TS>
__global__ __launch_bounds__(256) void cast_kernel(TD* dst, const TS* src, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        st1<TD>(dst + e, ld1<TS>(src + e));
}

inline int grid_for(int64_t work_items, int per_block = 256, int cap = 4096) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b > cap ? cap : b);
}

}  // namespace

#define ST(s) reinterpret_cast<hipStream_t>(s)

extern "C" int pa_version(void) { return 1; }

extern "C" int pa_embed_input_fwd(void* out, int32_t out_dtype, const float* const* tables, const int64_t* const* idx,
                                  const int32_t* rowmap, int32_t n_tables, int64_t n_tok, int32_t d, void* stream) {
    if (!out || !tables || !idx || n_tables < 1 || n_tables > 5 || (d & 3) || n_tok <= 0) return PA_EINVAL;
    EmbTabs tb; tb.n = n_tables; tb.rowmap = rowmap;
    for (int k = 0; k < 5; ++k) { tb.t[k] = k < n_tables ? tables[k] : nullptr; tb.idx[k] = k < n_tables ? idx[k] : nullptr; }
    const int grid = grid_for(n_tok * (d >> 2));
    if (out_dtype == PA_BF16) PA_LAUNCH(embed_input_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (bf16*)out, tb, n_tok, d);
    else PA_LAUNCH(embed_input_fwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (float*)out, tb, n_tok, d);
    return 0;
}

// Table gradients by sorted segments: the token rows were grouped by table row once when the batch was prepared
// (order = rows sorted by id, seg[r] .. seg[r+1] = the rows that use table row r).  Sums run in registers, 128 lanes x 4
// columns, four token rows in flight.
struct SegTab { float* t[PA_MAX_SEG_TABLES]; const int32_t* order[PA_MAX_SEG_TABLES]; const int32_t* seg[PA_MAX_SEG_TABLES];
                int rows[PA_MAX_SEG_TABLES]; int ch[PA_MAX_SEG_TABLES]; int begin[PA_MAX_SEG_TABLES + 1]; int n; };
constexpr int SEG_CHUNK = 128;
constexpr int SEG_SPLIT = 8, SEG_LONG = 64;
template <typename T>
__device__ __forceinline__ f32x4 seg_sum(const T* dout, const int32_t* order, int lo, int hi, int d, int c) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int i = lo;
    for (; i + 8 <= hi; i += 8) {                 // eight independent index loads, then eight row loads in flight
        int o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = order[i + j];
        f32x4 g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = ld4<T>(dout + (int64_t)o[j] * d + c);
        acc += ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
    }
    if (i < hi) {
        int o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = order[min(i + j, hi - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const f32x4 g = ld4<T>(dout + (int64_t)o[j] * d + c); if (i + j < hi) acc += g; }
    }
    return acc;
}
// ch[k] == 1: one block per table row (plain read-modify-write, no atomics) - tables with many rows / short segments.
// ch[k] == 0: one block per SEG_CHUNK consecutive entries of the sorted order, whatever rows they belong to (a table with a
// handful of rows has segments of thousands of tokens); per overlapped row one atomic per column.
template <typename T>
__global__ __launch_bounds__(128) void embed_segment_bwd_kernel(const T* dout, SegTab tb, int d, int seg_long) {
    int k = 0;
    while (k + 1 < tb.n && (int)blockIdx.x >= tb.begin[k + 1]) ++k;
    const int rel = blockIdx.x - tb.begin[k];
    const int32_t* order = tb.order[k];
    const int32_t* seg = tb.seg[k];
    if (tb.ch[k] == 1) {
        // SEG_SPLIT blocks per table row: a row used by at most SEG_LONG tokens is summed by its first block alone (plain
        // read-modify-write); a heavily used row (skewed token distributions: thousands of tokens on one value) is spread
        // over all of them, which then combine with atomics.
        const int r = rel / SEG_SPLIT, j = rel - r * SEG_SPLIT;
        const int b0 = seg[r], b1 = seg[r + 1], len = b1 - b0;
        if (len <= 0 || (len <= seg_long && j > 0)) return;      // (ordered mode: seg_long = INT_MAX, block 0 sums the whole row)
        const bool multi = len > seg_long;
        const int per = multi ? (len + SEG_SPLIT - 1) / SEG_SPLIT : len;
        const int lo = b0 + j * per, hi = min(b1, lo + per);
        if (lo >= hi) return;
        for (int c = threadIdx.x << 2; c < d; c += 512) {
            float* dst = tb.t[k] + (int64_t)r * d + c;
            const f32x4 acc = seg_sum<T>(dout, order, lo, hi, d, c);
            if (!multi) *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + acc;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, acc[e]);
            }
        }
    } else {
        const int c0 = rel * SEG_CHUNK, c1 = min(c0 + SEG_CHUNK, seg[tb.rows[k]]);
        if (c0 >= c1) return;
        for (int r = 0; r < tb.rows[k]; ++r) {
            const int lo = max(seg[r], c0), hi = min(seg[r + 1], c1);
            if (lo >= hi) continue;
            for (int c = threadIdx.x << 2; c < d; c += 512) {
                const f32x4 acc = seg_sum<T>(dout, order, lo, hi, d, c);
                float* dst = tb.t[k] + (int64_t)r * d + c;
#pragma unroll
                for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, acc[e]);
            }
        }
    }
}
// ordered (f32 parity path): one block per table row and no atomics, as embed_segment_bwd_kernel's ch == 1 branch with
// seg_long = INT_MAX - but FOUR thread groups walk four consecutive quarters of a long segment (each in order, eight rows in
// flight) and group 0 adds the four sums in quarter order: still one fixed order of additions per table row, a quarter of the
// serial chain (the type / position tables put thousands of tokens on one row: 197 us per launch on one group).
template <typename T>
__global__ __launch_bounds__(512) void embed_segment_ordered_kernel(const T* dout, SegTab tb, int d) {
    extern __shared__ __attribute__((aligned(16))) float seg_part[];       // [3][d]
    int k = 0;
    while (k + 1 < tb.n && (int)blockIdx.x >= tb.begin[k + 1]) ++k;
    const int rel = blockIdx.x - tb.begin[k];
    const int r = rel / SEG_SPLIT;
    if (rel - r * SEG_SPLIT) return;                              // (the grid keeps embed_segment_bwd_kernel's block table)
    const int32_t* order = tb.order[k];
    const int b0 = tb.seg[k][r], b1 = tb.seg[k][r + 1], len = b1 - b0;
    if (len <= 0) return;
    const int g = threadIdx.x >> 7, t = threadIdx.x & 127;
    const bool par = len >= 32;                                   // (block-uniform) short rows: group 0 alone
    const int per = par ? ((len + 3) / 4 + 7) / 8 * 8 : len;
    const int lo = b0 + g * per, hi = min(b1, lo + per);
    // (block-uniform trip count: every thread reaches both barriers of every pass, threads past the row's width idle - ADVICE r4)
    for (int c0 = 0; c0 < d; c0 += 512) {
        const int c = c0 + (t << 2);
        const bool on = c < d;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (on && lo < hi && (par || g == 0)) acc = seg_sum<T>(dout, order, lo, hi, d, c);
        if (par) {
            if (on && g) *reinterpret_cast<f32x4*>(seg_part + (g - 1) * d + c) = acc;
            __syncthreads();
            if (on && g == 0) acc = ((acc + *reinterpret_cast<const f32x4*>(seg_part + c)) + *reinterpret_cast<const f32x4*>(seg_part + d + c))
                              + *reinterpret_cast<const f32x4*>(seg_part + 2 * d + c);
            __syncthreads();
        }
        if (on && g == 0) {
            float* dst = tb.t[k] + (int64_t)r * d + c;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(dst) + acc;
        }
    }
}
extern "C" int pa_embed_segment_bwd(const void* dout, int32_t dtype, float* const* dtables, const int32_t* const* order,
                                    const int32_t* const* seg, const int32_t* table_rows, int32_t n_tables, int64_t n_rows,
                                    int32_t d, void* stream) {
    if (!dout || !dtables || !order || !seg || !table_rows || n_tables < 1 || n_tables > PA_MAX_SEG_TABLES || (d & 3) || n_rows <= 0) return PA_EINVAL;
    // ordered (f32 parity path): every table row is summed by ONE block walking its whole segment in order - no atomics
    const bool ordered = pa_ordered_reductions(dtype);
    const int seg_long = ordered ? 0x7fffffff : SEG_LONG;
    SegTab tb; tb.n = n_tables; tb.begin[0] = 0;
    for (int k = 0; k < n_tables; ++k) {
        tb.t[k] = dtables[k]; tb.order[k] = order[k]; tb.seg[k] = seg[k]; tb.rows[k] = table_rows[k];
        if (!tb.t[k] || !tb.order[k] || !tb.seg[k] || tb.rows[k] <= 0) return PA_EINVAL;
        tb.ch[k] = (ordered || tb.rows[k] > 64) ? 1 : 0;
        tb.begin[k + 1] = tb.begin[k] + (tb.ch[k] ? tb.rows[k] * SEG_SPLIT : (int)((n_rows + SEG_CHUNK - 1) / SEG_CHUNK));
    }
    static const bool ord4 = !(getenv("PA_EMBED_ORDERED_GROUPS") && atoi(getenv("PA_EMBED_ORDERED_GROUPS")) == 1);
    if (ordered && ord4 && d <= 2048) {
        const size_t shm = (size_t)3 * d * sizeof(float);
        if (dtype == PA_BF16) PA_LAUNCH(embed_segment_ordered_kernel<bf16>, dim3(tb.begin[n_tables]), dim3(512), shm, ST(stream), (const bf16*)dout, tb, d);
        else PA_LAUNCH(embed_segment_ordered_kernel<float>, dim3(tb.begin[n_tables]), dim3(512), shm, ST(stream), (const float*)dout, tb, d);
        return 0;
    }
    if (dtype == PA_BF16) PA_LAUNCH(embed_segment_bwd_kernel<bf16>, dim3(tb.begin[n_tables]), dim3(128), 0, ST(stream), (const bf16*)dout, tb, d, seg_long);
    else PA_LAUNCH(embed_segment_bwd_kernel<float>, dim3(tb.begin[n_tables]), dim3(128), 0, ST(stream), (const float*)dout, tb, d, seg_long);
    return 0;
}

extern "C" int pa_embed_input_bwd(const void* dout, int32_t dtype, float* const* dtables, const int64_t* const* idx,
                                  const int32_t* rowmap, const int32_t* table_rows, int32_t n_tables, int64_t n_tok, int32_t d,
                                  void* stream) {
    if (!dout || !dtables || !idx || !table_rows || n_tables < 1 || n_tables > 5 || (d & 3) || n_tok <= 0) return PA_EINVAL;
    if (d > 1024) return PA_ESHAPE;
    EmbGrads tb; tb.n = n_tables; tb.rowmap = rowmap;
    int nsmall = 0;
    for (int k = 0; k < 5; ++k) {
        tb.t[k] = k < n_tables ? dtables[k] : nullptr; tb.idx[k] = k < n_tables ? idx[k] : nullptr;
        tb.rows[k] = k < n_tables ? table_rows[k] : 0;
        if (tb.idx[k] && tb.rows[k] <= EMB_SMALL_ROWS) ++nsmall;
    }
    const int grid = (int)((n_tok + EMB_TOK_PER_BLOCK - 1) / EMB_TOK_PER_BLOCK);
    const size_t shm = (size_t)nsmall * EMB_SMALL_ROWS * d * sizeof(float);
    if (dtype == PA_BF16) PA_LAUNCH(embed_input_bwd_kernel<bf16>, dim3(grid), dim3(256), shm, ST(stream), (const bf16*)dout, tb, n_tok, d);
    else PA_LAUNCH(embed_input_bwd_kernel<float>, dim3(grid), dim3(256), shm, ST(stream), (const float*)dout, tb, n_tok, d);
    return 0;
}

extern "C" int pa_embed_output_fwd(void* out, int32_t out_dtype, const float* value, const float* coord, const float* pos,
                                   const int64_t* tok, int32_t tok_ld, int32_t B, int32_t T, int32_t d, int32_t dof,
                                   void* stream) {
    if (!out || !value || !coord || !pos || !tok || (d & 3) || B <= 0 || T <= 0 || dof <= 0) return PA_EINVAL;
    const int grid = grid_for((int64_t)B * T * (d >> 2));
    if (out_dtype == PA_BF16) PA_LAUNCH(embed_output_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (bf16*)out, value, coord, pos, tok, tok_ld, B, T, d, dof);
    else PA_LAUNCH(embed_output_fwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (float*)out, value, coord, pos, tok, tok_ld, B, T, d, dof);
    return 0;
}

extern "C" int pa_embed_output_bwd(const void* dout, int32_t dtype, float* dvalue, float* dcoord, float* dpos,
                                   const int64_t* tok, int32_t tok_ld, int32_t B, int32_t T, int32_t d, int32_t dof,
                                   void* stream) {
    if (!dout || !dvalue || !dcoord || !dpos || !tok || (d & 3) || B <= 0 || T <= 0 || dof <= 0) return PA_EINVAL;
    const int grid = grid_for((int64_t)B * T * (d >> 2));
    if (dtype == PA_BF16) PA_LAUNCH(embed_output_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (const bf16*)dout, dvalue, dcoord, dpos, tok, tok_ld, B, T, d, dof);
    else PA_LAUNCH(embed_output_bwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (const float*)dout, dvalue, dcoord, dpos, tok, tok_ld, B, T, d, dof);
    return 0;
}

extern "C" int pa_pack_rows(const uint8_t* mask, int32_t B, int32_t S, int32_t* cu, int32_t* rowmap, void* stream) {
    if (!mask || !cu || !rowmap || B <= 0 || S <= 0) return PA_EINVAL;
    if (B <= 1024) {                                  // one launch (the batch sizes of every shipped config)
        PA_LAUNCH(pack_rows_fused_kernel, dim3(1), dim3(1024), (size_t)(2 * B + 1) * sizeof(int), ST(stream), mask, B, S, cu, rowmap);
        return 0;
    }
    // cu[B+1 .. 2B] doubles as scratch for the per-row counts (caller allocates 2B+1 ints)
    PA_LAUNCH(pack_count_kernel, dim3(B), dim3(256), 0, ST(stream), mask, S, cu + B + 1);
    PA_LAUNCH(pack_scan_kernel, dim3(1), dim3(64), 0, ST(stream), cu + B + 1, B, cu);
    PA_LAUNCH(pack_fill_kernel, dim3(B), dim3(256), 0, ST(stream), mask, S, cu, rowmap);
    if (B <= 8192) PA_LAUNCH(pack_order_kernel, dim3(1), dim3(256), (size_t)B * sizeof(int), ST(stream), cu + B + 1, B);
    else PA_LAUNCH(pack_iota_kernel, dim3((B + 255) / 256), dim3(256), 0, ST(stream), cu + B + 1, B);
    return 0;
}

extern "C" int pa_group_rows(const pa_group_desc* descs, int32_t n_tables, void* stream) {
    if (!descs || n_tables < 1 || n_tables > PA_MAX_GROUP_TABLES) return PA_EINVAL;
    GroupTab tab;
    int rmax = 0;
    for (int k = 0; k < n_tables; ++k) {
        const pa_group_desc& g = descs[k];
        if (!g.order || !g.seg || g.n <= 0 || g.rows <= 0 || g.kind < 0 || g.kind > 3) return PA_EINVAL;
        if (g.kind <= 1 && !g.idx) return PA_EINVAL;
        if (g.kind >= 1 && (g.T < 2 || g.dof < 1 || g.n % (g.T - 1))) return PA_EINVAL;
        if (g.rows > 2048) return PA_ESHAPE;                       // 16 x rows cursors live in LDS
        tab.d[k] = g;
        rmax = g.rows > rmax ? g.rows : rmax;
    }
    const size_t shm = (size_t)(16 * rmax + rmax + 1 + 16) * sizeof(int);
    if (shm > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(group_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        if (e != hipSuccess) return (int)e;
    }
    PA_LAUNCH(group_rows_kernel, dim3(n_tables), dim3(1024), shm, ST(stream), tab);
    return 0;
}

extern "C" int64_t pa_layernorm_ws_floats(int64_t rows, int32_t d) {
    return ((rows + LNB_ROWS - 1) / LNB_ROWS) * 3 * (int64_t)d;
}

extern "C" int pa_layernorm_fwd_img(void* y, const void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                                    int64_t rows, int32_t d, float eps, int32_t dtype, void* img, int32_t img_pat, void* stream);
extern "C" int pa_layernorm_fwd(void* y, const void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                                int64_t rows, int32_t d, float eps, int32_t dtype, void* stream) {
    return pa_layernorm_fwd_img(y, z, gamma, beta, mean, rstd, rows, d, eps, dtype, nullptr, 0, stream);
}
extern "C" int pa_layernorm_fwd_img(void* y, const void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                                    int64_t rows, int32_t d, float eps, int32_t dtype, void* img_, int32_t img_pat, void* stream) {
    if (!y || !z || !gamma || !beta || !mean || !rstd || rows <= 0 || (d & 3) || d > 256 * MAXV) return PA_EINVAL;
    if (img_ && (dtype != PA_F32 || (reinterpret_cast<uintptr_t>(img_) & 7))) return PA_EINVAL;
    bf16* img = static_cast<bf16*>(img_);
    const int grid = (int)((rows + 3) / 4);
#define LNF_GO(T_, NV_) PA_LAUNCH((layernorm_fwd_kernel<T_, NV_>), dim3(grid), dim3(256), 0, ST(stream), (T_*)y, (const T_*)z, gamma, beta, mean, rstd, rows, d, eps, img, img_pat)
#define LNF_NV(T_) do { if (d <= 256) LNF_GO(T_, 1); else if (d <= 512) LNF_GO(T_, 2); else if (d <= 1024) LNF_GO(T_, 4); else LNF_GO(T_, 8); } while (0)
    if (dtype == PA_BF16) LNF_NV(bf16); else LNF_NV(float);
#undef LNF_NV
#undef LNF_GO
    return 0;
}

extern "C" int pa_layernorm_bwd_partial_img(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                                            const float* mean, const float* rstd, int32_t want_dzsum, float* partial,
                                            int64_t rows, int32_t d, int32_t dtype, float drop_p, uint32_t drop_seed,
                                            void* img, int32_t img_pat, void* stream);
extern "C" int pa_layernorm_bwd_partial(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                                        const float* mean, const float* rstd, int32_t want_dzsum, float* partial,
                                        int64_t rows, int32_t d, int32_t dtype, float drop_p, uint32_t drop_seed, void* stream) {
    return pa_layernorm_bwd_partial_img(dz, ddrop, dy, z, gamma, mean, rstd, want_dzsum, partial, rows, d, dtype, drop_p, drop_seed,
                                        nullptr, 0, stream);
}
// 1 when pa_layernorm_bwd_partial_img can write the bf16x3 image of its output for these arguments (the d = 512 f32 kernel)
extern "C" int pa_layernorm_bwd_can_img(int32_t d, int32_t dtype) {
    static const bool v512 = !(getenv("PA_LNB_512") && atoi(getenv("PA_LNB_512")) == 0);
    return (d == 512 && dtype == PA_F32 && v512) ? 1 : 0;
}
extern "C" int pa_layernorm_bwd_partial_img(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                                            const float* mean, const float* rstd, int32_t want_dzsum, float* partial,
                                            int64_t rows, int32_t d, int32_t dtype, float drop_p, uint32_t drop_seed,
                                            void* img_, int32_t img_pat, void* stream) {
    if (!dz || !dy || !z || !gamma || !mean || !rstd || !partial) return PA_EINVAL;
    bf16* img = static_cast<bf16*>(img_);
    if (rows <= 0 || (d & 3) || d > 256 * MAXV || drop_p < 0.f || drop_p >= 1.f) return PA_EINVAL;
    const uint32_t thr = (uint32_t)((double)drop_p * 4294967296.0);        // as pa_gemm: keep <=> 32-bit product >= thr (drop_keep_rc)
    if (thr && !ddrop) return PA_EINVAL;
    const float scale = (float)(1.0 / (1.0 - (double)thr / 4294967296.0));
    const int grid = (int)((rows + LNB_ROWS - 1) / LNB_ROWS);
    const size_t shm = (size_t)4 * 3 * d * sizeof(float);
#define LNB_GO(T_, NV_) PA_LAUNCH((layernorm_bwd_kernel<T_, NV_>), dim3(grid), dim3(256), shm, ST(stream), (T_*)dz, (T_*)ddrop, \
        (const T_*)dy, (const T_*)z, gamma, mean, rstd, partial, rows, d, thr, scale, drop_seed, want_dzsum ? 1 : 0)
#define LNB_NV(T_) do { if (d <= 256) LNB_GO(T_, 1); else if (d <= 512) LNB_GO(T_, 2); else if (d <= 1024) LNB_GO(T_, 4); \
        else LNB_GO(T_, 8); } while (0)
    static const bool v512 = !(getenv("PA_LNB_512") && atoi(getenv("PA_LNB_512")) == 0);
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (d == 512 && v512 && al16(dz) && al16(dy) && al16(z) && al16(gamma) && al16(partial) && (!thr || al16(ddrop))) {
        if (img && (dtype != PA_F32 || (reinterpret_cast<uintptr_t>(img) & 7))) return PA_EINVAL;
        if (dtype == PA_BF16)
            PA_LAUNCH(layernorm_bwd512_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (bf16*)dz, (bf16*)ddrop, (const bf16*)dy,
                      (const bf16*)z, gamma, mean, rstd, partial, rows, thr, scale, drop_seed, want_dzsum ? 1 : 0, nullptr, 0);
        else
            PA_LAUNCH(layernorm_bwd512_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (float*)dz, (float*)ddrop, (const float*)dy,
                      (const float*)z, gamma, mean, rstd, partial, rows, thr, scale, drop_seed, want_dzsum ? 1 : 0, img, img_pat);
        return 0;
    }
    if (img) return PA_EINVAL;                             // (callers ask pa_layernorm_bwd_can_img first)
    if (dtype == PA_BF16) LNB_NV(bf16); else LNB_NV(float);
#undef LNB_NV
#undef LNB_GO
    return 0;
}
extern "C" int32_t pa_layernorm_bwd_nparts(int64_t rows) { return (int32_t)((rows + LNB_ROWS - 1) / LNB_ROWS); }
extern "C" int pa_layernorm_finish_many(const pa_ln_finish_desc* descs, int32_t n, int32_t d, void* stream) {
    if (!descs || n <= 0 || n > PA_MAX_LN_FINISH || d <= 0) return PA_EINVAL;
    LnFinishTab t; t.n = n;
    bool any3 = false;
    for (int i = 0; i < n; ++i) {
        if (!descs[i].partial || !descs[i].dgamma || !descs[i].dbeta || descs[i].nparts <= 0) return PA_EINVAL;
        t.d[i] = descs[i];
        any3 = any3 || descs[i].dzsum != nullptr;
    }
    PA_LAUNCH(partial_finish_many_kernel, dim3((d + 63) / 64, any3 ? 3 : 2, n), dim3(256), 0, ST(stream), t, d);
    return 0;
}
extern "C" int pa_layernorm_bwd(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                                const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dzsum,
                                float* partial, int64_t rows, int32_t d, int32_t dtype,
                                float drop_p, uint32_t drop_seed, void* stream) {
    if (!dgamma || !dbeta) return PA_EINVAL;
    int rc = pa_layernorm_bwd_partial(dz, ddrop, dy, z, gamma, mean, rstd, dzsum ? 1 : 0, partial, rows, d, dtype, drop_p, drop_seed, stream);
    if (rc) return rc;
    pa_ln_finish_desc fd; fd.partial = partial; fd.nparts = pa_layernorm_bwd_nparts(rows); fd.pad_ = 0;
    fd.dgamma = dgamma; fd.dbeta = dbeta; fd.dzsum = dzsum;
    return pa_layernorm_finish_many(&fd, 1, d, stream);
}

// ---- ACTIVATION: gelu (element-wise; include/plank_hip.h pa_gelu_fwd / pa_gelu_bwd) -----------------------------------------
namespace {
__device__ __forceinline__ float gelu_f(float y) { return 0.5f * y * (1.0f + erff(y * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {       // Phi(x) + x phi(x)
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(T* out, const T* dh, const T* pre, int64_t rows, int cols, int ld, uint32_t thr,
                                                   float scale, uint32_t seed) {
    const int vec = cols >> 2;
    const int64_t total = rows * vec;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / vec;
        const int c = (int)(e - r * vec) << 2;
        const f32x4 x = ld4<T>(pre + r * ld + c);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (BWD) g = ld4<T>(dh + r * ld + c);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float v = BWD ? g[k] * gelu_grad_f(x[k]) : gelu_f(x[k]);
            o[k] = (thr == 0u || drop_keep_rc(seed, (uint32_t)r, (uint32_t)(c + k), thr)) ? v * scale : 0.f;
        }
        st4<T>(out + r * ld + c, o);
    }
}
template <bool BWD>
int gelu_launch(void* out, const void* dh, const void* pre, int64_t rows, int32_t cols, int32_t ld, int32_t dtype, float drop_p,
                uint32_t seed, void* stream) {
    if (!out || !pre || (BWD && !dh) || rows <= 0 || cols <= 0 || (cols & 3) || ld < cols || (ld & 3)) return PA_EINVAL;
    if (dtype != PA_BF16 && dtype != PA_F32) return PA_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return PA_EINVAL;
    const uint32_t thr = (uint32_t)((double)drop_p * 4294967296.0);         // as pa_gemm: keep <=> 32-bit product >= thr
    const float scale = (float)(1.0 / (1.0 - (double)thr / 4294967296.0));
    const int64_t total = rows * (cols >> 2);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == PA_BF16)
        PA_LAUNCH((gelu_kernel<bf16, BWD>), dim3(grid), dim3(256), 0, ST(stream), (bf16*)out, (const bf16*)dh, (const bf16*)pre, rows, cols, ld, thr, scale, seed);
    else
        PA_LAUNCH((gelu_kernel<float, BWD>), dim3(grid), dim3(256), 0, ST(stream), (float*)out, (const float*)dh, (const float*)pre, rows, cols, ld, thr, scale, seed);
    return 0;
}
}  // namespace
extern "C" int pa_gelu_fwd(void* out, const void* pre, int64_t rows, int32_t cols, int32_t ld, int32_t dtype, float drop_p, uint32_t drop_seed,
                           void* stream) {
    return gelu_launch<false>(out, nullptr, pre, rows, cols, ld, dtype, drop_p, drop_seed, stream);
}
extern "C" int pa_gelu_bwd(void* dpre, const void* dh, const void* pre, int64_t rows, int32_t cols, int32_t ld, int32_t dtype, float drop_p,
                           uint32_t drop_seed, void* stream) {
    return gelu_launch<true>(dpre, dh, pre, rows, cols, ld, dtype, drop_p, drop_seed, stream);
}

extern "C" int pa_switch_fwd(float* s, const void* h, int32_t dtype, const float* w, const float* b, int64_t rows,
                             int32_t d, void* stream) {
    if (!s || !h || !w || !b || rows <= 0 || (d & 3)) return PA_EINVAL;
    const int grid = (int)((rows + 3) / 4);
    if (dtype == PA_BF16) PA_LAUNCH(switch_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), s, (const bf16*)h, w, b, rows, d);
    else PA_LAUNCH(switch_fwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), s, (const float*)h, w, b, rows, d);
    return 0;
}

extern "C" int pa_switch_bwd(void* dh, int32_t accumulate, float* dw, float* db, const float* ds, const void* h,
                             int32_t dtype, const float* w, float* partial, int64_t rows, int32_t d, void* stream) {
    if (!dh || !dw || !db || !ds || !h || !w || !partial || rows <= 0 || (d & 3) || d > 256 * MAXV) return PA_EINVAL;
    const int grid = (int)((rows + LNB_ROWS - 1) / LNB_ROWS);
    const size_t shm = (size_t)(4 * d + 4) * sizeof(float);
    if (dtype == PA_BF16) PA_LAUNCH(switch_bwd_kernel<bf16>, dim3(grid), dim3(256), shm, ST(stream), (bf16*)dh, accumulate, ds, (const bf16*)h, w, partial, rows, d);
    else PA_LAUNCH(switch_bwd_kernel<float>, dim3(grid), dim3(256), shm, ST(stream), (float*)dh, accumulate, ds, (const float*)h, w, partial, rows, d);
    // partial rows: [blk][0][d] = dw, [blk][1][0] = db
    PA_LAUNCH(partial_finish_kernel, dim3((d + 63) / 64, 1), dim3(256), 0, ST(stream), partial, grid, 2 * d, d, d, dw, (float*)nullptr, (float*)nullptr);
    PA_LAUNCH(partial_finish_kernel, dim3(1, 1), dim3(256), 0, ST(stream), partial + d, grid, 2 * d, d, 1, db, (float*)nullptr, (float*)nullptr);
    return 0;
}

extern "C" int pa_mixture_nll_fwd(float* stats, float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                                  const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                                  void* stream) {
    if (!stats || !row_lse || !vocab || !ptr || !sw || !label || B <= 0 || T <= 0 || V <= 0 || ldv < V) return PA_EINVAL;
    const int grid = (int)(((int64_t)B * T + NLL_ROWS - 1) / NLL_ROWS);
    PA_LAUNCH(mixture_nll_fwd_kernel, dim3(grid), dim3(256), 0, ST(stream), stats, row_lse, vocab, ldv, ptr, sw, label, B, T, V, pad);
    return 0;
}
extern "C" int pa_mixture_nll_fwd_fin(float* stats8, float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                                      const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                                      void* stream) {
    if (!stats8 || !row_lse || !vocab || !ptr || !sw || !label || B <= 0 || T <= 0 || V <= 0 || ldv < V) return PA_EINVAL;
    const int grid = (int)(((int64_t)B * T + NLL_ROWS - 1) / NLL_ROWS);
    if (hipMemsetAsync(stats8, 0, 8 * sizeof(float), ST(stream)) != hipSuccess) return PA_EINVAL;      // sums, ticket (and the rest)
    PA_LAUNCH(mixture_nll_fwd_kernel, dim3(grid), dim3(256), 0, ST(stream), stats8, row_lse, vocab, ldv, ptr, sw, label, B, T, V, pad);
    PA_LAUNCH(mixture_nll_finish_kernel, dim3(1), dim3(64), 0, ST(stream), stats8);
    return 0;
}

extern "C" int pa_mixture_nll_bwd_up(void* dvocab, void* dptr, int32_t out_dtype, float* dsw, const float* stats,
                                  const float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                                  const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                                  float gscale, const float* upstream, void* stream) {
    if (!dvocab || !dptr || !dsw || !stats || !row_lse || !vocab || !ptr || !sw || !label) return PA_EINVAL;
    if (B <= 0 || T <= 0 || V <= 0 || ldv < V) return PA_EINVAL;
    const int grid = (int)(((int64_t)B * T + 3) / 4);
    if (out_dtype == PA_BF16) PA_LAUNCH(mixture_nll_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, ST(stream), (bf16*)dvocab, (bf16*)dptr, dsw, stats, row_lse, vocab, ldv, ptr, sw, label, B, T, V, pad, gscale, upstream);
    else PA_LAUNCH(mixture_nll_bwd_kernel<float>, dim3(grid), dim3(256), 0, ST(stream), (float*)dvocab, (float*)dptr, dsw, stats, row_lse, vocab, ldv, ptr, sw, label, B, T, V, pad, gscale, upstream);
    return 0;
}
extern "C" int pa_mixture_nll_bwd(void* dvocab, void* dptr, int32_t out_dtype, float* dsw, const float* stats,
                                  const float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                                  const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                                  float gscale, void* stream) {
    return pa_mixture_nll_bwd_up(dvocab, dptr, out_dtype, dsw, stats, row_lse, vocab, ldv, ptr, sw, label, B, T, V, pad, gscale, nullptr, stream);
}

extern "C" int pa_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float b1,
                            float b2, float eps, int32_t step, float gscale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
         reinterpret_cast<uintptr_t>(v)) & 15) return PA_EALIGN;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    PA_LAUNCH(adam_kernel, dim3(grid_for(n >> 2, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, (bf16*)p_bf16, n,
                       step_size, b1, b2, eps, inv_sqrt_bc2, gscale);
    return 0;
}

extern "C" int pa_cast(void* dst, int32_t dst_dtype, const void* src, int32_t src_dtype, int64_t n, void* stream) {
    if (!dst || !src || n <= 0) return PA_EINVAL;
    const int grid = grid_for(n, 256, 2048);
    if (dst_dtype == PA_BF16 && src_dtype == PA_F32) PA_LAUNCH((cast_kernel<bf16, float>), dim3(grid), dim3(256), 0, ST(stream), (bf16*)dst, (const float*)src, n);
    else if (dst_dtype == PA_F32 && src_dtype == PA_BF16) PA_LAUNCH((cast_kernel<float, bf16>), dim3(grid), dim3(256), 0, ST(stream), (float*)dst, (const bf16*)src, n);
    else if (dst_dtype == PA_F32 && src_dtype == PA_F32) PA_LAUNCH((cast_kernel<float, float>), dim3(grid), dim3(256), 0, ST(stream), (float*)dst, (const float*)src, n);
    else return PA_EINVAL;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-in for a ring all-reduce (one-GPU rehearsal of the data-parallel step, plankassembly_amd/distributed.py): `blocks`
// workgroups hold their CUs for `min_ticks` ticks of the 100 MHz wall clock and meanwhile stream the slice through HBM (read
// and write back in place, values unchanged) at least `passes` times - what a rank's RCCL kernels do to the chip while the
// gradient slices of reference configs/train_complete.yaml:18 (`strategy: ddp`) are exchanged: occupy a few CUs for about
// 2 (n-1)/n * bytes / bus bandwidth and move 2 (n-1)/n * bytes each way through the local memory.
__global__ __launch_bounds__(256) void fake_collective_kernel(f32x4* buf, long long n4, int passes, long long min_ticks) {
    const long long t0 = wall_clock64();
    const long long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long long lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    int pass = 0;
    do {
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            f32x4 v = __builtin_nontemporal_load(buf + i);
            __builtin_nontemporal_store(v, buf + i);
        }
        ++pass;
    } while (pass < passes || wall_clock64() - t0 < min_ticks);
}
extern "C" int pa_fake_collective(void* buf, int64_t bytes, int32_t blocks, int32_t passes, float min_us, void* stream) {
    if (!buf || bytes < 16 || blocks <= 0 || blocks > 256 || passes < 1 || (reinterpret_cast<uintptr_t>(buf) & 15)) return PA_EINVAL;
    PA_LAUNCH(fake_collective_kernel, dim3(blocks), dim3(256), 0, ST(stream), reinterpret_cast<f32x4*>(buf), (long long)(bytes / 16),
              passes, (long long)(min_us * 100.0f));
    return 0;
}
