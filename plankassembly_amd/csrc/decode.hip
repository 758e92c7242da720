// Greedy autoregressive decode for gfx950: reference plankassembly/models.py:267-307 (eval_step) with a
// K/V cache instead of the reference's O(T^2) prefix recompute (mathematically identical for a causal
// decoder in eval mode).  One decode step = a fixed sequence of kernels that
// read the step index from DEVICE memory, so the host can capture the step once in a hipGraph and
// replay it max_output_length times.
//   * single-query attention (HBM-bound batched GEMV): 16-byte K/V row chunks per lane, score
//     reduction across the lanes of a row with wave shuffles, online softmax per lane group,
//     partial (m, l, acc) merge through LDS.
//   * sampling kernel fuses: pointer logits against the hidden-state cache, switch head, vocab /
//     pointer softmax, gating, the 1e-6 pointer-mask fill, first-max argmax, pointer copy, END tracking
//     (reference models.py:168-186 eval branch of _create_dist and 235-256 _sample).
#include <math.h>
#include <new>
#include <string.h>
#include "pa_device.h"
#include "model.h"

struct DecodeLayout {
    int B = 0, S = 0, Tmax = 0;
    std::vector<void*> cross_k, cross_v, self_k, self_v;   // per decoder layer, per-head contiguous: [B][H][S|Tmax][dh]
    void* kv_tmp;                              // [B*S][2d] projection output before the per-head re-layout
    void *hid_cache, *x, *qkv, *ao, *z, *y, *q, *ff, *pfeat, *h;
    float *vlog, *mean, *rstd;
    int64_t *tokens, *attach; int32_t *first_end, *t_dev;
    uint8_t* kpm;                              // copy of input_mask: the captured step must not depend on batch tensors
    int32_t* cu;                               // copy of the packed-row offsets (NULL: dense memory + kpm)
    int32_t* cu_store;
    std::vector<hipEvent_t> pair_ev;           // pa_decode_step_pair: [2][2 * n_dec] attention-done events (lane A's layout owns them)
    // LayerNorm folded into the Linear that consumes it (bf16; pa_gemm_norm_a): per decoder layer, for the three Linears whose
    // input is a sublayer's LayerNorm output - [0] self-attention in_proj (norm3 of the layer before), [1] the cross-attention
    // query projection (norm1), [2] linear1 (norm2): gamma-scaled weight, u, v (include/plank_hip.h)
    bool fold = false;
    void* z2 = nullptr;                        // second pre-norm buffer (z of the cross-attention block)
    std::vector<void*> fw[3]; std::vector<float*> fu[3], fv[3];
    // f32 residual stream of the bf16 step (PLANK_DECODE_F32_RESID, default on with `fold`): x / y / z / z2 hold f32 rows, zb / z2b /
    // xb / hb their bf16 copies = the matrix operands of the Linears that read them, hf = decoder.norm's output in f32 (vocabulary head).
    // Everything else of the step - weights, Q / K / V, the caches, attention outputs, FFN hidden rows - stays bf16.
    bool f32res = false;
    void *zb = nullptr, *z2b = nullptr, *xb = nullptr;
    float* hf = nullptr;
    // Absorbed ("multi-query") cross-attention (csrc/decode_mq.h; bf16, d_model 512, <= 8 heads, with `fold`): the step attends over a
    // copy of the encoder output rows (`mem`) instead of per-layer K / V caches.  qt / ctx: [B][H][d] query / context rows of a step;
    // wo_t / bo_t per layer: W_o,h W_v,h as one [d][H d] bf16 matrix and b_o + W_o b_v.
    bool mq = false;
    bool mq_self = false;                      // exact f32: the SELF-attention absorbed too - self_k[i] caches the layer-input rows [B][Tmax][d]
    std::vector<void*> wvt_self;               // per layer: head-transposed W_v of the self-attention (mq_contract_v_kernel)
    bool mq_contract = false;                  // exact f32, dh 64: W_v as its own launch + the ordinary out-projection instead of wo_t (decode_mq.h)
    void *mem = nullptr, *qt = nullptr, *ctx = nullptr;
    std::vector<void*> wo_t; std::vector<float*> bo_t;
    // bf16 step (f32 residual stream), round 6: the SELF-attention absorbed as well - self_k[i] caches the layer-input rows (bf16), the
    // out-projection runs on the context rows with W~o,self = W_o,h W_v,h (wo_ts / bo_ts); self_v is not read
    bool mq_self_bf = false;
    std::vector<void*> wo_ts; std::vector<float*> bo_ts;
    void* mq_sp = nullptr; int64_t mq_sp_bytes = 0;   // range blocks of the absorbed cross-attention (decode_mq.h): tickets (zero between launches) + partials
};

namespace {

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)
constexpr float LOG2E_F = 1.4426950408889634f;
#include "split_merge.h"
}  // namespace
#include "decode_mq.h"
namespace {

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dec_embed_kernel(T* x, const float* value, const float* coord, const float* pos,
                                                        const int64_t* tokens, int Tmax, const int32_t* t_dev, int B, int d, int dof,
                                                        bf16* x_lp = nullptr) {
    const int t = *t_dev;
    const int vec = d >> 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < B * vec; e += gridDim.x * 256) {
        const int b = e / vec, c = (e % vec) << 2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            const int64_t v = tokens[(int64_t)b * Tmax + (t - 1)];
            acc = *reinterpret_cast<const f32x4*>(value + v * d + c);
            acc += *reinterpret_cast<const f32x4*>(coord + (int64_t)((t - 1) % dof) * d + c);
            acc += *reinterpret_cast<const f32x4*>(pos + (int64_t)((t - 1) / dof) * d + c);
        }
        st4<T>(x + (int64_t)b * d + c, acc);
        if (x_lp) st4<bf16>(x_lp + (int64_t)b * d + c, acc);
    }
}

// bf16 copy of f32 rows (the f32-residual step: decoder.norm's output as the pointer head's operand and for the hidden cache)
__global__ __launch_bounds__(256) void dec_cast_kernel(bf16* out, const float* in, int64_t n4) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (int64_t)gridDim.x * 256)
        st4<bf16>(out + 4 * e, *reinterpret_cast<const f32x4*>(in + 4 * e));
}

// K/V caches are kept per head ([B][H][L][dh], each (b, h) a contiguous stream): reading 128-byte head slices out of
// [B*L][2d] rows is a 2 KiB-strided access that camps on two L2 channels per XCD.  The self-attention caches are appended
// to by dec_attn_kernel itself (`newkv`).
// [B*S][2d] (k | v) -> K [B][H][S][dh], V [B][H][S][dh]
template <typename T>
__global__ __launch_bounds__(256) void dec_split_heads_kernel(T* kc, T* vc, const T* kv, int64_t rows, int S, int d, int H,
                                                              const int32_t* cu, const int32_t* rowmap) {
    const int dh = d / H, vec = d >> 2;
    const int64_t total = rows * vec;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / vec;
        const int c = (int)(e % vec) << 2;
        int b = (int)(row / S), s = (int)(row % S);
        if (cu) { b = rowmap[row] / S; s = (int)row - cu[b]; }          // packed memory: s-th valid position of element b
        const int h = c / dh, cc = c % dh;
        const int64_t dst = (((int64_t)b * H + h) * S + s) * dh + cc;
        st4<T>(kc + dst, ld4<T>(kv + row * 2 * d + c));
        st4<T>(vc + dst, ld4<T>(kv + row * 2 * d + d + c));
    }
}

// single-query attention; grid (H, B), 4 waves.  Lk = fixed_lk or *t_dev + 1.
// `newkv` (self-attention): the step's in_proj output [B][3d]; key t = Lk - 1 is not in the cache yet - its K / V head
// slices are taken from there (per-lane source select, no read-after-write through memory) and appended to the caches by
// this block for the later steps.  That is the whole of what a separate append launch did (6 launches per step).
template <typename T, int DH>
__global__ __launch_bounds__(256) void dec_attn_kernel(T* out, const T* q, int ldq, T* kc, T* vc, int Lmax,
                                                       const uint8_t* kpm, int fixed_lk, const int32_t* t_dev,
                                                       int d, float scale, const int32_t* cu, const T* newkv) {
    constexpr int EB = ET<T>::EB;
    constexpr int LPR = DH / EB;             // lanes per key row
    constexpr int KPW = 64 / LPR;            // keys per wave step
    constexpr int NG = 4 * KPW;              // partial groups per block
    __shared__ float part[NG * (DH + 2)];
    const int h = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane % LPR, slot = lane / LPR;
    const int Lk = cu ? (cu[b + 1] - cu[b]) : (fixed_lk > 0 ? fixed_lk : (*t_dev + 1));
    constexpr int ldkv = DH;                                     // per-head contiguous cache
    const int64_t bh = ((int64_t)b * gridDim.x + h) * Lmax * DH + c * EB;
    const T* kb = kc + bh;
    const T* vb = vc + bh;
    const T* nk = newkv ? newkv + (int64_t)b * 3 * d + d + h * DH + c * EB : nullptr;        // this lane's chunk of the new K row
    const T* nv = newkv ? nk + d : nullptr;
    if (newkv && tid < LPR) {                                                                // append for the steps to come
        const int64_t dst = ((int64_t)b * gridDim.x + h) * Lmax * DH + (int64_t)(Lk - 1) * DH + c * EB;
        *reinterpret_cast<u32x4*>(kc + dst) = *reinterpret_cast<const u32x4*>(nk);
        *reinterpret_cast<u32x4*>(vc + dst) = *reinterpret_cast<const u32x4*>(nv);
    }
    const uint8_t* mk = kpm ? kpm + (int64_t)b * Lmax : nullptr;
    float qv[EB];
    {
        const T* qp = q + (int64_t)b * ldq + h * DH + c * EB;
#pragma unroll
        for (int e = 0; e < EB; e += 4) { const f32x4 t4 = ld4<T>(qp + e); qv[e] = t4[0]; qv[e + 1] = t4[1]; qv[e + 2] = t4[2]; qv[e + 3] = t4[3]; }
    }
    const float sl = scale * LOG2E_F;
    float m = -INFINITY, l = 0.f, acc[EB];
#pragma unroll
    for (int e = 0; e < EB; ++e) acc[e] = 0.f;
    // UN key groups per iteration; K and V rows travel as raw 16-byte vectors (one load each per lane and key) and are
    // widened only when used, so 2*UN loads are in flight per lane at 2*UN*4 VGPRs
    constexpr int UN = 8;
    auto widen = [](const u32x4& raw, float* f) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int w = 0; w < 4; ++w) { f[2 * w] = bf16_lo(raw[w]); f[2 * w + 1] = bf16_hi(raw[w]); }
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) f[w] = __uint_as_float(raw[w]);
        }
    };
    for (int base0 = wave * KPW; base0 < Lk; base0 += 4 * KPW * UN) {
        u32x4 kraw[UN], vraw[UN];
        bool okk[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int key = base0 + j * 4 * KPW + slot;
            const bool in = key < Lk;
            okk[j] = in && !(mk && mk[in ? key : 0]);
            // unconditional loads (clamped row; out-of-range keys are discarded through okk): a branch around the loads
            // would put a wait between them
            const int64_t kc_ = (int64_t)min(key, Lk - 1) * ldkv;
            const bool fresh = newkv && key >= Lk - 1;                 // the row this step produced (clamped keys land here too)
            // (streamed once per step, never re-read by this CU: non-temporal, so the lines do not displace the weights in L2)
            kraw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(fresh ? nk : kb + kc_));
            vraw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(fresh ? nv : vb + kc_));
        }
        float s[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            float kf[EB];
            widen(kraw[j], kf);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < EB; ++e) a += qv[e] * kf[e];
#pragma unroll
            for (int o = 1; o < LPR; o <<= 1) a += __shfl_xor(a, o);
            s[j] = okk[j] ? a * sl : -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int j = 0; j < UN; ++j) mn = fmaxf(mn, s[j]);
        if (mn > -INFINITY) {
            const float alpha = exp2f(m - mn);
            l *= alpha;
#pragma unroll
            for (int e = 0; e < EB; ++e) acc[e] *= alpha;
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const float pj = exp2f(s[j] - mn);           // exp2(-inf) = 0 for masked / out-of-range keys
                l += pj;
                float vf[EB];
                widen(vraw[j], vf);
#pragma unroll
                for (int e = 0; e < EB; ++e) acc[e] += pj * vf[e];
            }
            m = mn;
        }
    }
    const int g = wave * KPW + slot;
    if (c == 0) { part[g * (DH + 2)] = m; part[g * (DH + 2) + 1] = l; }
#pragma unroll
    for (int e = 0; e < EB; ++e) part[g * (DH + 2) + 2 + c * EB + e] = acc[e];
    __syncthreads();
    if (tid < DH) {
        float M = -INFINITY;
        for (int i = 0; i < NG; ++i) M = fmaxf(M, part[i * (DH + 2)]);
        float num = 0.f, den = 0.f;
        for (int i = 0; i < NG; ++i) {
            const float mi = part[i * (DH + 2)];
            if (mi == -INFINITY) continue;
            const float w = exp2f(mi - M);
            den += part[i * (DH + 2) + 1] * w;
            num += part[i * (DH + 2) + 2 + tid] * w;
        }
        st1<T>(out + (int64_t)b * d + h * DH + tid, den > 0.f ? num / den : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------
struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {   // first-max: larger value, then smaller index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return v;
}
__device__ __forceinline__ bool ptr_allowed(int i, int j) {      // reference models.py:91-101, closed form
    if (i < 6) return false;
    return j < 6 ? (j == i % 6) : ((j % 6) == ((i % 6) + 3) % 6);
}

constexpr int MAX_T = 2048;

template <typename T>
__global__ __launch_bounds__(256) void dec_sample_kernel(const float* vlog, int ldv, const T* pfeat, const T* h, T* hid_cache,
                                                         const float* sw_w, const float* sw_b, int64_t* tokens,
                                                         int64_t* attach, int32_t* first_end, int32_t* t_dev,
                                                         int Tmax, int d, int V, int end_tok,
                                                         // fused tail (PLANK_DECODE_FUSE_TAIL, default on): the NEXT step's input
                                                         // embedding of this row and the step counter - what dec_embed_kernel and
                                                         // dec_advance_kernel did as two more launches of the serial chain
                                                         int fuse, float* x32, T* xT, bf16* x_lp, const float* value,
                                                         const float* coord, const float* pos, int dof) {
    __shared__ float plog[MAX_T];
    __shared__ long long s_tok;
    __shared__ float sh[4];
    __shared__ ArgMax sha[4];
    __shared__ float s_sw;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = *t_dev, sz = t + 1, i = t;
    const T* hb = h + (int64_t)b * d;
    // hidden-state cache laid out [Tmax][B][d]: the 256 blocks walk the rows j in step, and with a per-sequence
    // [B][Tmax][d] layout (1 MB apart) they would all hit the same HBM channel at the same time
    const int64_t nb = gridDim.x;
    T* cache = hid_cache + (int64_t)b * d;                          // row j of this sequence: cache + j * nb * d
    for (int c = tid; c < d; c += 256) cache[(int64_t)t * nb * d + c] = hb[c];
    const float* vr = vlog + (int64_t)b * ldv;
    // vocab softmax statistics
    float vmax = -INFINITY;
    for (int k = tid; k < V; k += 256) vmax = fmaxf(vmax, vr[k]);
    vmax = block_max(vmax, sh);
    float vsum = 0.f;
    for (int k = tid; k < V; k += 256) vsum += expf(vr[k] - vmax);
    vsum = block_sum(vsum, sh);
    ArgMax best{-INFINITY, 0x7fffffff};
    if (sz < 6) {                                                     // models.py:172-173: un-gated vocab softmax
        for (int k = tid; k < V; k += 256) best = better(best, ArgMax{expf(vr[k] - vmax) / vsum, k});
    } else {
        // pointer logits over the hidden prefix (row j = t is written above but masked: j >= i)
        // each wave takes 8 cached rows at a time, all their loads issued (unconditionally: rows clamped, tail discarded)
        // before the first reduction; the pointer-feature row stays in registers.  Per row the arithmetic order is the
        // one-row-at-a-time order (chunk by chunk per lane, then the wave sum): logits and greedy tokens are unchanged.
        const T* pf = pfeat + (int64_t)b * d;
        auto ptr_logits = [&](auto NCH_) {
            constexpr int NCH = decltype(NCH_)::value, PUN = 8;           // NCH = d / 256 chunks of 4 columns per lane
            f32x4 pa[NCH];
#pragma unroll
            for (int q = 0; q < NCH; ++q) pa[q] = ld4<T>(pf + (lane << 2) + q * 256);
            for (int j0 = wave * PUN; j0 < t; j0 += 4 * PUN) {
                f32x4 hh[PUN][NCH];
#pragma unroll
                for (int u = 0; u < PUN; ++u) {
                    const T* row = cache + (int64_t)min(j0 + u, t - 1) * nb * d + (lane << 2);
#pragma unroll
                    for (int q = 0; q < NCH; ++q) hh[u][q] = ld4<T>(row + q * 256);
                }
#pragma unroll
                for (int u = 0; u < PUN; ++u) {
                    float sacc = 0.f;
#pragma unroll
                    for (int q = 0; q < NCH; ++q)
                        sacc += pa[q][0] * hh[u][q][0] + pa[q][1] * hh[u][q][1] + pa[q][2] * hh[u][q][2] + pa[q][3] * hh[u][q][3];
                    sacc = wave_sum(sacc);
                    if (lane == 0 && j0 + u < t) plog[j0 + u] = sacc / (float)d;
                }
            }
        };
        if (d == 512) ptr_logits(std::integral_constant<int, 2>{});
        else if (d == 256) ptr_logits(std::integral_constant<int, 1>{});
        else if (d == 768) ptr_logits(std::integral_constant<int, 3>{});
        else if (d == 1024) ptr_logits(std::integral_constant<int, 4>{});
        else {
            for (int j = wave; j < t; j += 4) {
                float s = 0.f;
                for (int c = lane << 2; c < d; c += 256) {
                    const f32x4 a = ld4<T>(pf + c), hh = ld4<T>(cache + (int64_t)j * nb * d + c);
                    s += a[0] * hh[0] + a[1] * hh[1] + a[2] * hh[2] + a[3] * hh[3];
                }
                s = wave_sum(s);
                if (lane == 0) plog[j] = s / (float)d;
            }
        }
        if (wave == 0) {
            float s = 0.f;
            for (int c = lane << 2; c < d; c += 256) {
                const f32x4 hh = ld4<T>(hb + c); const f32x4 w = *reinterpret_cast<const f32x4*>(sw_w + c);
                s += hh[0] * w[0] + hh[1] * w[1] + hh[2] * w[2] + hh[3] * w[3];
            }
            s = wave_sum(s);
            if (lane == 0) s_sw = s + sw_b[0];
        }
        __syncthreads();
        const float prob = 1.0f / (1.0f + expf(-s_sw));
        float pmax = -INFINITY;
        for (int j = tid; j < t; j += 256) pmax = fmaxf(pmax, plog[j]);
        pmax = block_max(pmax, sh);
        float psum = 0.f;
        for (int j = tid; j < t; j += 256) psum += expf(plog[j] - pmax);
        psum = block_sum(psum, sh);
        const float gate_v = 1.0f - prob;
        for (int k = tid; k < V; k += 256) best = better(best, ArgMax{(expf(vr[k] - vmax) / vsum) * gate_v, k});
        for (int j = tid; j < sz; j += 256) {
            float val = 1e-6f;                                        // models.py:183-184 fill after gating
            if (ptr_allowed(i, j)) val = (j < i) ? (expf(plog[j] - pmax) / psum) * prob : 0.f;
            best = better(best, ArgMax{val, V + j});
        }
    }
    // block arg-max (first maximum)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax other{__shfl_xor(best.v, o), __shfl_xor(best.i, o)};
        best = better(best, other);
    }
    if (lane == 0) sha[wave] = best;
    __syncthreads();
    if (tid == 0) {
        best = better(better(sha[0], sha[1]), better(sha[2], sha[3]));
        int64_t tok = best.i, ptr = -1;
        if (best.i >= V) { ptr = best.i - V; tok = tokens[(int64_t)b * Tmax + ptr]; }   // models.py:248-251
        tokens[(int64_t)b * Tmax + t] = tok;
        attach[(int64_t)b * Tmax + t] = ptr;
        if (tok == end_tok && first_end[b] < 0) first_end[b] = t;
        s_tok = tok;
    }
    if (!fuse) return;
    __syncthreads();
    {   // x(t + 1) = value[token(t)] + coord[t % dof] + pos[t / dof]   (reference models.py:114-123 on the token just sampled)
        const long long tk = s_tok;
        for (int c = tid << 2; c < d; c += 1024) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(value + tk * d + c);
            acc += *reinterpret_cast<const f32x4*>(coord + (int64_t)(t % dof) * d + c);
            acc += *reinterpret_cast<const f32x4*>(pos + (int64_t)(t / dof) * d + c);
            if (x32) { *reinterpret_cast<f32x4*>(x32 + (int64_t)b * d + c) = acc; if (x_lp) st4<bf16>(x_lp + (int64_t)b * d + c, acc); }
            else st4<T>(xT + (int64_t)b * d + c, acc);
        }
    }
    // the block that finishes last advances the step counter: every block has read *t_dev long before its ticket
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned k = atomicAdd(reinterpret_cast<unsigned*>(t_dev + 1), 1u);
        if (k == gridDim.x - 1) { t_dev[1] = 0; t_dev[0] = t + 1; }
    }
}

// last decoder layer's norm3 and decoder.norm back to back on the same rows, plus the bf16 copy of the result (f32-residual step):
// one wave per row, two-pass statistics in registers - three launches (two LayerNorms, dec_cast_kernel) of the serial chain in one
__global__ __launch_bounds__(256) void dec_tail_norm_kernel(float* hf, bf16* h, const float* z, const float* g3, const float* b3, float eps3,
                                                            const float* gf, const float* bf, float epsf, int rows, int d) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int NV = 2;                                    // d <= 512: two 4-wide vectors per lane
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * 64) << 2;
        v[i] = c < d ? *reinterpret_cast<const f32x4*>(z + (int64_t)row * d + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    auto norm = [&](const float* g, const float* bb, float eps) {
        const float mu = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * 64) << 2;
            if (c < d) for (int j = 0; j < 4; ++j) { const float t = v[i][j] - mu; q += t * t; }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) / d + eps);          // (the arithmetic of layernorm_fwd_kernel, term for term)
        s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (lane + i * 64) << 2;
            if (c < d) {
                const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c), be = *reinterpret_cast<const f32x4*>(bb + c);
                for (int j = 0; j < 4; ++j) v[i][j] = (v[i][j] - mu) * rs * gg[j] + be[j];
                s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
            }
        }
    };
    norm(g3, b3, eps3);
    norm(gf, bf, epsf);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (lane + i * 64) << 2;
        if (c < d) { *reinterpret_cast<f32x4*>(hf + (int64_t)row * d + c) = v[i]; st4<bf16>(h + (int64_t)row * d + c, v[i]); }
    }
}

__global__ void dec_advance_kernel(int32_t* t_dev) { if (threadIdx.x == 0) *t_dev += 1; }

// ------------------------------------------------------------------------------------------------
int linear(pa_model* m, const void* A, const void* W, const float* bias, void* Cout, int ldc, int M, int N, int K,
           int relu, const void* R, int out_dtype, void* st) {
    pa_gemm_args g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = W; g.C = Cout; g.bias = bias; g.R = R;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc; g.ldr = ldc;
    g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1;
    g.in_dtype = m->cfg.dtype; g.out_dtype = out_dtype < 0 ? m->cfg.dtype : out_dtype;
    g.alpha = 1.f; g.relu = relu; g.aux_scale = 1.f; g.splitk = 1;
    return pa_gemm(&g, st);
}

// y = LayerNorm(R + A W^T + bias): pa_gemm into `z` followed by pa_layernorm_fwd.  PLANK_DECODE_FUSE_LN=1 swaps in the
// one-launch row-block kernel (pa_gemm_ln; bf16, d_model 512, K = 512) to reproduce the measurement that keeps it OFF:
// at 256 rows it is 8 blocks, each pulling the layer's whole 512 KB weight - cold, the step cycles through 38 MB of
// them - with ~100 KB in flight: 1.317 ms / step against 1.239 for the two launches (32 blocks x 64 KB each).
int linear_ln(pa_model* m, const void* A, const void* W, const float* bias, const void* R, void* z, void* y, const float* gamma,
              const float* beta, float eps, int M, int K, void* st) {
    const pa_model_cfg& c = m->cfg;
    static const bool fuse = getenv("PLANK_DECODE_FUSE_LN") && atoi(getenv("PLANK_DECODE_FUSE_LN")) != 0;
    if (fuse && c.dtype == PA_BF16 && c.d_model == 512 && K == 512 && M <= pa_gemm_ln_max_rows()) {
        pa_gemm_ln_args g; memset(&g, 0, sizeof(g));
        g.A = A; g.W = W; g.bias = bias; g.R = R; g.Z = nullptr; g.Y = y; g.gamma = gamma; g.beta = beta;
        g.M = M; g.N = c.d_model; g.K = K; g.lda = K; g.ldw = K; g.ldr = c.d_model; g.ldz = c.d_model; g.ldy = c.d_model;
        g.eps = eps;
        return pa_gemm_ln(&g, st);
    }
    RC(linear(m, A, W, bias, z, c.d_model, M, c.d_model, K, 0, R, -1, st));
    return pa_layernorm_fwd(y, z, gamma, beta, m->dec->mean, m->dec->rstd, M, c.d_model, eps, c.dtype, st);
}

// C = epi(LayerNorm(Z) W^T + b) with the LayerNorm folded into the product (pa_gemm_norm_a); Y (optional) receives LayerNorm(Z)
int linear_norm_a(pa_model* m, const void* Z, const void* Wf, const float* u, const float* v, const float* gamma, const float* beta,
                  float eps, void* Y, void* Cout, int ldc, int M, int N, int K, int relu, void* st) {
    pa_gemm_args g; memset(&g, 0, sizeof(g));
    g.A = Z; g.B = Wf; g.C = Cout; g.bias = v;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc;
    g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = m->cfg.dtype; g.out_dtype = m->cfg.dtype;
    g.alpha = 1.f; g.relu = relu; g.aux_scale = 1.f; g.splitk = 1;
    pa_gemm_norm_ext x; memset(&x, 0, sizeof(x));
    x.u = u; x.gamma = gamma; x.beta = beta; x.y = Y; x.ldy = K; x.eps = eps;
    if (m->cfg.dtype == PA_F32) { x.zf = (const float*)Z; x.ldzf = K; x.y_f32 = 1; }     // exact-f32 fold: statistics from the rows themselves
    return pa_gemm_norm_a(&g, &x, st);
}

// f32-residual forms (bf16 step): Z (f32) = A W^T + bias + R (f32), with its bf16 copy for the next matrix product
int linear_res32(pa_model* m, const void* A, const void* W, const float* bias, const float* R, float* Z, void* Zb, int M, int N, int K, void* st) {
    pa_gemm_args g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = W; g.C = Z; g.bias = bias; g.R = R; g.C_lp = Zb; g.ldc_lp = N;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.ldr = N;
    g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = PA_BF16; g.out_dtype = PA_F32;
    g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = 1;
    return pa_gemm(&g, st);
}
// C (bf16) = epi(LayerNorm(Zf) W^T + b) with the product on the bf16 copy Zb and the statistics / Y (f32) from the f32 rows
int linear_norm_a32(pa_model* m, const void* Zb, const float* Zf, const void* Wf, const float* u, const float* v, const float* gamma,
                    const float* beta, float eps, float* Y, void* Cout, int ldc, int M, int N, int K, int relu, void* st) {
    (void)m;
    pa_gemm_args g; memset(&g, 0, sizeof(g));
    g.A = Zb; g.B = Wf; g.C = Cout; g.bias = v;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc;
    g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = PA_BF16; g.out_dtype = PA_BF16;
    g.alpha = 1.f; g.relu = relu; g.aux_scale = 1.f; g.splitk = 1;
    pa_gemm_norm_ext x; memset(&x, 0, sizeof(x));
    x.u = u; x.gamma = gamma; x.beta = beta; x.y = Y; x.ldy = K; x.eps = eps; x.zf = Zf; x.ldzf = K; x.y_f32 = 1;
    return pa_gemm_norm_a(&g, &x, st);
}

// Which forms the step takes (shared by the workspace layout and pa_decode_begin).
// fold: LayerNorm folded into its consumer Linear (bf16 decode): 17 LayerNorm launches fewer per step.
// On the ring kernel (gemm3s_kernel's prologue statistics pass) this was MEASURED NULL on MI355X (B 256, 1024 steps, one lane:
// 1.228 ms / step folded against 1.210 - the statistics pass pays the memory round trips the LayerNorm launch paid).  On
// gemm_skinny_kernel the Z tile is resident in LDS and the statistics cost no round trip: 1.072 against 1.089 ms / step.  So: on by
// default exactly where pa_gemm_norm_a takes the skinny kernel (d_model 512, at most 512 rows); PLANK_DECODE_FOLD_LN=0 / 1 forces.
// The exact-f32 step folds too since round 4 (pa_ln_fold_weights_f32 + the f32 skinny kernel with the statistics taken from
// the rows themselves): 17 LayerNorm launches fewer per step - measured +0.5 % only (B 256: 1.995 vs 2.005 ms / step; the folded
// Linears re-read their rows for the statistics), token-exact against the reference in every decode test.  d_model 512, <= 512 rows.
// f32res: f32 residual stream inside the bf16 step (see DecodeLayout::f32res): on wherever every Linear of the step takes the skinny
// kernel (d_model 512, at most 512 rows).  tests/bf16_decode_sim.py / profiles/r04_bf16_decode_rounding_sim.txt: exact-prefix
// agreement with the f32 tokens 0.40 -> 0.63-0.70 on 32 rows x 128 steps.  PLANK_DECODE_F32_RESID=0 restores the all-bf16 step.
// mq: absorbed cross-attention (csrc/decode_mq.h) wherever the fold is on, with at most 8 heads; PLANK_DECODE_MQ=0 (bf16) /
// PLANK_DECODE_MQ_F32=0 (exact f32) restore the per-layer K / V caches.
struct DecodeModes { bool fold, f32res, mq; };
DecodeModes decode_modes(const pa_model_cfg& c, int B, int S) {
    const int d = c.d_model;
    static const int fold_force = getenv("PLANK_DECODE_FOLD_LN") ? atoi(getenv("PLANK_DECODE_FOLD_LN")) : -1;
    static const int f32res_env = getenv("PLANK_DECODE_F32_RESID") ? atoi(getenv("PLANK_DECODE_F32_RESID")) : 1;
    static const int mq_env = getenv("PLANK_DECODE_MQ") ? atoi(getenv("PLANK_DECODE_MQ")) : 1;
    static const int mq32_env = getenv("PLANK_DECODE_MQ_F32") ? atoi(getenv("PLANK_DECODE_MQ_F32")) : 1;
    const bool fold_env = fold_force >= 0 ? fold_force != 0 : (d == 512 && B <= 512);
    DecodeModes r;
    r.fold = fold_env && c.dtype == PA_BF16 && d % 64 == 0 && c.d_ff % 32 == 0 && c.d_ff >= d && (size_t)((B + 63) / 64) * ((3 * d + 63) / 64) <= 512;
    if (c.dtype == PA_F32) r.fold = fold_env && d == 512 && B <= 512 && c.d_ff % 32 == 0 && c.d_ff >= d;   // (N < K: pa_gemm_norm_a's f32 form cannot materialise y - ADVICE r4)
    r.f32res = r.fold && c.dtype == PA_BF16 && f32res_env != 0 && d == 512 && B <= 512 && c.d_ff % 512 == 0;
    r.mq = r.fold && (c.dtype == PA_BF16 ? mq_env != 0 : mq32_env != 0) && d == MQ_D && B <= 512 && c.n_head >= 1 && c.n_head <= MQ_MAXH && d % c.n_head == 0 && S <= (c.dtype == PA_F32 ? 16000 : MQ_MAXS);     // (f32: launch_cross_mq32 keeps 16 KB of partial-score slots beside the ring)
    return r;
}

size_t dec_layout(pa_model* m, DecodeLayout* L, char* base, int B, int S, int Tmax) {
    const pa_model_cfg& c = m->cfg;
    const size_t e = c.dtype == PA_BF16 ? 2 : 4, d = c.d_model, ff = c.d_ff;
    const DecodeModes md = decode_modes(c, B, S);
    Arena a{base, 0};
    L->B = B; L->S = S; L->Tmax = Tmax;
    L->cross_k.assign(c.n_dec, nullptr); L->cross_v.assign(c.n_dec, nullptr); L->self_k.resize(c.n_dec); L->self_v.resize(c.n_dec);
    L->wo_t.assign(c.n_dec, nullptr); L->bo_t.assign(c.n_dec, nullptr);
    for (int i = 0; i < c.n_dec; ++i) {
        if (!md.mq) { L->cross_k[i] = a.take((size_t)B * S * d * e); L->cross_v[i] = a.take((size_t)B * S * d * e); }
        L->self_k[i] = a.take((size_t)B * Tmax * d * e); L->self_v[i] = a.take((size_t)B * Tmax * d * e);
    }
    L->mem = L->qt = L->ctx = nullptr;
    if (md.mq) {
        const size_t H = c.n_head;
        L->mem = a.take((size_t)B * S * d * e);
        L->qt = a.take((size_t)B * H * d * e); L->ctx = a.take((size_t)B * H * d * e);
        L->mq_sp_bytes = mq_split_bytes(B, std::max(mq_parts(B, S), mq_parts(B, Tmax)));      // (cross-attention and the f32 step's self-attention form share it)
        L->mq_sp = L->mq_sp_bytes > 0 ? a.take((size_t)L->mq_sp_bytes) : nullptr;
        for (int i = 0; i < c.n_dec; ++i) { L->wo_t[i] = a.take(d * H * d * e); L->bo_t[i] = (float*)a.take(d * 4); }
        L->wvt_self.assign(c.n_dec, nullptr);
        if (c.dtype == PA_F32) for (int i = 0; i < c.n_dec; ++i) L->wvt_self[i] = a.take(d * d * e);
        L->wo_ts.assign(c.n_dec, nullptr); L->bo_ts.assign(c.n_dec, nullptr);
        if (c.dtype == PA_BF16) for (int i = 0; i < c.n_dec; ++i) { L->wo_ts[i] = a.take(d * H * d * e); L->bo_ts[i] = (float*)a.take(d * 4); }
    }
    L->kv_tmp = md.mq ? nullptr : a.take((size_t)B * S * 2 * d * e);
    L->hid_cache = a.take((size_t)B * Tmax * d * e);
    // (x / y / z / z2 are sized for f32 rows: the bf16 step keeps its residual stream in f32, `f32res`)
    L->x = a.take(B * d * 4); L->qkv = a.take(B * 3 * d * e); L->ao = a.take(B * d * e); L->z = a.take(B * d * 4);
    L->y = a.take(B * d * 4); L->q = a.take(B * d * e); L->ff = a.take(B * ff * e); L->pfeat = a.take(B * d * e);
    L->h = a.take(B * d * e);
    L->zb = a.take(B * d * 2); L->z2b = a.take(B * d * 2); L->xb = a.take(B * d * 2); L->hf = (float*)a.take(B * d * 4);
    L->vlog = (float*)a.take((size_t)B * ((c.vocab + 7) / 8 * 8) * 4);
    L->mean = (float*)a.take(B * 4); L->rstd = (float*)a.take(B * 4);
    L->tokens = (int64_t*)a.take((size_t)B * Tmax * 8); L->attach = (int64_t*)a.take((size_t)B * Tmax * 8);
    L->first_end = (int32_t*)a.take(B * 4); L->t_dev = (int32_t*)a.take(256);
    L->kpm = (uint8_t*)a.take((size_t)B * S);
    L->cu_store = (int32_t*)a.take((size_t)(B + 1) * 4);
    L->z2 = a.take(B * d * 4);
    for (int k = 0; k < 3; ++k) { L->fw[k].resize(c.n_dec); L->fu[k].resize(c.n_dec); L->fv[k].resize(c.n_dec); }
    for (int i = 0; i < c.n_dec; ++i) {
        const size_t rows[3] = {3 * d, d, ff};
        for (int k = 0; k < 3; ++k) {
            L->fw[k][i] = a.take(rows[k] * d * e);          // folded weight in the step's dtype (bf16, or f32 for the exact-f32 step)
            L->fu[k][i] = (float*)a.take(rows[k] * 4); L->fv[k][i] = (float*)a.take(rows[k] * 4);
        }
    }
    return a.off;
}

template <typename T>
int launch_attn(pa_model* m, T* out, const T* q, int ldq, T* kc, T* vc, int Lmax, const uint8_t* kpm,
                int fixed_lk, const int32_t* t_dev, int B, void* st, const int32_t* cu = nullptr, const T* newkv = nullptr) {
    const int d = m->cfg.d_model, H = m->cfg.n_head, dh = d / H;
    const float scale = 1.0f / sqrtf((float)dh);
    dim3 grid(H, B);
    hipStream_t s = (hipStream_t)st;
    switch (dh) {
        case 16: PA_LAUNCH((dec_attn_kernel<T, 16>), grid, dim3(256), 0, s, out, q, ldq, kc, vc, Lmax, kpm, fixed_lk, t_dev, d, scale, cu, newkv); break;
        case 32: PA_LAUNCH((dec_attn_kernel<T, 32>), grid, dim3(256), 0, s, out, q, ldq, kc, vc, Lmax, kpm, fixed_lk, t_dev, d, scale, cu, newkv); break;
        case 64: PA_LAUNCH((dec_attn_kernel<T, 64>), grid, dim3(256), 0, s, out, q, ldq, kc, vc, Lmax, kpm, fixed_lk, t_dev, d, scale, cu, newkv); break;
        default: return PA_ESHAPE;
    }
    return 0;
}

// One decode step in 2 * n_dec + 1 parts, each ending right after an attention launch (part 2i: self-attention of layer i,
// part 2i + 1: its cross-attention; the last part is the tail: final norm, heads, sampling).  The attention launches are the
// HBM-bound third of the step (they stream the K/V caches); everything between them is a chain of latency-bound launches on B
// rows.  pa_decode_step_pair() runs two half-batches ("lanes") on two streams and uses the part boundaries to keep the lanes
// out of phase: `wait_ev` is waited for immediately before the attention launch, `rec_ev` recorded immediately after it.
template <typename T>
int step_part(pa_model* m, int part, void* st, hipEvent_t wait_ev, hipEvent_t rec_ev) {
    const pa_model_cfg& c = m->cfg;
    DecodeLayout* L = m->dec;
    const int d = c.d_model, ff = c.d_ff, B = L->B, S = L->S, Tmax = L->Tmax;
    hipStream_t s = (hipStream_t)st;
    auto PF = [&](int i) { return (const float*)m->pf[i]; };
    auto PL = [&](int i) { return (const void*)m->pl[i]; };
    void* x = L->x;
    // FFN activation: ReLU rides in the first Linear's epilogue (relu flag); ACTIVATION gelu is a launch of its own behind it
    const bool gelu = c.activation == 2;
    const int act = gelu ? 0 : 1;
    auto gelu_ff = [&]() -> int { return gelu ? pa_gelu_fwd(L->ff, L->ff, B, ff, ff, c.dtype, 0.f, 0, st) : 0; };
    const int n_parts = 2 * c.n_dec + 1;
    if (part < 0 || part >= n_parts) return PA_EINVAL;
    auto fence_in = [&]() -> int { if (wait_ev) { hipError_t e = hipStreamWaitEvent(s, wait_ev, 0); if (e != hipSuccess) return (int)e; } return 0; };
    auto fence_out = [&]() -> int { if (rec_ev) { hipError_t e = hipEventRecord(rec_ev, s); if (e != hipSuccess) return (int)e; } return 0; };
    // absorbed cross-attention of layer i (csrc/decode_mq.h): q~_h = scale log2e W_k,h^T q_h for all heads, then attention over the
    // memory rows themselves; the Linear behind it applies W_o,h W_v,h to the context rows (wo_t)
    auto cross_mq = [&](pa_model* mm, int i, hipStream_t ss) -> int {
        const int H = c.n_head, pbi = mm->dec_base(i);
        const T* Wk = (const T*)mm->pl[pbi + D_CA_IN_W] + (size_t)d * d;
        const float sl = LOG2E_F / sqrtf((float)(d / H));
        const dim3 xg((B + MQ_XR - 1) / MQ_XR, H);
        if (d / H == 64) PA_LAUNCH((mq_expand_q_kernel<64, T>), xg, dim3(512), 0, ss, (T*)L->qt, (const T*)L->q, d, Wk, B, d, H, sl);
        else PA_LAUNCH((mq_expand_q_kernel<0, T>), xg, dim3(512), 0, ss, (T*)L->qt, (const T*)L->q, d, Wk, B, d, H, sl);
        RC(fence_in());
        if constexpr (sizeof(T) == 2)
            RC(launch_cross_mq((bf16*)L->ctx, (const bf16*)L->qt, (const bf16*)L->mem, L->cu ? nullptr : L->kpm, L->cu, B, S, H, d, ss,
                               L->mq_sp, L->mq_sp_bytes));
        else
            RC(launch_cross_mq32((float*)L->ctx, (const float*)L->qt, (const float*)L->mem, L->cu ? nullptr : L->kpm, L->cu, B, S, H, d, ss, nullptr,
                                 L->mq_sp, L->mq_sp_bytes));
        return fence_out();
    };
    // PLANK_DECODE_FUSE_TAIL=1 (default 0): the sampling kernel also writes the next step's input embedding and advances the step
    // counter, and (f32-residual step) norm3 + decoder.norm + the bf16 copy are one launch: 57 -> 52 launches per step, tokens
    // identical (tests/test_model_gpu.py decode tests pass either way).  MEASURED NULL on MI355X, round 5, B 256 x 1024 steps under
    // graph replay, A/B in one session: bf16 1.0905 ms / step fused against 1.0794 unfused, f32 1.9954 against 1.9813 - the five
    // launches it removes cost ~2 us each inside a replayed graph, and the sampling kernel's longer per-block tail (embedding row,
    // fence, ticket) costs the same again.  Fourth fusion of this decode step that does not pay.  The embedding
    // of step 0 is all zeros (models.py:114-123 with no token yet): pa_decode_begin clears x.
    static const int fuse_tail = getenv("PLANK_DECODE_FUSE_TAIL") ? atoi(getenv("PLANK_DECODE_FUSE_TAIL")) : 0;
    if (part == 0 && !fuse_tail) {
        const int g1 = (B * (d / 4) + 255) / 256;
        if (L->f32res)
            PA_LAUNCH(dec_embed_kernel<float>, dim3(g1), dim3(256), 0, s, (float*)L->x, PF(P_IN_VALUE), PF(P_Q_COORD), PF(P_Q_POS),
                               L->tokens, Tmax, L->t_dev, B, d, c.out_dof, (bf16*)L->xb);
        else
            PA_LAUNCH(dec_embed_kernel<T>, dim3(g1), dim3(256), 0, s, (T*)L->x, PF(P_IN_VALUE), PF(P_Q_COORD), PF(P_Q_POS),
                               L->tokens, Tmax, L->t_dev, B, d, c.out_dof, (bf16*)nullptr);
    }
    // absorbed SELF-attention of layer i (exact f32): q in L->q, the layer input rows in `xin`; the row cache is self_k[i]
    auto self_mq = [&](pa_model* mm, int i, const void* xin, hipStream_t ss) -> int {
        if constexpr (sizeof(T) == 4) {
            const int H = c.n_head, pbi = mm->dec_base(i);
            const float* Wk = (const float*)mm->pl[pbi + D_SA_IN_W] + (size_t)d * d;
            const float sl = LOG2E_F / sqrtf((float)(d / H));
            PA_LAUNCH((mq_expand_q_kernel<64, float>), dim3((B + MQ_XR - 1) / MQ_XR, H), dim3(512), 0, ss, (float*)L->qt, (const float*)L->q, d, Wk, B, d, H, sl,
                      (const float*)xin, (float*)L->self_k[i], (const int32_t*)L->t_dev, Tmax);
            RC(fence_in());
            RC(launch_cross_mq32((float*)L->ctx, (const float*)L->qt, (const float*)L->self_k[i], nullptr, nullptr, B, Tmax, H, d, ss, L->t_dev,
                                 L->mq_sp, L->mq_sp_bytes));
            RC(fence_out());
            PA_LAUNCH(mq_contract_v_kernel<float>, dim3((B + MQ_XR - 1) / MQ_XR, H), dim3(512), 0, ss, (float*)L->ao, d, (const float*)L->ctx,
                      (const float*)L->wvt_self[i], (const float*)mm->pf[pbi + D_SA_IN_B] + 2 * d, B, d, H);
            return 0;
        } else {
            (void)mm; (void)i; (void)xin; (void)ss;
            return PA_EINVAL;
        }
    };
    const bool fold = L->fold;
    if (L->f32res) {
        // ---- bf16 step with an f32 residual stream: x / y / z / z2 are f32 rows, zb / z2b / xb their bf16 copies (matrix operands)
        float* xf = (float*)L->x; float* yf = (float*)L->y; float* zf = (float*)L->z; float* z2f = (float*)L->z2;
        if (part >= 2 && (part & 1) == 0) {    // feed-forward block of the previous layer
            const int j = part / 2 - 1, pb = m->dec_base(j);
            if (L->mq && L->mq_contract) {
                PA_LAUNCH(mq_contract_v_kernel<bf16>, dim3((B + MQ_XR - 1) / MQ_XR, c.n_head), dim3(512), 0, s, (bf16*)L->ao, d, (const bf16*)L->ctx,
                          (const bf16*)L->wo_t[j], PF(pb + D_CA_IN_B) + 2 * d, B, d, c.n_head);
                RC(linear_res32(m, L->ao, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), yf, z2f, L->z2b, B, d, d, st));
            } else if (L->mq) RC(linear_res32(m, L->ctx, L->wo_t[j], L->bo_t[j], yf, z2f, L->z2b, B, d, c.n_head * d, st));     // W_o,h W_v,h on the context rows
            else RC(linear_res32(m, L->ao, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), yf, z2f, L->z2b, B, d, d, st));
            RC(linear_norm_a32(m, L->z2b, z2f, L->fw[2][j], L->fu[2][j], L->fv[2][j], PF(pb + D_N2_W), PF(pb + D_N2_B), c.eps_layer, xf,
                               L->ff, ff, B, ff, d, act, st));
            RC(gelu_ff());
            RC(linear_res32(m, L->ff, PL(pb + D_L2_W), PF(pb + D_L2_B), xf, zf, L->zb, B, d, ff, st));
            if (part == n_parts - 1 && !(fuse_tail && d <= 512 && (d & 3) == 0))
                RC(pa_layernorm_fwd(xf, zf, PF(pb + D_N3_W), PF(pb + D_N3_B), L->mean, L->rstd, B, d, c.eps_layer, PA_F32, st));
        }
        if (part == n_parts - 1) {
            const bool tail_norm = fuse_tail && d <= 512 && (d & 3) == 0;
            if (tail_norm) {
                const int pl_ = m->dec_base(c.n_dec - 1);
                PA_LAUNCH(dec_tail_norm_kernel, dim3((B + 3) / 4), dim3(256), 0, s, L->hf, (bf16*)L->h, (const float*)zf, PF(pl_ + D_N3_W),
                          PF(pl_ + D_N3_B), c.eps_layer, PF(m->dec_norm()), PF(m->dec_norm() + 1), c.eps_final, B, d);
            } else {
                RC(pa_layernorm_fwd(L->hf, xf, PF(m->dec_norm()), PF(m->dec_norm() + 1), L->mean, L->rstd, B, d, c.eps_final, PA_F32, st));
            }
            const int tl = m->tail(), ldv = (c.vocab + 7) / 8 * 8;
            {   // vocabulary head in f32 on the f32 hidden rows (f32 master weight; the f32 skinny kernel)
                pa_gemm_args g; memset(&g, 0, sizeof(g));
                g.A = L->hf; g.B = PF(tl + T_VOCAB_W); g.C = L->vlog; g.bias = PF(tl + T_VOCAB_B);
                g.M = B; g.N = c.vocab; g.K = d; g.lda = d; g.ldb = d; g.ldc = ldv;
                g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = PA_F32; g.out_dtype = PA_F32;
                g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = 1;
                RC(pa_gemm(&g, st));
            }
            if (!tail_norm)
                PA_LAUNCH(dec_cast_kernel, dim3((B * d / 4 + 255) / 256), dim3(256), 0, s, (bf16*)L->h, (const float*)L->hf, (int64_t)B * d / 4);
            RC(linear(m, L->h, PL(tl + T_PTR_W), PF(tl + T_PTR_B), L->pfeat, d, B, d, d, 0, nullptr, -1, st));
            PA_LAUNCH(dec_sample_kernel<T>, dim3(B), dim3(256), 0, s, L->vlog, ldv, (const T*)L->pfeat, (const T*)L->h,
                               (T*)L->hid_cache, PF(tl + T_SW_W), PF(tl + T_SW_B), L->tokens, L->attach, L->first_end, L->t_dev, Tmax, d,
                               c.vocab, c.end, fuse_tail, (float*)L->x, (T*)nullptr, (bf16*)L->xb, PF(P_IN_VALUE), PF(P_Q_COORD), PF(P_Q_POS),
                               c.out_dof);
            if (!fuse_tail) PA_LAUNCH(dec_advance_kernel, dim3(1), dim3(64), 0, s, L->t_dev);
            return 0;
        }
        const int i = part / 2, pb = m->dec_base(i);
        if ((part & 1) == 0 && L->mq_self_bf) {
            // absorbed self-attention: only the query rows are projected (the first d rows of the LayerNorm-folded in_proj); the expand launch
            // forms q~_h = scale log2e W_k,h^T q_h AND appends this step's layer-input row (f32 residual stream -> bf16) to the row cache
            if constexpr (sizeof(T) == 2) {
                if (i > 0) {
                    const int pp = m->dec_base(i - 1);
                    RC(linear_norm_a32(m, L->zb, zf, L->fw[0][i], L->fu[0][i], L->fv[0][i], PF(pp + D_N3_W), PF(pp + D_N3_B), c.eps_layer, xf,
                                       L->q, d, B, d, d, 0, st));
                } else {
                    RC(linear(m, L->xb, PL(pb + D_SA_IN_W), PF(pb + D_SA_IN_B), L->q, d, B, d, d, 0, nullptr, -1, st));
                }
                const int H = c.n_head;
                const bf16* Wk = (const bf16*)m->pl[pb + D_SA_IN_W] + (size_t)d * d;
                const float sl = LOG2E_F / sqrtf((float)(d / H));
                PA_LAUNCH((mq_expand_q_kernel<64, bf16, float>), dim3((B + MQ_XR - 1) / MQ_XR, H), dim3(512), 0, s, (bf16*)L->qt, (const bf16*)L->q, d, Wk, B, d, H, sl,
                          (const float*)xf, (bf16*)L->self_k[i], (const int32_t*)L->t_dev, Tmax);
                RC(fence_in());
                RC(launch_cross_mq((bf16*)L->ctx, (const bf16*)L->qt, (const bf16*)L->self_k[i], nullptr, nullptr, B, Tmax, H, d, s, L->mq_sp, L->mq_sp_bytes,
                                   L->t_dev));
                RC(fence_out());
            }
            return 0;
        }
        if ((part & 1) == 0) {
            if (i > 0) {
                const int pp = m->dec_base(i - 1);
                RC(linear_norm_a32(m, L->zb, zf, L->fw[0][i], L->fu[0][i], L->fv[0][i], PF(pp + D_N3_W), PF(pp + D_N3_B), c.eps_layer, xf,
                                   L->qkv, 3 * d, B, 3 * d, d, 0, st));
            } else {
                RC(linear(m, L->xb, PL(pb + D_SA_IN_W), PF(pb + D_SA_IN_B), L->qkv, 3 * d, B, 3 * d, d, 0, nullptr, -1, st));
            }
            RC(fence_in());
            RC(launch_attn<T>(m, (T*)L->ao, (const T*)L->qkv, 3 * d, (T*)L->self_k[i], (T*)L->self_v[i], Tmax, nullptr, 0,
                              L->t_dev, B, st, nullptr, (const T*)L->qkv));
            RC(fence_out());
        } else {
            if (L->mq_self_bf) RC(linear_res32(m, L->ctx, L->wo_ts[i], L->bo_ts[i], xf, z2f, L->z2b, B, d, c.n_head * d, st));   // W_o,h W_v,h on the context rows
            else RC(linear_res32(m, L->ao, PL(pb + D_SA_OUT_W), PF(pb + D_SA_OUT_B), xf, z2f, L->z2b, B, d, d, st));
            RC(linear_norm_a32(m, L->z2b, z2f, L->fw[1][i], L->fu[1][i], L->fv[1][i], PF(pb + D_N1_W), PF(pb + D_N1_B), c.eps_layer, yf,
                               L->q, d, B, d, d, 0, st));
            if (L->mq) { RC(cross_mq(m, i, s)); return 0; }
            RC(fence_in());
            RC(launch_attn<T>(m, (T*)L->ao, (const T*)L->q, d, (T*)L->cross_k[i], (T*)L->cross_v[i], S, L->cu ? nullptr : L->kpm, S,
                              L->t_dev, B, st, L->cu));
            RC(fence_out());
        }
        return 0;
    }
    if (part >= 2 && (part & 1) == 0) {
        // the feed-forward block of the previous layer (everything after its cross-attention)
        const int j = part / 2 - 1, pb = m->dec_base(j);
        if (fold) {
            // z2 = ao Wo^T + b + y1;  ff = relu(norm2(z2) W1^T + b1), y2 -> x;  z3 = ff W2^T + b2 + y2 -> z
            if (L->mq && L->mq_contract) {
                PA_LAUNCH(mq_contract_v_kernel<T>, dim3((B + MQ_XR - 1) / MQ_XR, c.n_head), dim3(512), 0, s, (T*)L->ao, d, (const T*)L->ctx,
                          (const T*)L->wo_t[j], PF(pb + D_CA_IN_B) + 2 * d, B, d, c.n_head);
                RC(linear(m, L->ao, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), L->z2, d, B, d, d, 0, L->y, -1, st));
            } else if (L->mq) RC(linear(m, L->ctx, L->wo_t[j], L->bo_t[j], L->z2, d, B, d, c.n_head * d, 0, L->y, -1, st));
            else RC(linear(m, L->ao, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), L->z2, d, B, d, d, 0, L->y, -1, st));
            RC(linear_norm_a(m, L->z2, L->fw[2][j], L->fu[2][j], L->fv[2][j], PF(pb + D_N2_W), PF(pb + D_N2_B), c.eps_layer, L->x,
                             L->ff, ff, B, ff, d, act, st));
            RC(gelu_ff());
            RC(linear(m, L->ff, PL(pb + D_L2_W), PF(pb + D_L2_B), L->z, d, B, d, ff, 0, L->x, -1, st));
            if (part == n_parts - 1)          // the last layer's norm3 has no Linear behind it: explicit
                RC(pa_layernorm_fwd(L->x, L->z, PF(pb + D_N3_W), PF(pb + D_N3_B), L->mean, L->rstd, B, d, c.eps_layer, c.dtype, st));
        } else {
            RC(linear_ln(m, L->ao, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), L->y, L->z, L->x, PF(pb + D_N2_W), PF(pb + D_N2_B), c.eps_layer, B, d, st));
            RC(linear(m, L->x, PL(pb + D_L1_W), PF(pb + D_L1_B), L->ff, ff, B, ff, d, act, nullptr, -1, st));
            RC(gelu_ff());
            RC(linear_ln(m, L->ff, PL(pb + D_L2_W), PF(pb + D_L2_B), L->x, L->z, L->x, PF(pb + D_N3_W), PF(pb + D_N3_B), c.eps_layer, B, ff, st));
        }
    }
    if (part == n_parts - 1) {
        RC(pa_layernorm_fwd(L->h, L->x, PF(m->dec_norm()), PF(m->dec_norm() + 1), L->mean, L->rstd, B, d, c.eps_final, c.dtype, st));
        const int tl = m->tail(), ldv = (c.vocab + 7) / 8 * 8;
        RC(linear(m, L->h, PL(tl + T_VOCAB_W), PF(tl + T_VOCAB_B), L->vlog, ldv, B, c.vocab, d, 0, nullptr, PA_F32, st));
        RC(linear(m, L->h, PL(tl + T_PTR_W), PF(tl + T_PTR_B), L->pfeat, d, B, d, d, 0, nullptr, -1, st));
        PA_LAUNCH(dec_sample_kernel<T>, dim3(B), dim3(256), 0, s, L->vlog, ldv, (const T*)L->pfeat, (const T*)L->h,
                           (T*)L->hid_cache, PF(tl + T_SW_W), PF(tl + T_SW_B), L->tokens, L->attach, L->first_end, L->t_dev, Tmax, d,
                           c.vocab, c.end, fuse_tail, (float*)nullptr, (T*)L->x, (bf16*)nullptr, PF(P_IN_VALUE), PF(P_Q_COORD), PF(P_Q_POS),
                           c.out_dof);
        if (!fuse_tail) PA_LAUNCH(dec_advance_kernel, dim3(1), dim3(64), 0, s, L->t_dev);
        return 0;
    }
    const int i = part / 2;
    const int pb = m->dec_base(i);
    if ((part & 1) == 0) {
        if (L->mq_self) {                      // only the query rows are projected; K / V never exist (self_mq)
            if (fold && i > 0) {
                const int pp = m->dec_base(i - 1);
                RC(linear_norm_a(m, L->z, L->fw[0][i], L->fu[0][i], L->fv[0][i], PF(pp + D_N3_W), PF(pp + D_N3_B), c.eps_layer, L->x,
                                 L->q, d, B, d, d, 0, st));
            } else {
                RC(linear(m, x, PL(pb + D_SA_IN_W), PF(pb + D_SA_IN_B), L->q, d, B, d, d, 0, nullptr, -1, st));
            }
            return self_mq(m, i, L->x, s);
        }
        if (fold && i > 0) {                   // x = norm3(z) of the layer before, materialised by this launch for the residual add
            const int pp = m->dec_base(i - 1);
            RC(linear_norm_a(m, L->z, L->fw[0][i], L->fu[0][i], L->fv[0][i], PF(pp + D_N3_W), PF(pp + D_N3_B), c.eps_layer, L->x,
                             L->qkv, 3 * d, B, 3 * d, d, 0, st));
        } else {
            RC(linear(m, x, PL(pb + D_SA_IN_W), PF(pb + D_SA_IN_B), L->qkv, 3 * d, B, 3 * d, d, 0, nullptr, -1, st));
        }
        // (the attention kernel appends this step's K / V rows to the caches itself)
        RC(fence_in());
        RC(launch_attn<T>(m, (T*)L->ao, (const T*)L->qkv, 3 * d, (T*)L->self_k[i], (T*)L->self_v[i], Tmax, nullptr, 0,
                          L->t_dev, B, st, nullptr, (const T*)L->qkv));
        RC(fence_out());
    } else {
        if (fold) {                            // z1 = ao Wo^T + b + x -> z2;  q = norm1(z1) Wq^T + bq, y1 -> y
            RC(linear(m, L->ao, PL(pb + D_SA_OUT_W), PF(pb + D_SA_OUT_B), L->z2, d, B, d, d, 0, x, -1, st));
            RC(linear_norm_a(m, L->z2, L->fw[1][i], L->fu[1][i], L->fv[1][i], PF(pb + D_N1_W), PF(pb + D_N1_B), c.eps_layer, L->y,
                             L->q, d, B, d, d, 0, st));
            if (L->mq) { RC(cross_mq(m, i, s)); return 0; }
        } else {
            RC(linear_ln(m, L->ao, PL(pb + D_SA_OUT_W), PF(pb + D_SA_OUT_B), x, L->z, L->y, PF(pb + D_N1_W), PF(pb + D_N1_B), c.eps_layer, B, d, st));
            RC(linear(m, L->y, PL(pb + D_CA_IN_W), PF(pb + D_CA_IN_B), L->q, d, B, d, d, 0, nullptr, -1, st));
        }
        RC(fence_in());
        RC(launch_attn<T>(m, (T*)L->ao, (const T*)L->q, d, (T*)L->cross_k[i], (T*)L->cross_v[i], S, L->cu ? nullptr : L->kpm, S,
                          L->t_dev, B, st, L->cu));
        RC(fence_out());
    }
    return 0;
}

template <typename T>
int step_impl(pa_model* m, void* st) {
    for (int part = 0; part <= 2 * m->cfg.n_dec; ++part) RC(step_part<T>(m, part, st, nullptr, nullptr));
    return 0;
}

int step_part_any(pa_model* m, int part, void* st, hipEvent_t w, hipEvent_t r) {
    return m->cfg.dtype == PA_BF16 ? step_part<bf16>(m, part, st, w, r) : step_part<float>(m, part, st, w, r);
}

}  // namespace

void pa_decode_free_layout(pa_model* m) {
    if (m && m->dec) { for (hipEvent_t e : m->dec->pair_ev) (void)hipEventDestroy(e); delete m->dec; m->dec = nullptr; }
}

extern "C" int64_t pa_decode_ws_bytes(pa_model* m, int32_t B, int32_t S, int32_t Tmax) {
    if (!m || B <= 0 || S <= 0 || Tmax <= 0 || Tmax > MAX_T) return PA_EINVAL;
    DecodeLayout tmp;
    return (int64_t)dec_layout(m, &tmp, nullptr, B, S, Tmax) + 256;
}

// Requires a preceding encoder-only pa_model_train_fwd (batch.output_value == NULL) on the same stream:
// uses its memory, batch size, sequence length and input_mask.
extern "C" int pa_decode_begin(pa_model* m, void* ws, int64_t ws_bytes, int32_t Tmax, void* stream) {
    if (!m || !m->bound || !ws || m->B <= 0 || Tmax <= 0 || Tmax > MAX_T) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return PA_EALIGN;
    if (!m->dec) m->dec = new (std::nothrow) DecodeLayout();
    if (!m->dec) return PA_EINVAL;
    DecodeLayout* L = m->dec;
    const size_t need = dec_layout(m, L, (char*)ws, m->B, m->S, Tmax);
    if ((int64_t)need > ws_bytes) return PA_EINVAL;
    const pa_model_cfg& c = m->cfg;
    const int d = c.d_model, B = m->B, S = m->S;
    const size_t e = c.dtype == PA_BF16 ? 2 : 4;
    const void* memory = c.has_enc_norm ? m->memory : m->X[c.n_enc];
    hipStream_t s = (hipStream_t)stream;
    const DecodeModes md = decode_modes(c, B, S);
    L->fold = md.fold; L->f32res = md.f32res; L->mq = md.mq;           // (what each is and where it was measured: decode_modes)
    static const int contract_env = getenv("PLANK_DECODE_MQ_CONTRACT") ? atoi(getenv("PLANK_DECODE_MQ_CONTRACT")) : -1;
    L->mq_contract = L->mq && d / c.n_head == 64 && (contract_env >= 0 ? contract_env != 0 : c.dtype == PA_F32);
    // exact f32: the self-attention in the same form - the step caches the layer-input rows x_t instead of K and V (half the bytes of what is
    // then the largest stream of the f32 step), q~ = W_k^T q, W_v behind the softmax (q . b_k cancels, b_v is added once).  PLANK_DECODE_MQ_SELF=0.
    static const int self_env = getenv("PLANK_DECODE_MQ_SELF") ? atoi(getenv("PLANK_DECODE_MQ_SELF")) : 1;
    L->mq_self = L->mq && L->mq_contract && c.dtype == PA_F32 && self_env != 0 && Tmax <= 16000;
    // bf16 (round 6; needs the f32 residual stream's step form): pays from ~200 batch elements on - measured in one session, B 256 x 1024 steps:
    // 0.969 -> 0.929 ms / step (264 k -> 276 k tokens/s; half the bytes of the self-attention stream against one more launch and a K = H d
    // out-projection per layer); B 64: 0.557 -> 0.662, B 16: 0.476 -> 0.609 (profiles/r06_decode_self_absorbed.txt).
    // PLANK_DECODE_MQ_SELF_BF16: 0 never, 1 always, unset = B >= PLANK_DECODE_MQ_SELF_MINB (200).
    static const int self_bf_env = getenv("PLANK_DECODE_MQ_SELF_BF16") ? atoi(getenv("PLANK_DECODE_MQ_SELF_BF16")) : -1;
    static const int self_bf_minb = getenv("PLANK_DECODE_MQ_SELF_MINB") ? atoi(getenv("PLANK_DECODE_MQ_SELF_MINB")) : 200;
    L->mq_self_bf = L->mq && !L->mq_contract && c.dtype == PA_BF16 && md.f32res && d / c.n_head == 64 && Tmax <= MQ_MAXS &&
                    (self_bf_env >= 0 ? self_bf_env != 0 : B >= self_bf_minb);
    if (L->mq) {
        // absorbed cross-attention: the step reads the encoder output rows themselves - no K / V projection of the memory at all
        hipError_t hm = hipMemcpyAsync(L->mem, memory, (size_t)m->NE * d * e, hipMemcpyDeviceToDevice, s);
        if (hm != hipSuccess) return (int)hm;
    }
    for (int i = 0; i < c.n_dec && !L->mq; ++i) {       // cross-attention K/V of the memory: once per sequence, not per step
        const int pb = m->dec_base(i);
        RC(linear(m, memory, (const char*)m->pl[pb + D_CA_IN_W] + (size_t)d * d * e, (const float*)m->pf[pb + D_CA_IN_B] + d,
                  L->kv_tmp, 2 * d, m->NE, 2 * d, d, 0, nullptr, -1, stream));
        if (c.dtype == PA_BF16)
            PA_LAUNCH(dec_split_heads_kernel<bf16>, dim3(2048), dim3(256), 0, s, (bf16*)L->cross_k[i], (bf16*)L->cross_v[i],
                      (const bf16*)L->kv_tmp, (int64_t)m->NE, S, d, c.n_head, m->batch.cu_in, m->batch.rowmap);
        else
            PA_LAUNCH(dec_split_heads_kernel<float>, dim3(2048), dim3(256), 0, s, (float*)L->cross_k[i], (float*)L->cross_v[i],
                      (const float*)L->kv_tmp, (int64_t)m->NE, S, d, c.n_head, m->batch.cu_in, m->batch.rowmap);
    }
    const bool fold32 = L->fold && c.dtype == PA_F32;
    if (L->fold) {
        for (int i = 0; i < c.n_dec; ++i) {
            const int pb = m->dec_base(i);
            auto F = [&](int idx) { return (const float*)m->pf[idx]; };
            auto foldw = [&](void* wf, float* u, float* v, const float* W, const float* bias, const float* g_, const float* b_, int N) -> int {
                return fold32 ? pa_ln_fold_weights_f32((float*)wf, u, v, W, bias, g_, b_, N, d, stream)
                              : pa_ln_fold_weights(wf, u, v, W, bias, g_, b_, N, d, stream);
            };
            if (i > 0) {
                const int pp = m->dec_base(i - 1);
                RC(foldw(L->fw[0][i], L->fu[0][i], L->fv[0][i], F(pb + D_SA_IN_W), F(pb + D_SA_IN_B), F(pp + D_N3_W), F(pp + D_N3_B), 3 * d));
            }
            RC(foldw(L->fw[1][i], L->fu[1][i], L->fv[1][i], F(pb + D_CA_IN_W), F(pb + D_CA_IN_B), F(pb + D_N1_W), F(pb + D_N1_B), d));
            if (L->mq && L->mq_contract) {     // the head-transposed copy of W_v for mq_contract_v_kernel (in the wo_t buffer, unused in this form)
                if (c.dtype == PA_BF16)
                    PA_LAUNCH(mq_transpose_v_kernel<bf16>, dim3(1024), dim3(256), 0, s, (bf16*)L->wo_t[i], (const bf16*)m->pl[pb + D_CA_IN_W] + (size_t)2 * d * d, d, c.n_head);
                else
                    PA_LAUNCH(mq_transpose_v_kernel<float>, dim3(1024), dim3(256), 0, s, (float*)L->wo_t[i], (const float*)m->pl[pb + D_CA_IN_W] + (size_t)2 * d * d, d, c.n_head);
            }
            if (L->mq_self)
                PA_LAUNCH(mq_transpose_v_kernel<float>, dim3(1024), dim3(256), 0, s, (float*)L->wvt_self[i], (const float*)m->pl[pb + D_SA_IN_W] + (size_t)2 * d * d, d, c.n_head);
            if (L->mq_self_bf)
                PA_LAUNCH(mq_absorb_o_kernel<bf16>, dim3(d), dim3(256), 0, s, (bf16*)L->wo_ts[i], L->bo_ts[i], F(pb + D_SA_OUT_W), F(pb + D_SA_OUT_B),
                          F(pb + D_SA_IN_W), F(pb + D_SA_IN_B), d, c.n_head);
            if (L->mq && !L->mq_contract) {    // W~o = W_o,h W_v,h, b~o = b_o + W_o b_v for the Linear behind the absorbed attention (csrc/decode_mq.h)
                if (c.dtype == PA_BF16)
                    PA_LAUNCH(mq_absorb_o_kernel<bf16>, dim3(d), dim3(256), 0, s, (bf16*)L->wo_t[i], L->bo_t[i], F(pb + D_CA_OUT_W), F(pb + D_CA_OUT_B),
                              F(pb + D_CA_IN_W), F(pb + D_CA_IN_B), d, c.n_head);
                else
                    PA_LAUNCH(mq_absorb_o_kernel<float>, dim3(d), dim3(256), 0, s, (float*)L->wo_t[i], L->bo_t[i], F(pb + D_CA_OUT_W), F(pb + D_CA_OUT_B),
                              F(pb + D_CA_IN_W), F(pb + D_CA_IN_B), d, c.n_head);
            }
            RC(foldw(L->fw[2][i], L->fu[2][i], L->fv[2][i], F(pb + D_L1_W), F(pb + D_L1_B), F(pb + D_N2_W), F(pb + D_N2_B), c.d_ff));
        }
    }
    L->cu = nullptr;
    if (m->batch.cu_in) {
        hipError_t hc = hipMemcpyAsync(L->cu_store, m->batch.cu_in, (size_t)(B + 1) * 4, hipMemcpyDeviceToDevice, s);
        if (hc != hipSuccess) return (int)hc;
        L->cu = L->cu_store;
    }
    hipError_t he = hipMemcpyAsync(L->kpm, m->batch.input_mask, (size_t)B * S, hipMemcpyDeviceToDevice, s);
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(L->first_end, 0xFF, (size_t)B * 4, s);
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(L->t_dev, 0, 8, s);                     // step counter and the sampling kernel's ticket
    if (he != hipSuccess) return (int)he;
    if (L->mq_sp) {                                             // the range blocks' tickets (every launch leaves them zero again)
        he = hipMemsetAsync(L->mq_sp, 0, ((size_t)B * 4 + 255) / 256 * 256, s);
        if (he != hipSuccess) return (int)he;
    }
    he = hipMemsetAsync(L->x, 0, (size_t)B * c.d_model * 4, s);  // input embedding of step 0: zeros (later steps: written by the sampling kernel)
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(L->xb, 0, (size_t)B * c.d_model * 2, s);
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(L->tokens, 0, (size_t)B * Tmax * 8, s);
    if (he != hipSuccess) return (int)he;
    he = hipMemsetAsync(L->attach, 0xFF, (size_t)B * Tmax * 8, s);
    return he == hipSuccess ? 0 : (int)he;
}

extern "C" int pa_decode_step(pa_model* m, void* stream) {
    if (!m || !m->dec || m->dec->B <= 0) return PA_EINVAL;
    return m->cfg.dtype == PA_BF16 ? step_impl<bf16>(m, stream) : step_impl<float>(m, stream);
}

// Two half-batches of one decode ("lanes" A and B: two handles over the same parameters, each with its own pa_decode_begin
// state) advance one step on two streams with their attention launches strictly alternating: A's k-th attention, B's k-th,
// A's (k+1)-th ...  While one lane streams its K/V caches, the other runs the latency-bound launches between two attentions;
// without the alternation both lanes reach their attention launches together and only share the HBM bandwidth
// (measured: 6.6 % overlap, +1 %).  MEASURED SLOWER than the unsynchronised lanes on MI355X / ROCm 7.2 (B 256: 2.15 ms per step
// against 1.23): the 24 cross-queue event edges per step cost more than the overlap they arrange, so the host side keeps it
// behind PLANK_DECODE_ALTERNATE=1.  Works eagerly and under stream capture (stream_b must already be forked from stream_a's
// capture; the caller joins them afterwards).  No reference counterpart (models.py:267-323 decodes one batch serially).
extern "C" int pa_decode_step_pair(pa_model* a, pa_model* b, void* stream_a, void* stream_b) {
    if (!a || !b || !a->dec || !b->dec || a->dec->B <= 0 || b->dec->B <= 0) return PA_EINVAL;
    if (a->cfg.n_dec != b->cfg.n_dec || a->cfg.dtype != b->cfg.dtype) return PA_EINVAL;
    const int n = 2 * a->cfg.n_dec;
    std::vector<hipEvent_t>& ev = a->dec->pair_ev;
    if ((int)ev.size() != 2 * n) {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.assign(2 * n, nullptr);
        for (int i = 0; i < 2 * n; ++i)
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) { ev.clear(); return PA_EINVAL; }
    }
    for (int k = 0; k < n; ++k) {
        RC(step_part_any(a, k, stream_a, k > 0 ? ev[n + k - 1] : nullptr, ev[k]));      // after B's attention k - 1
        RC(step_part_any(b, k, stream_b, ev[k], ev[n + k]));                              // after A's attention k
    }
    RC(step_part_any(a, n, stream_a, nullptr, nullptr));
    RC(step_part_any(b, n, stream_b, nullptr, nullptr));
    return 0;
}

// The absorbed cross-attention launch on its own (csrc/decode_mq.h; kernel tests / tools): ctx [B][H][512] (bf16) = per head the
// softmax(qt_h . m_s)-weighted sum of the memory rows m_s; qt [B][H][512] carries scale * log2 e; mem [rows][512] bf16 - dense
// [B][S] rows with the optional key-padding mask kpm [B][S] (1 = PAD), or packed rows with cu [B + 1].
extern "C" int pa_dec_cross_mq(void* ctx, const void* qt, const void* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                               int32_t S, int32_t H, int32_t d, void* stream) {
    if (!ctx || !qt || !mem) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(mem)) & 15) return PA_EALIGN;
    return launch_cross_mq((bf16*)ctx, (const bf16*)qt, (const bf16*)mem, kpm, cu, B, S, H, d, (hipStream_t)stream);
}

// The same launch with scratch for range blocks (decode_mq.h: at small batches a batch element's keys are walked by several blocks so
// that the launch uses the whole chip).  ws: pa_dec_cross_mq_ws_bytes(B, S) bytes, 256-byte aligned, its first ceil(4 B / 256) * 256
// bytes zero before the first launch (launches leave them zero); with ws = NULL or too small it is pa_dec_cross_mq.
extern "C" int64_t pa_dec_cross_mq_ws_bytes(int32_t B, int32_t S) { return (B > 0 && S > 0) ? mq_split_bytes(B, mq_parts(B, S)) : 0; }
extern "C" int pa_dec_cross_mq_ws(void* ctx, const void* qt, const void* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                                  int32_t S, int32_t H, int32_t d, void* ws, int64_t ws_bytes, void* stream) {
    if (!ctx || !qt || !mem) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(mem)) & 15) return PA_EALIGN;
    return launch_cross_mq((bf16*)ctx, (const bf16*)qt, (const bf16*)mem, kpm, cu, B, S, H, d, (hipStream_t)stream, ws, ws_bytes);
}

extern "C" int pa_dec_cross_mq32(float* ctx, const float* qt, const float* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                                 int32_t S, int32_t H, int32_t d, void* stream) {
    if (!ctx || !qt || !mem) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(mem) | reinterpret_cast<uintptr_t>(ctx)) & 15) return PA_EALIGN;
    return launch_cross_mq32(ctx, qt, mem, kpm, cu, B, S, H, d, (hipStream_t)stream);
}

extern "C" int pa_dec_cross_mq32_ws(float* ctx, const float* qt, const float* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                                    int32_t S, int32_t H, int32_t d, void* ws, int64_t ws_bytes, void* stream) {
    if (!ctx || !qt || !mem) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(mem) | reinterpret_cast<uintptr_t>(ctx)) & 15) return PA_EALIGN;
    return launch_cross_mq32(ctx, qt, mem, kpm, cu, B, S, H, d, (hipStream_t)stream, nullptr, ws, ws_bytes);
}

extern "C" int pa_dec_self_mq32(float* ctx, const float* qt, const float* xcache, const int32_t* t_dev, int32_t B, int32_t Tmax,
                                int32_t H, int32_t d, void* stream) {
    if (!ctx || !qt || !xcache || !t_dev) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(qt) | reinterpret_cast<uintptr_t>(xcache) | reinterpret_cast<uintptr_t>(ctx)) & 15) return PA_EALIGN;
    return launch_cross_mq32(ctx, qt, xcache, nullptr, nullptr, B, Tmax, H, d, (hipStream_t)stream, t_dev);
}

extern "C" int pa_decode_buffers(pa_model* m, void** tokens, void** attach, void** first_end, void** t_dev) {
    if (!m || !m->dec || !tokens || !attach || !first_end || !t_dev) return PA_EINVAL;
    *tokens = m->dec->tokens; *attach = m->dec->attach; *first_end = m->dec->first_end; *t_dev = m->dec->t_dev;
    return 0;
}
