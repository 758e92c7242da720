// Model-level runtime for the PlankAssembly hot path on MI355X: sequences the gfx950 kernels of
// gemm.hip / attention.hip / rowops.hip for a full training forward (reference
// plankassembly/models.py:190-233, train_step) and a hand-derived backward, over one caller-owned
// workspace ("arena").  Host-only logic: no device allocation, no synchronisation, enqueue-only
// (hipGraph capturable).  See include/plank_hip.h (pa_model_*).
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <new>
#include <string.h>
#include <vector>
#include "../../include/plank_hip.h"

#include "model.h"

namespace {

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

struct Ctx {   // convenience wrapper for kernel calls in the model's dtype
    pa_model* m; void* st;
    int dt() const { return m->cfg.dtype; }
    bool gelu() const { return m->cfg.activation == 2; }
    // hidden = drop(activation(A W1^T + b1)): the FFN's first Linear.  ReLU: one launch (activation and dropout in the GEMM epilogue).
    // GELU (ACTIVATION: gelu - no shipped config; reference models.py:60-61,66-67): the Linear writes the pre-activation and
    // pa_gelu_fwd turns it into drop(gelu(.)) in place - the GEMM epilogues stay ReLU-only (erf in every epilogue variant
    // doubled the library and touched the register budgets of the benchmarked kernels).
    int ffn1(const void* A, const void* W, const float* bias, void* hidden, int rows, float drop_p, uint32_t seed) const {
        const int d = m->cfg.d_model, ff = m->cfg.d_ff;
        if (!gelu()) return linear(A, d, W, bias, hidden, ff, rows, ff, d, 1, drop_p, seed);
        RC(linear(A, d, W, bias, hidden, ff, rows, ff, d, 0));
        return pa_gelu_fwd(hidden, hidden, rows, ff, ff, dt(), drop_p, seed, st);
    }

    // C[M,N] = epi(A[M,K] x W^T) with W a torch Linear weight [N][K]
    int linear(const void* A, int lda, const void* W, const float* bias, void* Cout, int ldc, int M, int N, int K,
               int relu = 0, float drop_p = 0.f, uint32_t seed = 0, const void* R = nullptr, int ldr = 0,
               int out_dtype = -1) const {
        pa_gemm_args g; memset(&g, 0, sizeof(g));
        g.A = A; g.B = W; g.C = Cout; g.bias = bias; g.R = R;
        g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldr = ldr;
        g.batch = 1; g.a_kcontig = 1; g.b_kcontig = 1;
        g.in_dtype = dt(); g.out_dtype = out_dtype < 0 ? dt() : out_dtype;
        g.alpha = 1.f; g.relu = relu; g.aux_scale = 1.f; g.drop_p = drop_p; g.drop_seed = seed; g.splitk = 1;
        return pa_gemm(&g, st);
    }
    // dX[M,K] = dY[M,N] x W[N][K]  (+ R)   (optionally relu-backward gated by aux).
    // WT (optional): transposed weight, element (k, n) at WT[k * ldwt + n]  -> both operands k-contiguous.
    int linear_dx(const void* dY, int lddy, const void* W, int ldw, void* dX, int lddx, int M, int N, int K,
                  const void* R = nullptr, int ldr = 0, const void* aux = nullptr, int ldaux = 0,
                  float aux_scale = 1.f, const void* WT = nullptr, int ldwt = 0) const {
        pa_gemm_args g; memset(&g, 0, sizeof(g));
        g.A = dY; g.B = WT ? WT : W; g.C = dX; g.R = R; g.aux = aux;
        g.M = M; g.N = K; g.K = N; g.lda = lddy; g.ldb = WT ? ldwt : ldw; g.ldc = lddx; g.ldr = ldr; g.ldaux = ldaux;
        g.batch = 1; g.a_kcontig = 1; g.b_kcontig = WT ? 1 : 0;
        g.in_dtype = dt(); g.out_dtype = dt();
        g.alpha = 1.f; g.aux_scale = aux_scale; g.splitk = 1;
        return pa_gemm(&g, st);
    }
    // dW[N][K] = dY[M,N]^T x X[M,K] (f32 out, split-K over the rows), db[N] = colsum(dY) unless db == NULL
    int linear_dw(const void* dY, int lddy, const void* Xin, int ldx, float* dW, float* db, int M, int N, int K) const {
        pa_gemm_args g; memset(&g, 0, sizeof(g));
        g.A = dY; g.B = Xin; g.C = dW;
        g.M = N; g.N = K; g.K = M; g.lda = lddy; g.ldb = ldx; g.ldc = K;
        g.batch = 1; g.a_kcontig = 0; g.b_kcontig = 0;
        g.in_dtype = dt(); g.out_dtype = PA_F32;
        g.alpha = 1.f; g.aux_scale = 1.f;
        const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
        const bool x3 = dt() == PA_F32 && pa_gemm_split_active();          // bf16x3: the bf16 kernels over 3 M stacked rows
        const int ktile = (dt() == PA_BF16 || x3) ? 64 : 16;
        const int nkt = ((x3 ? 3 * M : M) + ktile - 1) / ktile;
        static const bool plan = !(getenv("PA_DW_PLAN") && atoi(getenv("PA_DW_PLAN")) == 0);
        if (plan && m->defer_ok && tiles < 256 && m->ndwq < PA_MAX_GROUP) {
            // member of the segment's grouped launch: its split is chosen at the flush, when all members are known
            // (plan_group: one round of the one-block-per-CU kernel with the longest K slice as short as possible)
            g.splitk = 0; g.ws = nullptr;
            m->dwq[m->ndwq++] = g;
            return queue_bias(dY, lddy, db, M, N);
        }
        int sk = 1;
        if (tiles < 256) {
            // a lone launch wants one round of the one-block-per-CU kernel (256 tiles x slices); the members of a grouped
            // launch share the CUs, so each is split 4x less (measured: 64 -> 6.61 ms/step, 256 -> 6.73, 32 -> 6.63)
            static const int target_env = getenv("PA_DW_UNITS") ? atoi(getenv("PA_DW_UNITS")) : 0;
            const int target = target_env > 0 ? target_env : (m->defer_ok ? 64 : 256);
            sk = target / tiles > 0 ? target / tiles : 1;
            if (sk > 16) sk = 16;
            if (sk > nkt / 4) sk = nkt / 4 > 0 ? nkt / 4 : 1;
            while (sk > 1 && (size_t)sk * N * K > m->splitws_floats) --sk;
        }
        sk = pa_gemm_effective_splitk(M, dt(), sk);             // slabs actually written
        g.splitk = sk; g.ws = m->splitws;
        // queue the reduction: all weight gradients of a segment are reduced by one launch at its end (flush_reduce)
        const size_t need = (size_t)sk * N * K;
        if (sk > 1 && m->ndefer < PA_MAX_REDUCE && m->slab_used + need <= m->splitws_floats && (K & 3) == 0) {
            g.ws = m->splitws + m->slab_used;
            g.splitk_defer = 1;
            pa_reduce_desc& rd = m->defer[m->ndefer++];
            rd.ws = m->splitws + m->slab_used; rd.out = dW; rd.rows = N; rd.cols = K; rd.ld_out = K; rd.splitk = sk;
            m->slab_used += (need + 3) / 4 * 4;
        }
        if (m->defer_ok && g.splitk_defer == (sk > 1 ? 1 : 0) && m->ndwq < PA_MAX_GROUP) {
            m->dwq[m->ndwq++] = g;                               // launched with the segment's other weight gradients (flush)
        } else {
            RC(pa_gemm(&g, st));
        }
        return queue_bias(dY, lddy, db, M, N);
    }
    // db[N] += colsum(dY[M][N])  (queued for the segment tail when the buffer stays untouched until then)
    int queue_bias(const void* dY, int lddy, float* db, int M, int N) const {
        if (db) {   // gradients are zero-initialised by the caller: accumulate
            const int EB = dt() == PA_BF16 ? 8 : 4;
            const bool vec = (reinterpret_cast<uintptr_t>(dY) & 15) == 0 && lddy % EB == 0 && lddy >= (N + EB - 1) / EB * EB;
            if (m->defer_ok && vec && m->ncs < PA_MAX_COLSUM) {
                // inside a layer segment every dY buffer is written once: sum all its bias gradients in one launch at the end
                pa_colsum_desc& cd = m->cs[m->ncs++];
                cd.X = dY; cd.out = db; cd.M = M; cd.N = N; cd.ldx = lddy; cd.pad_ = 0;
            } else {
                RC(pa_colsum(dY, dt(), M, N, lddy, db, 1, m->partial, st));
            }
        }
        return 0;
    }
    // z = R + drop(A W^T + bias), y = LayerNorm(z): the post-norm sublayer tail of a decoder layer.  PLANK_TRAIN_FUSE_LN=<rows>
    // sends launches of at most that many rows to the one-launch row-block kernel pa_gemm_ln (bit-identical to the two
    // launches, round 2).  OFF by default: measured in round 4 at the decoder's 2 048 rows (18 sites per step) the step is
    // 5.29 ms with it against 5.07 ms without - 64 blocks each pulling a whole 512 x K weight through their own CU take ~26 us
    // where gemm3s + LayerNorm take ~14 (the same per-CU weight streaming that made it lose at 256 and at 7 940 rows).
    int linear_ln(const void* A, int lda, const void* W, const float* bias, const void* R, void* z, void* y, const float* g,
                  const float* b, float* mean, float* rstd, int M, int K, float drop_p, uint32_t seed, float eps) const {
        static const int fuse_max = getenv("PLANK_TRAIN_FUSE_LN") ? atoi(getenv("PLANK_TRAIN_FUSE_LN")) : 0;
        const int d = m->cfg.d_model;
        if (dt() == PA_BF16 && d == 512 && M <= fuse_max && M <= pa_gemm_ln_max_rows() && K % 64 == 0) {
            pa_gemm_ln_args a; memset(&a, 0, sizeof(a));
            a.A = A; a.W = W; a.bias = bias; a.R = R; a.Z = z; a.Y = y; a.gamma = g; a.beta = b; a.mean = mean; a.rstd = rstd;
            a.M = M; a.N = d; a.K = K; a.lda = lda; a.ldw = K; a.ldr = d; a.ldz = d; a.ldy = d;
            a.eps = eps; a.drop_p = drop_p; a.drop_seed = seed;
            return pa_gemm_ln(&a, st);
        }
        RC(linear(A, lda, W, bias, z, d, M, d, K, 0, drop_p, seed, R, d));
        return ln_fwd(y, z, g, b, mean, rstd, M, eps);
    }
    int ln_fwd(void* y, const void* z, const float* g, const float* b, float* mean, float* rstd, int64_t rows, float eps) const {
        // bf16x3 mode: the Linear that consumes y finds its cut image written here (pa_gemm_split_reserve)
        int32_t pat = 0;
        void* img = dt() == PA_F32 ? pa_gemm_split_reserve(y, (int32_t)rows, m->cfg.d_model, m->cfg.d_model, &pat) : nullptr;
        return pa_layernorm_fwd_img(y, z, g, b, mean, rstd, rows, m->cfg.d_model, eps, dt(), img, pat, st);
    }
    int ln_bwd(void* dz, void* ddrop, const void* dy, const void* z, const float* g, const float* mean, const float* rstd,
               float* dg, float* db, float* dzsum, int64_t rows, float drop_p, uint32_t seed) const {
        if (m->defer_ok && m->nlnq < PA_MAX_LN_FINISH && m->nlnq < 3) {
            // inside a layer segment: partial sums now (own buffer), one finishing launch for all of them at the end
            float* part = m->lnp[m->nlnq];
            // bf16x3 mode: the dX GEMM that consumes ddrop (dz without dropout) finds its cut image written here
            int32_t pat = 0;
            void* const out = ddrop ? ddrop : dz;
            const bool al = ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(z) |
                              reinterpret_cast<uintptr_t>(ddrop) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(part)) & 15) == 0;
            void* img = (al && pa_layernorm_bwd_can_img(m->cfg.d_model, dt())) ?
                        pa_gemm_split_reserve(out, (int32_t)rows, m->cfg.d_model, m->cfg.d_model, &pat) : nullptr;
            RC(pa_layernorm_bwd_partial_img(dz, ddrop, dy, z, g, mean, rstd, dzsum ? 1 : 0, part, rows, m->cfg.d_model, dt(), drop_p, seed,
                                            img, pat, st));
            pa_ln_finish_desc& fd = m->lnq[m->nlnq++];
            fd.partial = part; fd.nparts = pa_layernorm_bwd_nparts(rows); fd.pad_ = 0; fd.dgamma = dg; fd.dbeta = db; fd.dzsum = dzsum;
            return 0;
        }
        return pa_layernorm_bwd(dz, ddrop, dy, z, g, mean, rstd, dg, db, dzsum, m->partial, rows, m->cfg.d_model, dt(),
                                drop_p, seed, st);
    }
    int attn(bool bwd, const void* q, int ldq, const void* k, const void* v, int ldkv, void* o, float* lse,
             const uint8_t* kpm, int Lq, int Lk, int causal, float drop_p, uint32_t seed,
             const void* dout = nullptr, void* dq = nullptr, int lddq = 0, void* dk = nullptr, void* dv = nullptr,
             int lddkv = 0, const int32_t* cu_q = nullptr, const int32_t* cu_k = nullptr) const {
        pa_attn_args a; memset(&a, 0, sizeof(a));
        a.cu_q = cu_q; a.cu_k = cu_k;
        // packed encoder rows: dispatch the batch elements longest first (pa_pack_rows left that order behind cu_in)
        static const bool use_order = !(getenv("PA_ATTN_ORDER") && atoi(getenv("PA_ATTN_ORDER")) == 0);
        if (use_order && cu_k && cu_k == m->batch.cu_in) a.order = cu_k + m->B + 1;
        if (cu_q && cu_q == cu_k) { a.ws = m->attn_ws; a.ws_bytes = m->attn_ws_bytes; }      // packed self-attention: range blocks
        const int d = m->cfg.d_model, H = m->cfg.n_head;
        a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.kpm = kpm;
        a.B = m->B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dh = d / H;
        a.ldq = ldq; a.ldk = ldkv; a.ldv = ldkv; a.ldo = d;
        a.causal = causal; a.scale = 1.0f / sqrtf((float)(d / H));
        a.drop_p = drop_p; a.drop_seed = seed; a.dtype = dt();
        if (!bwd) return pa_attn_fwd(&a, st);
        a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv; a.delta = m->delta;
        a.lddo = d; a.lddq = lddq; a.lddk = lddkv; a.lddv = lddkv;
        return pa_attn_bwd(&a, st);
    }
};

}  // namespace
size_t pa_train_layout(pa_model* m, char* base, int B, int S, int T) {
    const pa_model_cfg& c = m->cfg;
    const size_t e = c.dtype == PA_BF16 ? 2 : 4;
    const size_t d = c.d_model, ff = c.d_ff, H = c.n_head;
    const size_t BS = (size_t)B * S, BT = (size_t)B * T, R = BS > BT ? BS : BT;
    Arena a{base, 0};
    m->esz = e;
    {   // (first, so that it only moves with the buffer itself)
        const int64_t n = c.dtype == PA_BF16 ? pa_attn_ws_bytes(B * S, B, (int32_t)H, S) : 0;
        m->attn_ws = n > 0 ? a.take((size_t)n) : nullptr;
        m->attn_ws_bytes = n > 0 ? n : 0;
    }
    m->X.resize(c.n_enc + 1); m->Y.resize(c.n_dec + 1); m->ea.resize(c.n_enc); m->da.resize(c.n_dec);
    for (int i = 0; i <= c.n_enc; ++i) m->X[i] = a.take(BS * d * e);
    for (int i = 0; i < c.n_enc; ++i) {
        EncAct& t = m->ea[i];
        t.qkv = a.take(BS * 3 * d * e); t.o = a.take(BS * d * e); t.z1 = a.take(BS * d * e); t.y1 = a.take(BS * d * e);
        t.hff = a.take(BS * ff * e); t.z2 = a.take(BS * d * e);
        t.lse = (float*)a.take((size_t)B * H * S * 4);
        t.m1 = (float*)a.take(BS * 4); t.r1 = (float*)a.take(BS * 4); t.m2 = (float*)a.take(BS * 4); t.r2 = (float*)a.take(BS * 4);
    }
    m->memory = a.take(BS * d * e); m->mem_m = (float*)a.take(BS * 4); m->mem_r = (float*)a.take(BS * 4);
    for (int i = 0; i <= c.n_dec; ++i) m->Y[i] = a.take(BT * d * e);
    for (int i = 0; i < c.n_dec; ++i) {
        DecAct& t = m->da[i];
        t.qkv = a.take(BT * 3 * d * e); t.o_sa = a.take(BT * d * e); t.z1 = a.take(BT * d * e); t.y1 = a.take(BT * d * e);
        t.q_ca = a.take(BT * d * e); t.kv_ca = nullptr; t.o_ca = a.take(BT * d * e);
        t.z2 = a.take(BT * d * e); t.y2 = a.take(BT * d * e); t.hff = a.take(BT * ff * e); t.z3 = a.take(BT * d * e);
        t.lse_sa = (float*)a.take((size_t)B * H * T * 4); t.lse_ca = (float*)a.take((size_t)B * H * T * 4);
        t.m1 = (float*)a.take(BT * 4); t.r1 = (float*)a.take(BT * 4); t.m2 = (float*)a.take(BT * 4);
        t.r2 = (float*)a.take(BT * 4); t.m3 = (float*)a.take(BT * 4); t.r3 = (float*)a.take(BT * 4);
    }
    // cross-attention K | V of all decoder layers side by side: layer i's slice starts at column i * 2d (row stride n_dec * 2d)
    m->kv_all = a.take(BS * 2 * d * e * (c.n_dec > 0 ? c.n_dec : 1));
    for (int i = 0; i < c.n_dec; ++i) { m->da[i].kv_ca = (char*)m->kv_all + (size_t)i * 2 * d * e; m->da[i].ld_kv = c.n_dec * 2 * (int)d; }
    m->hid = a.take(BT * d * e); m->hid_m = (float*)a.take(BT * 4); m->hid_r = (float*)a.take(BT * 4);
    m->ldv = (c.vocab + 63) / 64 * 64;     // logits / d(logits) rows padded to a whole K tile: the pad columns are written as zeros
                                           // (mixture_nll_bwd), so dX = d(logits) x W runs as an aligned GEMM with K = ldv
    m->vlog = (float*)a.take(BT * m->ldv * 4); m->pfeat = a.take(BT * d * e);
    m->plog = (float*)a.take(BT * T * 4); m->sw = (float*)a.take(BT * 4); m->row_lse = (float*)a.take(BT * 2 * 4);
    // backward temporaries
    m->gA = a.take(R * d * e); m->gB = a.take(R * d * e); m->gC = a.take(R * d * e); m->gD = a.take(R * d * e);
    // per-site copies of the LayerNorm-backward outputs (dz, dropped dz): the weight-gradient GEMMs that read them are
    // queued and run together at the end of the layer's backward segment, so a later site must not overwrite them
    m->gBs[0] = m->gB; m->gCs[0] = m->gC;
    for (int s_ = 1; s_ < 3; ++s_) { m->gBs[s_] = a.take(R * d * e); m->gCs[s_] = a.take(R * d * e); }
    m->gE = a.take(R * d * e); m->gF = a.take(R * ff * e); m->gQ3 = a.take(R * 3 * d * e);
    m->gPre = c.activation == 2 ? a.take(R * ff * e) : nullptr;
    m->gKV = a.take(BS * 2 * d * e); m->dmem = a.take(BS * d * e);
    m->gKV_all = a.take(BS * 2 * d * e * (c.n_dec > 0 ? c.n_dec : 1));     // d(K|V) of every decoder layer side by side (packed K/V shadow mode)
    {   // parity set 1 (set 0 = the buffers above)
        pa_model::SegSet& s1 = m->seg_set[1];
        for (int s_ = 0; s_ < 3; ++s_) { s1.gBs[s_] = a.take(R * d * e); s1.gCs[s_] = a.take(R * d * e); }
        s1.gE = a.take(R * d * e); s1.gF = a.take(R * ff * e); s1.gQ3 = a.take(R * 3 * d * e); s1.gKV = a.take(BS * 2 * d * e);
    }
    m->dvlog = a.take(BT * m->ldv * e); m->dplog = a.take(BT * T * e); m->dsw = (float*)a.take(BT * 4);
    m->delta = (float*)a.take((size_t)B * H * R * 4);
    size_t part = (size_t)pa_layernorm_ws_floats((int64_t)R, (int)d);
    const size_t wide = 3 * d > ff ? 3 * d : ff;
    const size_t cs = (size_t)pa_colsum_ws_floats((int)R, (int)(wide > (size_t)m->ldv ? wide : m->ldv));
    if (cs > part) part = cs;
    m->partial = (float*)a.take(part * 4);
    for (int s_ = 0; s_ < 3; ++s_) m->lnp[s_] = (float*)a.take((size_t)pa_layernorm_ws_floats((int64_t)R, (int)d) * 4);   // queued LN-backward partials
    for (int s_ = 0; s_ < 3; ++s_) m->seg_set[1].lnp[s_] = (float*)a.take((size_t)pa_layernorm_ws_floats((int64_t)R, (int)d) * 4);
    // split-K slabs of every weight gradient of one backward segment (they are reduced together at its end):
    // a decoder layer is the largest segment (self in/out, cross in/out, two FFN weights), <= 16 slices each
    const size_t wmax = (3 * d * d > d * ff ? 3 * d * d : d * ff);
    const size_t wlayer = 8 * d * d + 2 * d * ff + (size_t)m->ldv * d;
    m->splitws_floats = 16 * (wlayer > wmax ? wlayer : wmax);
    m->splitws = (float*)a.take(m->splitws_floats * 4);
    m->seg_set[1].splitws = (float*)a.take(m->splitws_floats * 4);
    {   // record set 0 and make it current
        pa_model::SegSet& s0 = m->seg_set[0];
        for (int s_ = 0; s_ < 3; ++s_) { s0.gBs[s_] = m->gBs[s_]; s0.gCs[s_] = m->gCs[s_]; s0.lnp[s_] = m->lnp[s_]; }
        s0.gE = m->gE; s0.gF = m->gF; s0.gQ3 = m->gQ3; s0.gKV = m->gKV; s0.splitws = m->splitws;
    }
    return a.off;
}
namespace {

}  // namespace
// --------------------------------------------------------------------------------------------------
int pa_train_forward_impl(pa_model* m, void* st) {
    const pa_model_cfg& c = m->cfg;
    Ctx k{m, st};
    const int d = c.d_model, ff = c.d_ff, B = m->B, S = m->S, T = m->T;
    const int BS = m->NE, BT = B * T;          // BS: encoder rows (packed when the batch carries cu_in)
    const int32_t* cu = m->batch.cu_in;
    const uint8_t* in_mask = cu ? nullptr : m->batch.input_mask;   // packed rows need no padding mask
    const float p = m->p_drop;
    auto PF = [&](int i) { return (const float*)m->pf[i]; };
    auto PL = [&](int i) { return (const void*)m->pl[i]; };
    // ---- encoder (reference models.py:103-112, 206) ----
    {
        const float* tabs[5] = {PF(P_IN_VALUE), PF(P_IN_POS), PF(P_IN_COORD), PF(P_IN_VIEW), PF(P_IN_TYPE)};
        RC(pa_embed_input_fwd(m->X[0], c.dtype, tabs, m->batch.input_idx, m->batch.rowmap, 5, (int64_t)BS, d, st));
    }
    for (int i = 0; i < c.n_enc; ++i) {
        const int pb = m->enc_base(i);
        EncAct& t = m->ea[i];
        const size_t e = m->esz;
        RC(k.linear(m->X[i], d, PL(pb + E_IN_W), PF(pb + E_IN_B), t.qkv, 3 * d, BS, 3 * d, d));
        RC(k.attn(false, t.qkv, 3 * d, (char*)t.qkv + d * e, (char*)t.qkv + 2 * d * e, 3 * d, t.o, t.lse,
                  in_mask, S, S, 0, p, site_seed(m->seed, 8 * i + 0), nullptr, nullptr, 0, nullptr, nullptr, 0, cu, cu));
        RC(k.linear(t.o, d, PL(pb + E_OUT_W), PF(pb + E_OUT_B), t.z1, d, BS, d, d, 0, p, site_seed(m->seed, 8 * i + 1), m->X[i], d));
        RC(k.ln_fwd(t.y1, t.z1, PF(pb + E_N1_W), PF(pb + E_N1_B), t.m1, t.r1, BS, c.eps_layer));
        RC(k.ffn1(t.y1, PL(pb + E_L1_W), PF(pb + E_L1_B), t.hff, BS, p, site_seed(m->seed, 8 * i + 2)));
        RC(k.linear(t.hff, ff, PL(pb + E_L2_W), PF(pb + E_L2_B), t.z2, d, BS, d, ff, 0, p, site_seed(m->seed, 8 * i + 3), t.y1, d));
        RC(k.ln_fwd(m->X[i + 1], t.z2, PF(pb + E_N2_W), PF(pb + E_N2_B), t.m2, t.r2, BS, c.eps_layer));
    }
    const void* memory = m->X[c.n_enc];
    if (c.has_enc_norm) {
        RC(k.ln_fwd(m->memory, m->X[c.n_enc], PF(m->enc_norm()), PF(m->enc_norm() + 1), m->mem_m, m->mem_r, BS, c.eps_final));
        memory = m->memory;
    }
    if (!m->batch.output_value) return 0;      // encoder-only call (greedy decode prologue)
    // ---- decoder (reference models.py:114-138, 204, 209-214) ----
    RC(pa_embed_output_fwd(m->Y[0], c.dtype, PF(P_IN_VALUE), PF(P_Q_COORD), PF(P_Q_POS), m->batch.output_value, T, B, T, d,
                           c.out_dof, st));
    // Cross-attention K | V projections of ALL decoder layers as one batched GEMM over the memory (they depend on nothing
    // the decoder computes): the layers' in_proj weights / biases sit at a constant stride in the flat parameter buffers, so
    // batch member i reads weight rows d..3d of layer i and writes columns i*2d.. of kv_all.  One launch with n_dec x the tiles
    // instead of n_dec launches in the decoder's latency-bound chain.
    bool kv_fused = false;
    if (c.n_dec > 1) {
        const size_t e = m->esz;
        const char* w0 = (const char*)PL(m->dec_base(0) + D_CA_IN_W); const char* w1 = (const char*)PL(m->dec_base(1) + D_CA_IN_W);
        const float* b0 = PF(m->dec_base(0) + D_CA_IN_B); const float* b1 = PF(m->dec_base(1) + D_CA_IN_B);
        const ptrdiff_t ws = w1 - w0, bs = (const char*)b1 - (const char*)b0;
        bool uniform = ws > 0 && bs > 0 && ws % (ptrdiff_t)e == 0 && bs % 16 == 0;
        for (int i = 2; i < c.n_dec && uniform; ++i)
            uniform = (const char*)PL(m->dec_base(i) + D_CA_IN_W) - w0 == ws * i && (const char*)PF(m->dec_base(i) + D_CA_IN_B) - (const char*)b0 == bs * i;
        static const bool fuse_env = !(getenv("PLANK_CROSS_KV_FWD") && atoi(getenv("PLANK_CROSS_KV_FWD")) == 0);
        if (uniform && fuse_env) {
            pa_gemm_args g; memset(&g, 0, sizeof(g));
            g.A = memory; g.B = w0 + (size_t)d * d * e; g.C = m->kv_all; g.bias = b0 + d;
            g.M = BS; g.N = 2 * d; g.K = d; g.lda = d; g.ldb = d; g.ldc = c.n_dec * 2 * d;
            g.batch = c.n_dec; g.sA = 0; g.sB = ws / (ptrdiff_t)e; g.sC = 2 * d; g.sBias = bs / 4;
            g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = c.dtype; g.out_dtype = c.dtype;
            g.alpha = 1.f; g.aux_scale = 1.f; g.splitk = 1;
            RC(pa_gemm(&g, st));
            kv_fused = true;
        }
    }
    for (int i = 0; i < c.n_dec; ++i) {
        const int pb = m->dec_base(i);
        DecAct& t = m->da[i];
        const size_t e = m->esz;
        const uint32_t sb = 1000 + 8 * i;
        RC(k.linear(m->Y[i], d, PL(pb + D_SA_IN_W), PF(pb + D_SA_IN_B), t.qkv, 3 * d, BT, 3 * d, d));
        RC(k.attn(false, t.qkv, 3 * d, (char*)t.qkv + d * e, (char*)t.qkv + 2 * d * e, 3 * d, t.o_sa, t.lse_sa,
                  m->batch.output_mask, T, T, 1, p, site_seed(m->seed, sb + 0)));
        RC(k.linear_ln(t.o_sa, d, PL(pb + D_SA_OUT_W), PF(pb + D_SA_OUT_B), m->Y[i], t.z1, t.y1, PF(pb + D_N1_W), PF(pb + D_N1_B),
                       t.m1, t.r1, BT, d, p, site_seed(m->seed, sb + 1), c.eps_layer));
        // cross attention: q from y1 (rows 0..d of in_proj), k/v from memory (rows d..3d)
        RC(k.linear(t.y1, d, PL(pb + D_CA_IN_W), PF(pb + D_CA_IN_B), t.q_ca, d, BT, d, d));
        if (!kv_fused)
            RC(k.linear(memory, d, (const char*)PL(pb + D_CA_IN_W) + (size_t)d * d * e, PF(pb + D_CA_IN_B) + d, t.kv_ca, t.ld_kv, BS, 2 * d, d));
        RC(k.attn(false, t.q_ca, d, t.kv_ca, (char*)t.kv_ca + d * e, t.ld_kv, t.o_ca, t.lse_ca, in_mask, T, S, 0, p,
                  site_seed(m->seed, sb + 2), nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr, cu));
        RC(k.linear_ln(t.o_ca, d, PL(pb + D_CA_OUT_W), PF(pb + D_CA_OUT_B), t.y1, t.z2, t.y2, PF(pb + D_N2_W), PF(pb + D_N2_B),
                       t.m2, t.r2, BT, d, p, site_seed(m->seed, sb + 3), c.eps_layer));
        RC(k.ffn1(t.y2, PL(pb + D_L1_W), PF(pb + D_L1_B), t.hff, BT, p, site_seed(m->seed, sb + 4)));
        RC(k.linear_ln(t.hff, ff, PL(pb + D_L2_W), PF(pb + D_L2_B), t.y2, t.z3, m->Y[i + 1], PF(pb + D_N3_W), PF(pb + D_N3_B),
                       t.m3, t.r3, BT, ff, p, site_seed(m->seed, sb + 5), c.eps_layer));
    }
    RC(k.ln_fwd(m->hid, m->Y[c.n_dec], PF(m->dec_norm()), PF(m->dec_norm() + 1), m->hid_m, m->hid_r, BT, c.eps_final));
    // ---- heads + mixture NLL (reference models.py:140-166, 219-227) ----
    const int tl = m->tail();
    RC(k.linear(m->hid, d, PL(tl + T_VOCAB_W), PF(tl + T_VOCAB_B), m->vlog, m->ldv, BT, c.vocab, d, 0, 0.f, 0, nullptr, 0, PA_F32));
    RC(k.linear(m->hid, d, PL(tl + T_PTR_W), PF(tl + T_PTR_B), m->pfeat, d, BT, d, d));
    {
        pa_gemm_args g; memset(&g, 0, sizeof(g));
        g.A = m->pfeat; g.B = m->hid; g.C = m->plog;
        g.M = T; g.N = T; g.K = d; g.lda = d; g.ldb = d; g.ldc = T;
        g.sA = (int64_t)T * d; g.sB = (int64_t)T * d; g.sC = (int64_t)T * T; g.batch = B;
        g.a_kcontig = 1; g.b_kcontig = 1; g.in_dtype = c.dtype; g.out_dtype = PA_F32;
        g.alpha = 1.0f / (float)d; g.aux_scale = 1.f; g.splitk = 1;
        RC(pa_gemm(&g, st));
    }
    RC(pa_switch_fwd(m->sw, m->hid, c.dtype, PF(tl + T_SW_W), PF(tl + T_SW_B), (int64_t)BT, d, st));
    // (one memset inside; loss, accuracy and the upstream 1.0 are written by the kernel's last block: stats is 8 floats)
    m->upstream = nullptr;
    RC(pa_mixture_nll_fwd_fin(m->stats, m->row_lse, m->vlog, m->ldv, m->plog, m->sw, m->batch.output_label, B, T, c.vocab, c.pad, st));
    return 0;
}
namespace {

// --------------------------------------------------------------------------------------------------
// backward segments (execution order): 0 heads+decoder.norm | 1..n_dec decoder layers (last first)
// | n_dec+1 output embedding | n_dec+2 encoder.norm | then encoder layers (last first) | input embedding
int bwd_heads(pa_model* m, float gscale, void* st) {
    const pa_model_cfg& c = m->cfg;
    Ctx k{m, st};
    const int d = c.d_model, B = m->B, T = m->T, BT = B * T, tl = m->tail();
    auto G = [&](int i) { return (float*)m->gr[i]; };
    RC(pa_mixture_nll_bwd_up(m->dvlog, m->dplog, c.dtype, m->dsw, m->stats, m->row_lse, m->vlog, m->ldv, m->plog, m->sw,
                             m->batch.output_label, B, T, c.vocab, c.pad, gscale, m->upstream, st));
    // vocab head
    if (m->plT[tl + T_VOCAB_W])       // padded W^T shadow [d][ldv] (pad columns zero): contraction over the padded width, both k-contiguous
        RC(k.linear_dx(m->dvlog, m->ldv, m->pl[tl + T_VOCAB_W], d, m->gA, d, BT, m->ldv, d, nullptr, 0, nullptr, 0, 1.f,
                       m->plT[tl + T_VOCAB_W], m->ldv));
    else
        RC(k.linear_dx(m->dvlog, m->ldv, m->pl[tl + T_VOCAB_W], d, m->gA, d, BT, c.vocab, d));
    RC(k.linear_dw(m->dvlog, m->ldv, m->hid, d, G(tl + T_VOCAB_W), G(tl + T_VOCAB_B), BT, c.vocab, d));
    // pointer head: plog[b] = pfeat[b] hid[b]^T / d
    {
        pa_gemm_args g; memset(&g, 0, sizeof(g));
        g.A = m->dplog; g.B = m->hid; g.C = m->gB;                 // dpfeat = dplog x hid / d
        g.M = T; g.N = d; g.K = T; g.lda = T; g.ldb = d; g.ldc = d;
        g.sA = (int64_t)T * T; g.sB = (int64_t)T * d; g.sC = (int64_t)T * d; g.batch = B;
        g.a_kcontig = 1; g.b_kcontig = 0; g.in_dtype = c.dtype; g.out_dtype = c.dtype;
        g.alpha = 1.0f / (float)d; g.aux_scale = 1.f; g.splitk = 1;
        RC(pa_gemm(&g, st));
        g.A = m->dplog; g.B = m->pfeat; g.C = m->gA; g.R = m->gA;   // dhid += dplog^T x pfeat / d
        g.a_kcontig = 0; g.ldr = d; g.sR = (int64_t)T * d;
        RC(pa_gemm(&g, st));
    }
    RC(k.linear_dx(m->gB, d, m->pl[tl + T_PTR_W], d, m->gA, d, BT, d, d, m->gA, d, nullptr, 0, 1.f, m->plT[tl + T_PTR_W], d));
    RC(k.linear_dw(m->gB, d, m->hid, d, G(tl + T_PTR_W), G(tl + T_PTR_B), BT, d, d));
    RC(pa_switch_bwd(m->gA, 1, G(tl + T_SW_W), G(tl + T_SW_B), m->dsw, m->hid, c.dtype, (const float*)m->pf[tl + T_SW_W],
                     m->partial, (int64_t)BT, d, st));
    // decoder.norm
    const int dn = m->dec_norm();
    // (dz may alias dy: a wave holds its whole row in registers before it stores)
    return k.ln_bwd(m->gA, nullptr, m->gA, m->Y[c.n_dec], (const float*)m->pf[dn], m->hid_m, m->hid_r, G(dn), G(dn + 1), nullptr,
                    BT, 0.f, 0);
}

// FFN block backward shared by encoder/decoder layers.  In: gA = d(layer output).  Out: gA = d(FFN input y).
int bwd_ffn(pa_model* m, Ctx& k, int rows, const void* z, const float* mean, const float* rstd, const void* hff,
            const void* yin, int w1, int w2, int nw, uint32_t seed_inner, uint32_t seed_out) {
    void* const gB = m->gBs[0]; void* const gC = m->gCs[0];
    const pa_model_cfg& c = m->cfg;
    const int d = c.d_model, ff = c.d_ff;
    const float p = m->p_drop;
    auto G = [&](int i) { return (float*)m->gr[i]; };
    void* ddrop = p > 0.f ? gC : gB;
    RC(k.ln_bwd(gB, p > 0.f ? gC : nullptr, m->gA, z, (const float*)m->pf[nw], mean, rstd, G(nw), G(nw + 1),
                G(w2 + 1), rows, p, seed_out));
    RC(k.linear_dw(ddrop, d, hff, ff, G(w2), nullptr, rows, d, ff));
    if (c.activation == 2) {
        // GELU: d pre = (dY W2) * gelu'(pre) * keep / (1 - p).  The forward keeps only drop(gelu(pre)) (linear2's operand), so the
        // pre-activation y W1^T + b1 is recomputed here (one more Linear per FFN in this mode; ReLU reads its gate off the saved hidden
        // rows), and pa_gelu_bwd re-applies the FFN1 dropout decisions - a pure function of (site seed, row, column).
        RC(k.linear(yin, d, m->pl[w1], (const float*)m->pf[w1 + 1], m->gPre, ff, rows, ff, d, 0));
        RC(k.linear_dx(ddrop, d, m->pl[w2], ff, m->gF, ff, rows, d, ff, nullptr, 0, nullptr, 0, 1.f, m->plT[w2], d));
        RC(pa_gelu_bwd(m->gF, m->gF, m->gPre, rows, ff, ff, c.dtype, p, seed_inner, k.st));
    } else {
        RC(k.linear_dx(ddrop, d, m->pl[w2], ff, m->gF, ff, rows, d, ff, nullptr, 0, hff, ff, 1.0f / (1.0f - p), m->plT[w2], d));
    }
    RC(k.linear_dw(m->gF, ff, yin, d, G(w1), G(w1 + 1), rows, ff, d));
    RC(k.linear_dx(m->gF, ff, m->pl[w1], d, m->gA, d, rows, ff, d, gB, d, nullptr, 0, 1.f, m->plT[w1], ff));
    return 0;
}

int bwd_dec_layer(pa_model* m, int i, void* st) {
    const pa_model_cfg& c = m->cfg;
    Ctx k{m, st};
    const int d = c.d_model, B = m->B, S = m->S, T = m->T, BS = m->NE, BT = B * T;
    const int32_t* cu = m->batch.cu_in;
    const uint8_t* in_mask = cu ? nullptr : m->batch.input_mask;
    const float p = m->p_drop;
    const size_t e = m->esz;
    const int pb = m->dec_base(i);
    DecAct& t = m->da[i];
    const uint32_t sb = 1000 + 8 * i;
    auto G = [&](int j) { return (float*)m->gr[j]; };
    const void* memory = c.has_enc_norm ? m->memory : m->X[c.n_enc];
    RC(bwd_ffn(m, k, BT, t.z3, t.m3, t.r3, t.hff, t.y2, pb + D_L1_W, pb + D_L2_W, pb + D_N3_W,
               site_seed(m->seed, sb + 4), site_seed(m->seed, sb + 5)));
    // cross attention block: z2 = y1 + drop(out_proj(attn(q(y1), kv(memory))))
    void* gB = m->gBs[1]; void* gC = m->gCs[1];
    void* ddrop = p > 0.f ? gC : gB;
    RC(k.ln_bwd(gB, p > 0.f ? gC : nullptr, m->gA, t.z2, (const float*)m->pf[pb + D_N2_W], t.m2, t.r2, G(pb + D_N2_W),
                G(pb + D_N2_B), G(pb + D_CA_OUT_B), BT, p, site_seed(m->seed, sb + 3)));
    RC(k.linear_dw(ddrop, d, t.o_ca, d, G(pb + D_CA_OUT_W), nullptr, BT, d, d));
    RC(k.linear_dx(ddrop, d, m->pl[pb + D_CA_OUT_W], d, m->gD, d, BT, d, d, nullptr, 0, nullptr, 0, 1.f, m->plT[pb + D_CA_OUT_W], d));
    // With the packed K/V weight shadow bound (bf16 mode) every layer writes d(K|V) into its column slice of one
    // [BS][n_dec * 2d] matrix and d(memory) is ONE GEMM over all layers after the last decoder segment
    // (backward_segment_body) instead of n_dec accumulating launches on the dX chain.
    const bool kv_all = m->kvT_all != nullptr;
    void* gKV = kv_all ? (void*)((char*)m->gKV_all + (size_t)i * 2 * d * e) : m->gKV;
    const int ldg = kv_all ? c.n_dec * 2 * d : 2 * d;
    RC(k.attn(true, t.q_ca, d, t.kv_ca, (char*)t.kv_ca + d * e, t.ld_kv, t.o_ca, t.lse_ca, in_mask, T, S, 0, p,
              site_seed(m->seed, sb + 2), m->gD, m->gE, d, gKV, (char*)gKV + d * e, ldg, nullptr, cu));
    float* dWin = G(pb + D_CA_IN_W); float* dbin = G(pb + D_CA_IN_B);
    RC(k.linear_dw(m->gE, d, t.y1, d, dWin, dbin, BT, d, d));
    RC(k.linear_dw(gKV, ldg, memory, d, dWin + (size_t)d * d, dbin + d, BS, 2 * d, d));
    if (!kv_all) {
        const void* wt = m->plT[pb + D_CA_IN_W];                       // W_in^T is [d][3d]; K/V rows of W_in = its columns d..3d
        RC(k.linear_dx(gKV, 2 * d, (const char*)m->pl[pb + D_CA_IN_W] + (size_t)d * d * e, d, m->dmem, d, BS, 2 * d, d,
                       m->dmem_written ? m->dmem : nullptr, d, nullptr, 0, 1.f, wt ? (const char*)wt + (size_t)d * e : nullptr, 3 * d));
        m->dmem_written = true;
    }
    RC(k.linear_dx(m->gE, d, m->pl[pb + D_CA_IN_W], d, m->gA, d, BT, d, d, gB, d, nullptr, 0, 1.f, m->plT[pb + D_CA_IN_W], 3 * d));
    // self attention block: z1 = Y[i] + drop(out_proj(attn(qkv(Y[i]))))
    gB = m->gBs[2]; gC = m->gCs[2]; ddrop = p > 0.f ? gC : gB;
    RC(k.ln_bwd(gB, p > 0.f ? gC : nullptr, m->gA, t.z1, (const float*)m->pf[pb + D_N1_W], t.m1, t.r1, G(pb + D_N1_W),
                G(pb + D_N1_B), G(pb + D_SA_OUT_B), BT, p, site_seed(m->seed, sb + 1)));
    RC(k.linear_dw(ddrop, d, t.o_sa, d, G(pb + D_SA_OUT_W), nullptr, BT, d, d));
    RC(k.linear_dx(ddrop, d, m->pl[pb + D_SA_OUT_W], d, m->gD, d, BT, d, d, nullptr, 0, nullptr, 0, 1.f, m->plT[pb + D_SA_OUT_W], d));
    RC(k.attn(true, t.qkv, 3 * d, (char*)t.qkv + d * e, (char*)t.qkv + 2 * d * e, 3 * d, t.o_sa, t.lse_sa,
              m->batch.output_mask, T, T, 1, p, site_seed(m->seed, sb + 0), m->gD, m->gQ3, 3 * d,
              (char*)m->gQ3 + d * e, (char*)m->gQ3 + 2 * d * e, 3 * d));
    RC(k.linear_dw(m->gQ3, 3 * d, m->Y[i], d, G(pb + D_SA_IN_W), G(pb + D_SA_IN_B), BT, 3 * d, d));
    RC(k.linear_dx(m->gQ3, 3 * d, m->pl[pb + D_SA_IN_W], d, m->gA, d, BT, 3 * d, d, gB, d, nullptr, 0, 1.f, m->plT[pb + D_SA_IN_W], 3 * d));
    return 0;
}

int bwd_enc_layer(pa_model* m, int i, void* st) {
    const pa_model_cfg& c = m->cfg;
    Ctx k{m, st};
    const int d = c.d_model, S = m->S, BS = m->NE;
    const int32_t* cu = m->batch.cu_in;
    const uint8_t* in_mask = cu ? nullptr : m->batch.input_mask;
    const float p = m->p_drop;
    const size_t e = m->esz;
    const int pb = m->enc_base(i);
    EncAct& t = m->ea[i];
    auto G = [&](int j) { return (float*)m->gr[j]; };
    RC(bwd_ffn(m, k, BS, t.z2, t.m2, t.r2, t.hff, t.y1, pb + E_L1_W, pb + E_L2_W, pb + E_N2_W,
               site_seed(m->seed, 8 * i + 2), site_seed(m->seed, 8 * i + 3)));
    void* const gB = m->gBs[2]; void* const gC = m->gCs[2];
    void* ddrop = p > 0.f ? gC : gB;
    RC(k.ln_bwd(gB, p > 0.f ? gC : nullptr, m->gA, t.z1, (const float*)m->pf[pb + E_N1_W], t.m1, t.r1, G(pb + E_N1_W),
                G(pb + E_N1_B), G(pb + E_OUT_B), BS, p, site_seed(m->seed, 8 * i + 1)));
    RC(k.linear_dw(ddrop, d, t.o, d, G(pb + E_OUT_W), nullptr, BS, d, d));
    RC(k.linear_dx(ddrop, d, m->pl[pb + E_OUT_W], d, m->gD, d, BS, d, d, nullptr, 0, nullptr, 0, 1.f, m->plT[pb + E_OUT_W], d));
    RC(k.attn(true, t.qkv, 3 * d, (char*)t.qkv + d * e, (char*)t.qkv + 2 * d * e, 3 * d, t.o, t.lse, in_mask, S, S, 0,
              p, site_seed(m->seed, 8 * i + 0), m->gD, m->gQ3, 3 * d, (char*)m->gQ3 + d * e, (char*)m->gQ3 + 2 * d * e, 3 * d, cu, cu));
    RC(k.linear_dw(m->gQ3, 3 * d, m->X[i], d, G(pb + E_IN_W), G(pb + E_IN_B), BS, 3 * d, d));
    RC(k.linear_dx(m->gQ3, 3 * d, m->pl[pb + E_IN_W], d, m->gA, d, BS, 3 * d, d, gB, d, nullptr, 0, 1.f, m->plT[pb + E_IN_W], 3 * d));
    return 0;
}

int backward_segment_body(pa_model* m, int seg, float gscale, void* st);
#define HC(x) do { hipError_t he_ = (x); if (he_ != hipSuccess) return (int)he_; } while (0)
// enqueue the segment's queued work (grouped weight-gradient GEMM, split-K reductions, bias column sums, LayerNorm
// finishes) on stream `q`
// Split-K plan for the queued weight-gradient GEMMs of a segment (members queued with splitk == 0).  The grouped launch
// runs one block per CU and a unit is (128x128 tile, K slice), so its duration is the longest K slice: starting from no
// split, the member with the longest slice gets one more slice for as long as the launch still fits one round of the CUs.
// Encoder layer (K = 7 940 rows, 128 tiles): every member split in two -> 256 units of 63 K tiles; decoder layer: only the
// cross-attention K/V gradient (K = encoder rows) is split.  Slabs of the split members are reduced by the segment tail.
void plan_group(pa_model* m) {
    const int n = m->ndwq;
    int tiles[PA_MAX_GROUP], nkt[PA_MAX_GROUP], cap[PA_MAX_GROUP], sk[PA_MAX_GROUP];
    const bool x3 = m->cfg.dtype == PA_F32 && pa_gemm_split_active();     // bf16x3: planned like the bf16 grouped launch, 3 K rows each
    const int ktile = (m->cfg.dtype == PA_BF16 || x3) ? 64 : 16;
    int total = 0, planned = 0;
    for (int i = 0; i < n; ++i) {
        const pa_gemm_args& g = m->dwq[i];
        tiles[i] = ((g.M + 127) / 128) * ((g.N + 127) / 128);
        nkt[i] = ((x3 ? 3 * g.K : g.K) + ktile - 1) / ktile;
        sk[i] = g.splitk > 0 ? g.splitk : 1;
        cap[i] = g.splitk > 0 ? sk[i] : ((g.N & 3) ? 1 : (nkt[i] / 4 < 16 ? (nkt[i] / 4 > 0 ? nkt[i] / 4 : 1) : 16));
        total += tiles[i] * sk[i];
        planned += g.splitk == 0;
    }
    if (!planned) return;
    if (m->cfg.dtype != PA_BF16 && !x3) {
        // f32 (the parity path): pa_gemm_group takes bf16 members only, so flush_segment launches these one by one on the
        // two-blocks-per-CU kernel - each member then wants the whole chip for itself (512 units), not its share of one grouped
        // round.  Rounds 1-3 planned them like a grouped launch: the encoder's out_proj gradient ran on 32 of the 512 block
        // slots, in_proj on 96, and the weight gradients were HALF of the f32 step (30 of 60 ms, 21 TFLOP/s = 0.13 of the f32
        // MFMA peak; profiles/r04_train_kernel_trace_summary.txt before / after).
        static const int lone_env = getenv("PA_DW_UNITS") ? atoi(getenv("PA_DW_UNITS")) : 0;
        const int want = lone_env > 0 ? lone_env : 512;
        for (int i = 0; i < n; ++i)
            if (m->dwq[i].splitk == 0) { const int s = want / tiles[i]; sk[i] = s < 1 ? 1 : (s > cap[i] ? cap[i] : s); }
    } else {
    static const int budget_env = getenv("PA_DW_BUDGET") ? atoi(getenv("PA_DW_BUDGET")) : 0;
    // whole rounds of the 256 CUs: one round for the benchmark model, more when the unsplit tiles already exceed it
    const int budget = budget_env > 0 ? budget_env : (total + 255) / 256 * 256;
    for (;;) {
        int best = -1, len = 0;
        for (int i = 0; i < n; ++i) {
            const int li = (nkt[i] + sk[i] - 1) / sk[i];
            if (li > len) { len = li; best = i; }
        }
        if (best < 0 || sk[best] >= cap[best] || total + tiles[best] > budget) break;
        ++sk[best]; total += tiles[best];
    }
    }
    for (int i = 0; i < n; ++i) {
        pa_gemm_args& g = m->dwq[i];
        if (g.splitk > 0) continue;
        int s = pa_gemm_effective_splitk(g.K, g.in_dtype, sk[i]);
        const size_t need = (size_t)s * g.M * g.N;
        if (s > 1 && !(m->ndefer < PA_MAX_REDUCE && m->slab_used + need <= m->splitws_floats)) s = 1;
        g.splitk = s; g.ws = m->splitws; g.splitk_defer = 0;
        if (s > 1) {
            g.ws = m->splitws + m->slab_used;
            g.splitk_defer = 1;
            pa_reduce_desc& rd = m->defer[m->ndefer++];
            rd.ws = (const float*)g.ws; rd.out = (float*)g.C; rd.rows = g.M; rd.cols = g.N; rd.ld_out = g.ldc; rd.splitk = s;
            m->slab_used += (need + 3) / 4 * 4;
        }
    }
}
int flush_segment(pa_model* m, void* q) {
    if (m->ndwq > 0) {                                     // all weight-gradient GEMMs of the segment: one ring-kernel launch
        plan_group(m);
        int rc = m->ndwq > 1 ? pa_gemm_group(m->dwq, m->ndwq, q) : PA_EINVAL;
        if (rc == PA_EINVAL) { rc = 0; for (int i = 0; i < m->ndwq && !rc; ++i) rc = pa_gemm(&m->dwq[i], q); }
        m->ndwq = 0;
        RC(rc);
    }
    // LayerNorm gamma/beta finishes + bias column sums + split-K slab reductions of the segment: one launch
    if (m->nlnq > 0 || m->ncs > 0 || m->ndefer > 0) {
        RC(pa_segment_tail(m->lnq, m->nlnq, m->cfg.d_model, m->cs, m->ncs, m->cfg.dtype, m->defer, m->ndefer, q));
        m->nlnq = 0; m->ncs = 0; m->ndefer = 0; m->slab_used = 0;
    }
    return 0;
}
// main stream waits for the side-stream work of parity `par` (if any is outstanding)
int join_side(pa_model* m, int par, void* st) {
    if (m->ev_pending[par]) {
        HC(hipStreamWaitEvent((hipStream_t)st, (hipEvent_t)m->ev_done[par], 0));
        m->ev_pending[par] = false;
    }
    return 0;
}
int backward_segment(pa_model* m, int seg, float gscale, void* st) {
    const int par = seg & 1;
    m->ndefer = 0; m->slab_used = 0; m->ncs = 0; m->ndwq = 0; m->nlnq = 0;
    // heads and layer segments: every buffer the queued work reads (dY of the weight / bias gradients, LayerNorm partials)
    // is written once per segment and stays untouched until its end
    m->defer_ok = seg != m->cfg.n_dec + 1 && seg != m->cfg.n_dec + 2;
    if (m->side_on) {
        RC(join_side(m, par, st));             // segment seg-2 used this parity set: its queued work must be finished
        m->select_set(par);
    }
    RC(backward_segment_body(m, seg, gscale, st));
    m->defer_ok = false;
    const bool queued = m->ndwq > 0 || m->nlnq > 0 || m->ncs > 0 || m->ndefer > 0;
    // (bf16x3 mode: the queued GEMMs cut their operands into the process-wide split scratch, which the next segment's GEMMs on
    // the main stream reuse - the queued work stays on the main stream)
    if (queued && m->side_on && !pa_gemm_split_active()) {
        HC(hipEventRecord((hipEvent_t)m->ev_ready[par], (hipStream_t)st));
        HC(hipStreamWaitEvent((hipStream_t)m->side, (hipEvent_t)m->ev_ready[par], 0));
        RC(flush_segment(m, m->side));
        HC(hipEventRecord((hipEvent_t)m->ev_done[par], (hipStream_t)m->side));
        m->ev_pending[par] = true;
    } else if (queued) {
        RC(flush_segment(m, st));
    }
    return 0;
}
int backward_segment_body(pa_model* m, int seg, float gscale, void* st) {
    const pa_model_cfg& c = m->cfg;
    Ctx k{m, st};
    const int d = c.d_model, B = m->B, T = m->T, BS = m->NE;
    auto G = [&](int j) { return (float*)m->gr[j]; };
    if (seg == 0) { m->dmem_written = false; return bwd_heads(m, gscale, st); }
    if (seg <= c.n_dec) return bwd_dec_layer(m, c.n_dec - seg, st);
    if (seg == c.n_dec + 1)
    {
        const pa_batch& bt = m->batch;
        if (bt.out_order[0] && bt.out_seg[0] && bt.out_order[1] && bt.out_seg[1] && bt.out_order[2] && bt.out_seg[2]) {
            float* gt[3] = {G(P_IN_VALUE), G(P_Q_COORD), G(P_Q_POS)};
            const int32_t* go[3] = {bt.out_order[0], bt.out_order[1], bt.out_order[2]};
            const int32_t* gs[3] = {bt.out_seg[0], bt.out_seg[1], bt.out_seg[2]};
            int32_t gr[3] = {c.in_table_rows[0], c.out_dof, (T + c.out_dof - 1) / c.out_dof};
            return pa_embed_segment_bwd(m->gA, c.dtype, gt, go, gs, gr, 3, (int64_t)B * T, d, st);
        }
        return pa_embed_output_bwd(m->gA, c.dtype, G(P_IN_VALUE), G(P_Q_COORD), G(P_Q_POS), m->batch.output_value, T, B, T, d,
                                   c.out_dof, st);
    }
    if (seg == c.n_dec + 2) {
        if (m->kvT_all && c.n_dec > 0) {
            // d(memory) [BS][d] = d(K|V)_all [BS][n_dec*2d] x W_kv_all [n_dec*2d][d]; the packed shadow holds W_kv_all^T
            const int kk = c.n_dec * 2 * d;
            RC(k.linear_dx(m->gKV_all, kk, nullptr, 0, m->dmem, d, BS, kk, d, nullptr, 0, nullptr, 0, 1.f, m->kvT_all, kk));
            m->dmem_written = true;
        }
        if (!m->dmem_written) {       // no decoder layers: memory got no gradient
            hipError_t he = hipMemsetAsync(m->dmem, 0, (size_t)BS * d * m->esz, (hipStream_t)st);
            if (he != hipSuccess) return (int)he;
        }
        if (c.has_enc_norm) {
            const int en = m->enc_norm();
            return k.ln_bwd(m->gA, nullptr, m->dmem, m->X[c.n_enc], (const float*)m->pf[en], m->mem_m, m->mem_r, G(en), G(en + 1),
                            nullptr, BS, 0.f, 0);
        }
        hipError_t he = hipMemcpyAsync(m->gA, m->dmem, (size_t)BS * d * m->esz, hipMemcpyDeviceToDevice, (hipStream_t)st);
        return he == hipSuccess ? 0 : (int)he;
    }
    const int es = seg - (c.n_dec + 3);
    if (es < c.n_enc) return bwd_enc_layer(m, c.n_enc - 1 - es, st);
    if (es == c.n_enc) {
        float* dt[5] = {G(P_IN_VALUE), G(P_IN_POS), G(P_IN_COORD), G(P_IN_VIEW), G(P_IN_TYPE)};
        const pa_batch& bt = m->batch;
        {   // tables with a per-batch grouping: segment sums; the rest: scatter-add kernel
            float* gt[5]; const int32_t* go[5]; const int32_t* gs[5]; int32_t gr[5]; int ng = 0;
            const int64_t* idx2[5];
            bool any = false;
            for (int j = 0; j < 5; ++j) {
                idx2[j] = bt.input_idx[j];
                if (bt.input_idx[j] && bt.in_order[j] && bt.in_seg[j]) {
                    gt[ng] = dt[j]; go[ng] = bt.in_order[j]; gs[ng] = bt.in_seg[j]; gr[ng] = c.in_table_rows[j]; ++ng;
                    idx2[j] = nullptr;
                }
                any = any || idx2[j] != nullptr;
            }
            if (ng > 0) {
                RC(pa_embed_segment_bwd(m->gA, c.dtype, gt, go, gs, gr, ng, (int64_t)BS, d, st));
                if (!any) return 0;
                return pa_embed_input_bwd(m->gA, c.dtype, dt, idx2, bt.rowmap, c.in_table_rows, 5, (int64_t)BS, d, st);
            }
        }
        return pa_embed_input_bwd(m->gA, c.dtype, dt, m->batch.input_idx, m->batch.rowmap, c.in_table_rows, 5, (int64_t)BS, d, st);
    }
    return PA_EINVAL;
}

}  // namespace

extern "C" int pa_model_create(const pa_model_cfg* cfg, pa_model** out) {
    if (!cfg || !out) return PA_EINVAL;
    if (cfg->d_model <= 0 || cfg->n_head <= 0 || cfg->d_model % cfg->n_head) return PA_EINVAL;
    const int dh = cfg->d_model / cfg->n_head;
    if (dh != 16 && dh != 32 && dh != 64) return PA_ESHAPE;
    if (cfg->d_model % 8 || cfg->d_ff % 8) return PA_ESHAPE;
    if (cfg->dtype != PA_F32 && cfg->dtype != PA_BF16) return PA_EINVAL;
    if (cfg->dropout < 0.f || cfg->dropout >= 1.f) return PA_EINVAL;
    if (cfg->activation < 0 || cfg->activation > 2) return PA_EINVAL;
    pa_model* m = new (std::nothrow) pa_model();
    if (!m) return PA_EINVAL;
    m->cfg = *cfg;
    m->n_params = P_FIXED_HEAD + cfg->n_enc * E_COUNT + 2 + cfg->n_dec * D_COUNT + 2 + T_COUNT;
    m->pf.assign(m->n_params, nullptr); m->pl.assign(m->n_params, nullptr); m->gr.assign(m->n_params, nullptr);
    m->plT.assign(m->n_params, nullptr);
    // Optional side stream for the queued end-of-segment backward work (PA_SIDE_STREAM=1).  Off by default: measured
    // on MI355X the event fences cost more than the overlap gains (7.68 vs 7.49 ms/step, profiles/README.md).
    const char* ss = getenv("PA_SIDE_STREAM");
    if (ss && atoi(ss) != 0) {
        hipStream_t q = nullptr;
        bool ok = hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4 && ok; ++i) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (ok) {
            m->side = q; m->ev_ready[0] = ev[0]; m->ev_ready[1] = ev[1]; m->ev_done[0] = ev[2]; m->ev_done[1] = ev[3];
            m->side_on = true;
        } else {
            for (int i = 0; i < 4; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (q) (void)hipStreamDestroy(q);
            (void)hipGetLastError();
        }
    }
    *out = m;
    return 0;
}

void pa_decode_free_layout(pa_model* m);   // decode.hip
extern "C" void pa_model_destroy(pa_model* m) {
    if (!m) return;
    for (int i = 0; i < 2; ++i) {
        if (m->ev_ready[i]) (void)hipEventDestroy((hipEvent_t)m->ev_ready[i]);
        if (m->ev_done[i]) (void)hipEventDestroy((hipEvent_t)m->ev_done[i]);
    }
    if (m->side) (void)hipStreamDestroy((hipStream_t)m->side);
    pa_decode_free_layout(m);
    delete m;
}

extern "C" int pa_model_num_params(const pa_model* m) { return m ? m->n_params : PA_EINVAL; }

extern "C" int pa_model_bind(pa_model* m, void* const* params_f32, void* const* params_lp, void* const* grads) {
    if (!m || !params_f32 || !params_lp) return PA_EINVAL;
    for (int i = 0; i < m->n_params; ++i) {
        m->pf[i] = params_f32[i]; m->pl[i] = params_lp[i]; m->gr[i] = grads ? grads[i] : nullptr;
    }
    m->bound = true;
    return 0;
}

extern "C" int pa_model_bind_transposed(pa_model* m, void* const* params_lpT) {
    if (!m) return PA_EINVAL;
    for (int i = 0; i < m->n_params; ++i) m->plT[i] = params_lpT ? params_lpT[i] : nullptr;
    return 0;
}

extern "C" int pa_model_bind_cross_kv_t(pa_model* m, const void* kvT_all) {
    if (!m) return PA_EINVAL;
    m->kvT_all = kvT_all;
    return 0;
}

extern "C" int64_t pa_model_train_ws_bytes(pa_model* m, int32_t B, int32_t S, int32_t T) {
    if (!m || B <= 0 || S <= 0 || T <= 0) return PA_EINVAL;
    return (int64_t)pa_train_layout(m, nullptr, B, S, T) + 256;
}

extern "C" int pa_model_train_fwd(pa_model* m, const pa_batch* batch, void* ws, int64_t ws_bytes, uint32_t seed,
                                  int32_t training, float* stats, void* stream) {
    if (!m || !m->bound || !batch || !ws || !stats) return PA_EINVAL;
    if (batch->B <= 0 || batch->S <= 0 || batch->T <= 0 || !batch->input_idx[0] || !batch->input_mask) return PA_EINVAL;
    if (batch->output_value && (!batch->output_label || !batch->output_mask)) return PA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ws) & 255) != 0) return PA_EALIGN;
    const size_t need = pa_train_layout(m, (char*)ws, batch->B, batch->S, batch->T);
    if ((int64_t)need > ws_bytes) return PA_EINVAL;
    m->batch = *batch; m->B = batch->B; m->S = batch->S; m->T = batch->T;
    if (batch->cu_in) {
        if (!batch->rowmap || batch->n_valid <= 0 || batch->n_valid > batch->B * batch->S) return PA_EINVAL;
        m->NE = batch->n_valid;
    } else {
        m->NE = batch->B * batch->S;
    }
    m->seed = seed; m->p_drop = training ? m->cfg.dropout : 0.f;
    m->stats = stats;
    if (m->attn_ws && (m->attn_ws != m->attn_ws_zeroed || m->attn_ws_bytes != m->attn_ws_zeroed_bytes)) {
        // contract of pa_attn_args.ws: ticket words zero before the first launch that uses the buffer (launches leave them zero)
        if (hipMemsetAsync(m->attn_ws, 0, (size_t)pa_attn_ws_ticket_bytes(m->attn_ws_bytes), (hipStream_t)stream) != hipSuccess) return PA_EINVAL;
        m->attn_ws_zeroed = m->attn_ws; m->attn_ws_zeroed_bytes = m->attn_ws_bytes;
    }
    m->have_fwd = false;
    int rc = pa_train_forward_impl(m, stream);
    if (rc == 0) m->have_fwd = batch->output_value != nullptr;
    return rc;
}

extern "C" int pa_model_set_upstream(pa_model* m, const float* upstream) {
    if (!m) return PA_EINVAL;
    m->upstream = upstream;
    return 0;
}
extern "C" int32_t pa_model_stats_floats(void) { return PA_MODEL_STATS_FLOATS; }
extern "C" int pa_model_train_num_segments(const pa_model* m) {
    return m ? m->cfg.n_dec + m->cfg.n_enc + 4 : PA_EINVAL;
}

extern "C" int pa_model_train_bwd(pa_model* m, int32_t seg_lo, int32_t seg_hi, float gscale, void* stream) {
    if (!m || !m->have_fwd) return PA_EINVAL;
    for (int i = 0; i < m->n_params; ++i) if (!m->gr[i]) return PA_EINVAL;
    const int nseg = m->cfg.n_dec + m->cfg.n_enc + 4;
    if (seg_lo < 0 || seg_hi > nseg || seg_lo > seg_hi) return PA_EINVAL;
    for (int s = seg_lo; s < seg_hi; ++s) RC(backward_segment(m, s, gscale, stream));
    if (seg_hi == nseg && m->side_on) { RC(join_side(m, 0, stream)); RC(join_side(m, 1, stream)); m->select_set(0); }
    return 0;
}
// Segments by which gradient finality lags behind pa_model_train_bwd: with the side stream on, the gradients of segment
// s are final (in `stream` order) once segment s+2 has been enqueued, or after the last segment.  0 = immediately.
extern "C" int pa_model_grad_lag(const pa_model* m) { return (m && m->side_on) ? 2 : 0; }

extern "C" int pa_model_tensor(pa_model* m, int32_t which, void** ptr, int64_t* numel) {
    if (!m || !ptr || !numel || m->B == 0) return PA_EINVAL;
    const int64_t d = m->cfg.d_model;
    switch (which) {
        case PA_T_MEMORY: *ptr = m->cfg.has_enc_norm ? m->memory : m->X[m->cfg.n_enc]; *numel = (int64_t)m->NE * d; return 0;
        case PA_T_HIDDENS: *ptr = m->hid; *numel = (int64_t)m->B * m->T * d; return 0;
        case PA_T_VOCAB_LOGITS: *ptr = m->vlog; *numel = (int64_t)m->B * m->T * m->ldv; return 0;
        case PA_T_PTR_LOGITS: *ptr = m->plog; *numel = (int64_t)m->B * m->T * m->T; return 0;
        default: break;
    }
    if (which >= PA_T_ENC_FFN(0) && which < PA_T_ENC_FFN(m->cfg.n_enc) && (int)m->ea.size() == m->cfg.n_enc) {
        *ptr = m->ea[which - PA_T_ENC_FFN(0)].hff; *numel = (int64_t)m->NE * m->cfg.d_ff; return *ptr ? 0 : PA_EINVAL;
    }
    if (which >= PA_T_DEC_FFN(0) && which < PA_T_DEC_FFN(m->cfg.n_dec) && (int)m->da.size() == m->cfg.n_dec) {
        *ptr = m->da[which - PA_T_DEC_FFN(0)].hff; *numel = (int64_t)m->B * m->T * m->cfg.d_ff; return *ptr ? 0 : PA_EINVAL;
    }
    return PA_EINVAL;
}
