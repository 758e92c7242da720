// Internal (not part of the C ABI): host-side model object shared by runtime.hip and decode.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/plank_hip.h"

struct Arena {
    char* base; size_t off;
    void* take(size_t bytes) {
        size_t o = (off + 255) & ~(size_t)255;
        off = o + bytes;
        return base ? base + o : nullptr;
    }
};

static inline uint32_t site_seed(uint32_t base, uint32_t site) {
    uint32_t x = base + 0x9e3779b9u * (site + 1);
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// canonical parameter order (must match plankassembly_amd/models.py PARAM_ORDER)
enum { P_IN_VALUE = 0, P_IN_POS, P_IN_COORD, P_IN_VIEW, P_IN_TYPE, P_Q_COORD, P_Q_POS, P_FIXED_HEAD };
enum { E_IN_W = 0, E_IN_B, E_OUT_W, E_OUT_B, E_L1_W, E_L1_B, E_L2_W, E_L2_B, E_N1_W, E_N1_B, E_N2_W, E_N2_B, E_COUNT };
enum { D_SA_IN_W = 0, D_SA_IN_B, D_SA_OUT_W, D_SA_OUT_B, D_CA_IN_W, D_CA_IN_B, D_CA_OUT_W, D_CA_OUT_B,
       D_L1_W, D_L1_B, D_L2_W, D_L2_B, D_N1_W, D_N1_B, D_N2_W, D_N2_B, D_N3_W, D_N3_B, D_COUNT };
enum { T_VOCAB_W = 0, T_VOCAB_B, T_PTR_W, T_PTR_B, T_SW_W, T_SW_B, T_COUNT };

struct EncAct {   // saved activations of one encoder layer
    void *qkv, *o, *z1, *y1, *hff, *z2; float *lse, *m1, *r1, *m2, *r2;
};
struct DecAct {
    void *qkv, *o_sa, *z1, *y1, *q_ca, *kv_ca, *o_ca, *z2, *y2, *hff, *z3;
    int ld_kv = 0;                     // row stride of kv_ca: 2d, or n_dec * 2d when all layers' K|V projections share one matrix
    float *lse_sa, *lse_ca, *m1, *r1, *m2, *r2, *m3, *r3;
};

struct DecodeLayout;
struct pa_model {
    pa_model_cfg cfg;
    int n_params;
    std::vector<void*> pf, pl, gr;     // f32 params, low-precision (GEMM operand) params, f32 grads
    std::vector<void*> plT;            // optional transposed low-precision weights (NULL = absent)
    bool bound = false;
    // ---- per-step state (valid between train_fwd and train_bwd) ----
    bool have_fwd = false;
    pa_batch batch;
    int B = 0, S = 0, T = 0;
    int NE = 0;                        // encoder rows actually processed: B*S, or the number of valid rows when packed
    uint32_t seed = 0; float p_drop = 0.f;
    size_t esz = 4;
    std::vector<void*> X, Y;           // layer inputs/outputs chain (X[0..n_enc], Y[0..n_dec])
    std::vector<EncAct> ea; std::vector<DecAct> da;
    void *memory = nullptr, *hid = nullptr; float *mem_m = nullptr, *mem_r = nullptr, *hid_m = nullptr, *hid_r = nullptr;
    float *vlog = nullptr, *plog = nullptr, *sw = nullptr, *row_lse = nullptr; void* pfeat = nullptr; int ldv = 0;
    float* stats = nullptr;
    // scratch of the packed encoder self-attention's range blocks (pa_attn_args.ws): FIRST region of the train workspace; its ticket
    // words are zeroed whenever the region moved or changed size (pa_model_train_fwd)
    void* attn_ws = nullptr; int64_t attn_ws_bytes = 0; void* attn_ws_zeroed = nullptr; int64_t attn_ws_zeroed_bytes = 0;
    const float* upstream = nullptr;      // d(loss) of the caller's autograd (pa_model_set_upstream), or nullptr = stats[3]
    // backward temporaries
    void *gA, *gB, *gC, *gD, *gE, *gF, *gQ3, *gKV, *dmem, *dvlog, *dplog; float *dsw, *delta, *partial, *splitws;
    void* gPre = nullptr;                        // ACTIVATION gelu: the FFN pre-activation, recomputed by the backward pass (rows x d_ff)
    size_t splitws_floats = 0;
    void* gBs[3] = {nullptr, nullptr, nullptr}; void* gCs[3] = {nullptr, nullptr, nullptr};   // per-site LN-backward outputs (ffn, cross, self)
    float* lnp[3] = {nullptr, nullptr, nullptr}; pa_ln_finish_desc lnq[PA_MAX_LN_FINISH]; int nlnq = 0;   // queued LayerNorm-backward finishes
    pa_gemm_args dwq[PA_MAX_GROUP]; int ndwq = 0;                               // queued weight-gradient GEMMs of the current segment
    pa_colsum_desc cs[PA_MAX_COLSUM]; int ncs = 0; bool defer_ok = false;       // queued bias-gradient column sums of the current segment
    pa_reduce_desc defer[PA_MAX_REDUCE]; int ndefer = 0; size_t slab_used = 0;   // queued split-K reductions of the current segment
    bool dmem_written = false;
    void* kv_all = nullptr;            // [rows][n_dec * 2d]: cross-attention K|V of every decoder layer (one batched GEMM over the memory)
    void* gKV_all = nullptr; const void* kvT_all = nullptr;   // packed cross-attention K/V path (pa_model_bind_cross_kv_t)
    // Two copies ("parity sets") of every buffer the queued end-of-segment work reads: with the side stream on, that
    // work of segment s runs concurrently with the main stream's segments s+1 (other set) and is joined before s+2.
    struct SegSet { void* gBs[3]; void* gCs[3]; void *gE, *gF, *gQ3, *gKV; float* lnp[3]; float* splitws; };
    SegSet seg_set[2];
    void select_set(int par) {
        const SegSet& s = seg_set[par];
        for (int i = 0; i < 3; ++i) { gBs[i] = s.gBs[i]; gCs[i] = s.gCs[i]; lnp[i] = s.lnp[i]; }
        gE = s.gE; gF = s.gF; gQ3 = s.gQ3; gKV = s.gKV; splitws = s.splitws;
    }
    bool side_on = false;                        // PA_SIDE_STREAM (default on): queued segment work goes to `side`
    void* side = nullptr;                        // hipStream_t
    void* ev_ready[2] = {nullptr, nullptr};      // main -> side: the segment's dY buffers are complete
    void* ev_done[2] = {nullptr, nullptr};       // side -> main: the segment's queued work is complete
    bool ev_pending[2] = {false, false};
    // ---- greedy decode state (decode.hip) ----
    struct DecodeLayout* dec = nullptr;

    int enc_base(int i) const { return P_FIXED_HEAD + i * E_COUNT; }
    int enc_norm() const { return P_FIXED_HEAD + cfg.n_enc * E_COUNT; }
    int dec_base(int i) const { return enc_norm() + 2 + i * D_COUNT; }
    int dec_norm() const { return enc_norm() + 2 + cfg.n_dec * D_COUNT; }
    int tail() const { return dec_norm() + 2; }
};


