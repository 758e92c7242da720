// Cross-attention of the greedy decode step in ABSORBED ("multi-query") form - bf16, d_model 512, at most 8 heads.
//
// Reference: plankassembly/models.py:284-307 runs nn.TransformerDecoderLayer's cross-attention with K = W_k memory + b_k and
// V = W_v memory + b_v.  The K/V-cache form of it (dec_attn_kernel) streams 2 x [B][S][d] per layer and step: at B 256, S 1024 that
// is 537 MB per launch, 3.2 of the 4.9 GB a decode step moves, and the kernel sits at the HBM wall.  The projections are linear, so
//     q_h . k_s  =  (W_k,h^T q_h) . m_s  +  q_h . b_k,h          (the second term is constant over s: it cancels in the softmax)
//     sum_s p_s v_s  =  W_v,h (sum_s p_s m_s)  +  b_v,h           (sum_s p_s = 1)
// i.e. every head can attend over the ENCODER OUTPUT ROWS m_s themselves with a 512-wide query qt_h = W_k,h^T q_h, and the value
// projection moves behind the softmax.  All heads of a batch element then read the same rows: one [S][d] stream per layer and step
// (268 MB) instead of two - half the bytes of the kernel that bounds the step - for 8 x the (tiny) arithmetic, which goes to the
// matrix pipe: per 16-key tile S^T[16 keys][16 head slots] = M_tile[16][512] Qt^T (16 x v_mfma_f32_16x16x32_bf16) and
// O^T[512][16 head slots] += M_tile^T P^T (32 x v_mfma_f32_16x16x16_bf16, the A operand read from the same natural LDS image with
// ds_read_b64_tr_b16).  The value side is folded into the Linear behind the attention (pa_decode_begin: W~o = W_o,h W_v,h, one
// [d][H d] matrix per layer); the query side is a small launch of its own (mq_expand_q_kernel).
//
// One block per batch element, four waves.  All waves walk every 16-key tile: a wave recomputes the tile's 16 x 16 scores (the
// matrix pipe has the room) and owns 128 of the 512 output dims (8 accumulator blocks = 32 registers), so there is no per-wave
// partial to merge and the registers stay free for read-ahead.  Tiles are DMA'd straight into an 8-stage LDS ring
// (global_load_lds, 16 KB per tile, a wave issues 4 of a tile's 16 rows): 7 tiles = 112 KB per CU are in flight while one is
// multiplied; one raw s_barrier per tile (counted s_waitcnt vmcnt - a __syncthreads() would drain the DMA queue).
#pragma once

namespace {
typedef short mq_s16x4 __attribute__((ext_vector_type(4)));
constexpr int MQ_D = 512, MQ_KT = 16, MQ_ROW = MQ_D * 2, MQ_TILE = MQ_KT * MQ_ROW, MQ_NS = 8, MQ_MAXH = 8;
constexpr int MQ_MAXS = 16384;                // key-padding mask bytes kept in LDS behind the ring

// LDS image of a tile: row r (key) at r * 1024; its 16-byte chunk c sits at position (c & 48) | ((c ^ r) & 15) so that the 16 rows a
// fragment read touches fall into 16 different bank groups.
// max / sum over the four 16-lane rows of a wave (same column = head slot): gfx950's v_permlane16_swap / v_permlane32_swap exchange
// whole rows between two registers in the VALU (SWAP), against two ds_bpermute round trips through the LDS crossbar.
template <bool SWAP> __device__ __forceinline__ float mq_rows_max(float v) {
    if constexpr (SWAP) {
        const unsigned u = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        const float a = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
        const unsigned ua = __float_as_uint(a);
        const auto r2 = __builtin_amdgcn_permlane32_swap(ua, ua, false, false);
        return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
    } else {
        v = fmaxf(v, __shfl_xor(v, 16));
        return fmaxf(v, __shfl_xor(v, 32));
    }
}

// Range blocks (round 6): with nparts > 1 a batch element's key tiles are walked by nparts blocks (blockIdx.x = b * nparts + part),
// whose partial (O^T, m, l) meet in `sp` - MQ_SP_BYTES per block, behind one ticket word per batch element - and are merged by the
// last block to arrive (split_merge.h).  One block per element streams its 1 MB of memory rows at what ONE CU's DMA sustains
// (~22 GB/s: 47 us per launch at S 1024 whatever the batch); at the reference's evaluation batch of 16 that was 16 of 256 CUs
// busy and 44 % of the decode step (profiles/r06_decode_small_batch.txt).
constexpr int MQ_SP_BYTES = 40 * 1024;
template <int AUX, bool SWAP, bool MASK>
__global__ __launch_bounds__(256, 1) void dec_cross_mq_kernel(bf16* ctx, const bf16* qt, const bf16* mem, const uint8_t* kpm,
                                                              const int32_t* cu, int S, int H, int nparts, char* sp, const int32_t* t_dev) {
    extern __shared__ __attribute__((aligned(256))) char mq_smem[];
    const int b = blockIdx.x / nparts, part = blockIdx.x - b * nparts, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int row0 = cu ? cu[b] : b * S;
    const int Lk = t_dev ? *t_dev + 1 : (cu ? cu[b + 1] - row0 : S);     // (t_dev: rows 0 .. t of a [B][S] row cache - the self-attention form)
    bf16* out = ctx + (size_t)b * H * MQ_D;
    if (Lk <= 0) {                                                      // (block-uniform) no key: zeros, as dec_attn_kernel
        if (part == 0) for (int idx = tid; idx < H * MQ_D; idx += 256) out[idx] = (bf16)0.f;
        return;
    }
    const uint8_t* mk = MASK ? kpm + (size_t)b * S : nullptr;      // (MASK: dense rows with a key-padding mask)
    uint8_t* mlds = reinterpret_cast<uint8_t*>(mq_smem + MQ_NS * MQ_TILE);
    const int ntiles = (Lk + MQ_KT - 1) / MQ_KT;
    int t0 = 0, t1 = ntiles;                                            // this block's key tiles
    if (nparts > 1) split_range(ntiles, part, nparts, t0, t1);
    if constexpr (MASK) {
        for (int s = tid; s < ntiles * MQ_KT; s += 256) mlds[s] = s < Lk ? mk[s] : (uint8_t)1;
        __syncthreads();
    }
    // query fragments (B operand of the score product): head slot n, dims 32 ks + 8 g .. + 7; slots >= H are zero
    u32x4 qf[16];
    {
        const bf16* qrow = qt + ((size_t)b * H + min(n, H - 1)) * MQ_D + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            u32x4 v = *reinterpret_cast<const u32x4*>(qrow + 32 * ks);
            if (n >= H) v = u32x4{0u, 0u, 0u, 0u};
            qf[ks] = v;
        }
        // the query is complete HERE, in front of the DMA prologue: left to itself hipcc waits for it with vmcnt(0) behind the
        // prologue, i.e. for all seven tiles, before the first MFMA
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]),
                     "+v"(qf[8]), "+v"(qf[9]), "+v"(qf[10]), "+v"(qf[11]), "+v"(qf[12]), "+v"(qf[13]), "+v"(qf[14]), "+v"(qf[15]) :: "memory");
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)mq_smem;
    const uint32_t abase = (uint32_t)n * MQ_ROW + (uint32_t)((g ^ n) << 4);                 // score A operand: row n, chunk 4 ks + g
    const int keyt = 4 * g + (n >> 2);
    // transposing read of this wave's dims 128 wave + 16 nb + 4 (n & 3) ..: row keyt, chunk 16 wave + 2 nb + ((n >> 1) & 1)
    const uint32_t tbase = (uint32_t)keyt * MQ_ROW + (uint32_t)((((n >> 1) & 1) ^ keyt) << 4) + (uint32_t)(n & 1) * 8u;
    uint32_t toffr[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) toffr[nb] = (uint32_t)wave * 256u + (tbase ^ (uint32_t)(nb << 5));
    const char* const mbase = reinterpret_cast<const char*>(mem + (size_t)row0 * MQ_D);
    const uint32_t lch = (uint32_t)(lane & 48);
    // The DMA is inline asm on purpose: as a builtin hipcc knows an LDS-DMA write is in flight and, unable to tell which reads it
    // may alias, puts s_waitcnt vmcnt(0) in front of the transposing LDS reads - the whole ring would be waited for every tile.
    // Ordering is ours: counted s_waitcnt vmcnt + s_barrier below (the compiler's own vmcnt waits only become more conservative).
    // (m0 is set and consumed inside one asm statement; it is a reserved register that hipcc never keeps a value in across code.)
    auto issue = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)(t & (MQ_NS - 1)) * MQ_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * wave + i;                                 // (wave-uniform) this wave's rows of the tile
            const int row = min(t * MQ_KT + r, Lk - 1);                 // rows past the end re-read the last row; masked below
            const uint32_t c = lch | (uint32_t)((lane ^ r) & 15);
            const char* src = mbase + (size_t)row * MQ_ROW + c * 16u;
            const uint32_t ldst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)r * MQ_ROW));
            if constexpr (AUX == 2)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(ldst) : "memory");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(ldst) : "memory");
        }
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 fa[16];                                                       // the 16 score fragments of a tile, read one tile ahead
    auto read_scores = [&](int t) {
        const char* base = mq_smem + (t & (MQ_NS - 1)) * MQ_TILE;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            fa[ks] = *reinterpret_cast<const u32x4*>(base + (abase ^ (uint32_t)((ks & 3) << 6)) + (ks >> 2) * 256);
        __builtin_amdgcn_sched_barrier(0);                              // all reads issued here, not re-serialised in front of their MFMAs
    };
    auto scores = [&]() -> f32x4 {
        f32x4 sp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            sp[ks & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&fa[ks]), *reinterpret_cast<const bf16x8*>(&qf[ks]),
                                                                 sp[ks & 3], 0, 0, 0);
        return (sp[0] + sp[1]) + (sp[2] + sp[3]);
    };
    // softmax update + O^T += M^T P^T for tile t; s4[i]: key t * 16 + 4 g + i, head slot n (scale * log2 e came in through the query)
    // the tile's transposed fragments (A operand of O^T += M^T P^T: dims 128 wave + 16 nb .., keys 4 g ..): requested right behind the
    // score MFMAs, in FRONT of the next tile's 16 score fragments (LDS returns in order: they land first)
    mq_s16x4 ta[8];
    auto read_tr = [&](int t) {
        const uint32_t soff = (uint32_t)(t & (MQ_NS - 1)) * MQ_TILE;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
            ta[nb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mq_s16x4*)(mq_smem + soff + toffr[nb]));
    };
    auto finish = [&](int t, const f32x4& s4, uint32_t m4) {
        const int key0 = t * MQ_KT + 4 * g;
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = (key0 + i < Lk) && !((m4 >> (8 * i)) & 0xffu);
            s[i] = valid ? s4[i] : -INFINITY;
        }
        const float mx = mq_rows_max<SWAP>(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = __builtin_amdgcn_exp2f(s[i] - m_safe);
        l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {               // (wave-uniform) a reference point moved: rescale
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] *= alpha;
        }
        u32x2 pbu; pbu[0] = pack_bf16(p[0], p[1]); pbu[1] = pack_bf16(p[2], p[3]);
        const mq_s16x4 pb = *reinterpret_cast<const mq_s16x4*>(&pbu);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ta[nb], pb, acc[nb], 0, 0, 0);   // O^T[128 wave + 16 nb + 4 g + i][n]
    };

    // Ring schedule (8 stages, tile t in stage t & 7): tiles 0..6 are requested up front; in the middle of iteration t - behind the
    // score MFMAs of tile t - a wave waits for ITS rows of tile t + 1 (at most the 5 younger tiles' 20 DMAs still outstanding),
    // the barrier makes that true for every wave's rows and says every wave is through with tile t - 1, whose stage then takes tile
    // t + 7; the score fragments of tile t + 1 are read there, one tile ahead, so their LDS latency hides behind tile t's softmax.
#pragma unroll
    for (int i = 0; i < MQ_NS - 1; ++i)
        if (t0 + i < t1) issue(t0 + i);
    if (t1 - t0 >= MQ_NS - 1) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t0 < t1) read_scores(t0);
    for (int t = t0; t < t1; ++t) {
        const f32x4 s4 = scores();
        uint32_t m4 = 0;                                                // mask bytes of this lane's 4 keys (read ahead of the next tile's fragments)
        if constexpr (MASK) m4 = *reinterpret_cast<const uint32_t*>(mlds + t * MQ_KT + 4 * g);
        read_tr(t);
        if (t + 1 < t1) {
            if (t + MQ_NS - 2 < t1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // tiles t + 1 .. t + 6 outstanding
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + MQ_NS - 1 < t1) issue(t + MQ_NS - 1);
            read_scores(t + 1);
        }
        finish(t, s4, m4);
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    if (nparts > 1) {
        // publish (O^T, m, l); the element's last range block to arrive rescales all of them to the common reference point and stores
        char* const pbase = sp + 256 * (((size_t)gridDim.x / nparts * 4 + 255) / 256) + (size_t)b * nparts * MQ_SP_BYTES;
        const SplitOut so(pbase + (size_t)part * MQ_SP_BYTES, MQ_SP_BYTES);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) so.put(nb, 256, acc[nb]);
        so.put(8, 256, f32x4{m_run, l_run, 0.f, 0.f});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the ring's last LDS reads are done before its first bytes become the flag)
        if (!split_arrive(reinterpret_cast<int*>(sp) + b, nparts, reinterpret_cast<int*>(mq_smem))) return;
        // merged IN RANGE ORDER (own partial re-read like the others): the same sums whichever block arrives last
        float m_all = -INFINITY;
        for (int pp = 0; pp < nparts; ++pp) m_all = fmaxf(m_all, split_get(pbase + (size_t)pp * MQ_SP_BYTES, 8, 256, MQ_SP_BYTES)[0]);
        const float ms = m_all == -INFINITY ? 0.f : m_all;
        l_run = 0.f;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pp = 0; pp < nparts; ++pp) {
            const char* ob = pbase + (size_t)pp * MQ_SP_BYTES;
            const f32x4 ml = split_get(ob, 8, 256, MQ_SP_BYTES);
            const float a1 = __builtin_amdgcn_exp2f(ml[0] - ms);
            l_run += ml[1] * a1;
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] += split_get(ob, nb, 256, MQ_SP_BYTES) * a1;
        }
    }
    if (n < H) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16* orow = out + (size_t)n * MQ_D + 128 * wave + 4 * g;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
            u32x2 u; u[0] = pack_bf16(acc[nb][0] * inv, acc[nb][1] * inv); u[1] = pack_bf16(acc[nb][2] * inv, acc[nb][3] * inv);
            *reinterpret_cast<u32x2*>(orow + 16 * nb) = u;
        }
    }
}

// ---- exact-f32 form (the parity path of the greedy decode: token indices bit-exact against the reference) ---------------------------
// Same algebra, f32 memory rows (2 KB per key), v_mfma_f32_16x16x4_f32 (1/16 of the bf16 rate: the score product cannot be recomputed
// by every wave).  A wave takes the 128 dims it owns in BOTH products: its partial scores S_w^T[16 keys][16 head slots] over those dims
// go through a 1 KB LDS slot per wave, every wave adds the four partials in the same order (identical statistics in all waves), and
// O^T[128 w ..][head] += M^T P^T with key <-> k-slot mapping key = 4 g + kk so that a lane's own four probabilities ARE the B operand of
// k-step kk (no exchange).  The contraction / row indices of the MFMAs are assigned so that a lane's operands are CONTIGUOUS in a row:
// scores - lane group g contracts dims 128 w + 32 g .. + 31 (8 ds_read_b128 instead of 32 scalar reads); P^T M - row m of accumulator
// block nb is dim 128 w + 8 m + nb (two ds_read_b128 per key).  With scalar reads the launch took 147 us (latency of 64 dependent LDS
// reads per tile), level with the K / V-cache kernel it replaces.  Requesting ALL of an iteration's LDS reads at its top (partials,
// next tile's score operands, this tile's P^T M operands) was measured slower: 148 vs 135 us stand-alone, 1.873 vs 1.849 ms per step.  4-stage ring of 32 KB tiles; software pipeline: iteration t finishes tile t (softmax, P^T M) and forms the
// partial scores of tile t + 1, so one barrier per tile serves the ring and the partial-score hand-over.
constexpr int MQF_ROW = MQ_D * 4, MQF_TILE = MQ_KT * MQF_ROW, MQF_NS = 4, MQF_SCR = 2 * 4 * 1024;
template <bool MASK>
__global__ __launch_bounds__(256, 1) void dec_cross_mq32_kernel(float* ctx, const float* qt, const float* mem, const uint8_t* kpm,
                                                                const int32_t* cu, int S, int H, const int32_t* t_dev) {
    extern __shared__ __attribute__((aligned(256))) char mq_smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int row0 = cu ? cu[b] : b * S;
    const int Lk = t_dev ? *t_dev + 1 : (cu ? cu[b + 1] - row0 : S);     // (t_dev: rows 0 .. t of a [B][S] cache - the self-attention form)
    float* out = ctx + (size_t)b * H * MQ_D;
    if (Lk <= 0) {
        for (int idx = tid; idx < H * MQ_D; idx += 256) out[idx] = 0.f;
        return;
    }
    const uint8_t* mk = MASK ? kpm + (size_t)b * S : nullptr;
    char* scr = mq_smem + MQF_NS * MQF_TILE;                            // [2][4 waves][64 lanes] f32x4
    uint8_t* mlds = reinterpret_cast<uint8_t*>(scr + MQF_SCR);
    const int ntiles = (Lk + MQ_KT - 1) / MQ_KT;
    if constexpr (MASK) {
        for (int s = tid; s < ntiles * MQ_KT; s += 256) mlds[s] = s < Lk ? mk[s] : (uint8_t)1;
        __syncthreads();
    }
    // query values (B operand of the partial score product): head slot n, k-step 4 j + e <-> dim 128 wave + 32 g + 4 j + e
    f32x4 qf[8];
    {
        const float* qrow = qt + ((size_t)b * H + min(n, H - 1)) * MQ_D + 128 * wave + 32 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[j] = n < H ? *reinterpret_cast<const f32x4*>(qrow + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]), "+v"(qf[4]), "+v"(qf[5]), "+v"(qf[6]), "+v"(qf[7]) :: "memory");
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)mq_smem;
    // LDS image of a tile: row r (key) at r * 2048; 16-byte chunk c (4 dims) at position (c & ~15) | ((c ^ r) & 15).
    // score A operand (row = key n, dims 128 wave + 32 g + 4 j ..): chunk 32 wave + 8 g + j
    const uint32_t sbase = (uint32_t)n * MQF_ROW + 512u * (uint32_t)wave + 256u * (uint32_t)(g >> 1) + ((((uint32_t)n) ^ (8u * (uint32_t)(g & 1))) << 4);
    // P^T M A operand for k-step kk (k <-> key 4 g + kk; row m = n <-> dims 128 wave + 8 n + 0..7): chunks 32 wave + 2 n + {0, 1}
    uint32_t pbase[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const uint32_t key = 4u * (uint32_t)g + (uint32_t)kk;
        pbase[kk] = key * MQF_ROW + 512u * (uint32_t)wave + 256u * (uint32_t)(n >> 3) + ((((2u * (uint32_t)(n & 7)) ^ key) & 15u) << 4);
    }
    const char* const mbase = reinterpret_cast<const char*>(mem + (size_t)row0 * MQ_D);
    // per-lane byte offset of the chunk this lane fetches for row 4 wave + i, half hh (loop-invariant: 8 registers)
    uint32_t coff[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const uint32_t pz = 64u * hh + (uint32_t)lane, r = 4u * (uint32_t)wave + (uint32_t)i;
            coff[i][hh] = ((pz & ~15u) | ((pz ^ r) & 15u)) * 16u;
        }
    auto issue = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)(t & (MQF_NS - 1)) * MQF_TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * wave + i;
            const int row = min(t * MQ_KT + r, Lk - 1);
            const char* rowp = mbase + (size_t)row * MQF_ROW;            // (wave-uniform)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const char* src = rowp + coff[i][hh];
                const uint32_t ldst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)r * MQF_ROW + 1024u * hh));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(ldst) : "memory");
            }
        }
    };
    auto wait_vm = [&](int tiles_younger) {                             // (wave-uniform) at most that many younger tiles (8 DMAs each) outstanding
        if (tiles_younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (tiles_younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // partial scores of tile t over this wave's 128 dims -> this wave's slot of scratch buffer t & 1
    auto partial_scores = [&](int t) {
        const char* base = mq_smem + (t & (MQF_NS - 1)) * MQF_TILE;
        f32x4 av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) av[j] = *reinterpret_cast<const f32x4*>(base + (sbase ^ (uint32_t)(j << 4)));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], qf[j][e], sp[e], 0, 0, 0);
        const f32x4 s4 = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        *reinterpret_cast<f32x4*>(scr + (t & 1) * 4096 + wave * 1024 + lane * 16) = s4;
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int npro = min(ntiles, MQF_NS - 1);
    for (int t = 0; t < npro; ++t) issue(t);
    wait_vm(npro - 1);
    __builtin_amdgcn_s_barrier();
    partial_scores(0);
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) wait_vm(t + 2 < ntiles ? 1 : 0);            // my rows of tile t + 1 (tile t + 2 may still be on its way)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // my partial scores of tile t are written
        __builtin_amdgcn_s_barrier();                                   // ... everyone's; tile t + 1 landed; stage of tile t - 1 free
        asm volatile("" ::: "memory");
        if (t + MQF_NS - 1 < ntiles) issue(t + MQF_NS - 1);
        f32x4 s4;
        {
            const char* sb = scr + (t & 1) * 4096 + lane * 16;
            const f32x4 p0 = *reinterpret_cast<const f32x4*>(sb), p1 = *reinterpret_cast<const f32x4*>(sb + 1024);
            const f32x4 p2 = *reinterpret_cast<const f32x4*>(sb + 2048), p3 = *reinterpret_cast<const f32x4*>(sb + 3072);
            s4 = (p0 + p1) + (p2 + p3);
        }
        uint32_t m4 = 0;
        if constexpr (MASK) m4 = *reinterpret_cast<const uint32_t*>(mlds + t * MQ_KT + 4 * g);
        if (t + 1 < ntiles) partial_scores(t + 1);
        // s4[i]: key t * 16 + 4 g + i, head slot n (log2 domain)
        const int key0 = t * MQ_KT + 4 * g;
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = (key0 + i < Lk) && !((m4 >> (8 * i)) & 0xffu);
            s[i] = valid ? s4[i] : -INFINITY;
        }
        const float mx = mq_rows_max<true>(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);       // (v_exp_f32, 1 ulp; arguments <= 0)
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = __builtin_amdgcn_exp2f(s[i] - m_safe);
        l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) acc[nb] *= alpha;
        }
        const char* base = mq_smem + (t & (MQF_NS - 1)) * MQF_TILE;
        f32x4 pv[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 2; ++e) pv[kk][e] = *reinterpret_cast<const f32x4*>(base + (pbase[kk] ^ (uint32_t)(e << 4)));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb)                              // O^T[128 wave + 8 (4 g + i) + nb][n]
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[kk][nb >> 2][nb & 3], p[kk], acc[nb], 0, 0, 0);
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    if (n < H) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* orow = out + (size_t)n * MQ_D + 128 * wave + 32 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(orow + 8 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * inv;
            *reinterpret_cast<f32x4*>(orow + 8 * i + 4) = f32x4{acc[4][i], acc[5][i], acc[6][i], acc[7][i]} * inv;
        }
    }
}

// ---- exact f32, eight waves (two per SIMD) ----------------------------------------------------------------------------------------
// The four-wave kernel above keeps the f32 matrix pipe 43 % busy (profiles/r05_decode_f32_cross_mq_sq_counters.txt): with one wave per
// SIMD the LDS round trips, the softmax and the barrier of a tile are all exposed between its two MFMA batches.  Here a wave owns 64
// dims (16 + 16 MFMAs per tile) and the two waves of a SIMD (w and w + 4: a workgroup's waves go to the SIMDs in cyclic order) run the
// iteration's two halves in OPPOSITE order - one forms the next tile's partial scores while the other is in its softmax / LDS phase.
constexpr int MQ8_SCR = 2 * 8 * 1024;
template <bool MASK>
__global__ __launch_bounds__(512, 1) void dec_cross_mq32w8_kernel(float* ctx, const float* qt, const float* mem, const uint8_t* kpm,
                                                                  const int32_t* cu, int S, int H, const int32_t* t_dev, int nparts, char* sp) {
    extern __shared__ __attribute__((aligned(256))) char mq_smem[];
    const int b = blockIdx.x / nparts, part = blockIdx.x - b * nparts, tid = threadIdx.x, lane = tid & 63;     // (range blocks: dec_cross_mq_kernel)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int row0 = cu ? cu[b] : b * S;
    const int Lk = t_dev ? *t_dev + 1 : (cu ? cu[b + 1] - row0 : S);
    float* out = ctx + (size_t)b * H * MQ_D;
    if (Lk <= 0) {
        if (part == 0) for (int idx = tid; idx < H * MQ_D; idx += 512) out[idx] = 0.f;
        return;
    }
    const uint8_t* mk = MASK ? kpm + (size_t)b * S : nullptr;
    char* scr = mq_smem + MQF_NS * MQF_TILE;                            // [2][8 waves][64 lanes] f32x4
    uint8_t* mlds = reinterpret_cast<uint8_t*>(scr + MQ8_SCR);
    const int ntiles = (Lk + MQ_KT - 1) / MQ_KT;
    if constexpr (MASK) {
        for (int s = tid; s < ntiles * MQ_KT; s += 512) mlds[s] = s < Lk ? mk[s] : (uint8_t)1;
        __syncthreads();
    }
    // query values: head slot n, k-step 4 j + e <-> dim 64 wave + 16 g + 4 j + e
    f32x4 qf[4];
    {
        const float* qrow = qt + ((size_t)b * H + min(n, H - 1)) * MQ_D + 64 * wave + 16 * g;
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[j] = n < H ? *reinterpret_cast<const f32x4*>(qrow + 4 * j) : f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]) :: "memory");
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)mq_smem;
    // tile image: row r at r * 2048, chunk c at position (c & ~15) | ((c ^ f(r)) & 15) with f(r) = r ^ ((r & 4) << 1): a ds_read_b128 is
    // served in four groups of 16 lanes that each take lanes of TWO 16-lane rows ({0-3, 12-15, 20-27}, ..), so the rows a group
    // touches must land in disjoint bank quads - with the plain key r both products had 2-way conflicts in every group.
    // Scores: row n, chunks 16 wave + 4 g + j
    auto fz = [](uint32_t r) { return r ^ ((r & 4u) << 1); };
    const uint32_t sbase = (uint32_t)n * MQF_ROW + 256u * (uint32_t)wave + ((fz((uint32_t)n) ^ (4u * (uint32_t)g)) << 4);
    // P^T M: k <-> key 4 g + kk, row m = n <-> dims 64 wave + 4 n + 0..3: chunk 16 wave + n
    uint32_t pbase[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const uint32_t key = 4u * (uint32_t)g + (uint32_t)kk;
        pbase[kk] = key * MQF_ROW + 256u * (uint32_t)wave + ((((uint32_t)n) ^ fz(key)) << 4);
    }
    const char* const mbase = reinterpret_cast<const char*>(mem + (size_t)row0 * MQ_D);
    uint32_t coff[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const uint32_t pz = 64u * hh + (uint32_t)lane, r = 2u * (uint32_t)wave + (uint32_t)i;
            coff[i][hh] = ((pz & ~15u) | ((pz ^ fz(r)) & 15u)) * 16u;
        }
    auto issue = [&](int t) {
        const uint32_t dst = lds0 + (uint32_t)(t & (MQF_NS - 1)) * MQF_TILE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = 2 * wave + i;
            const int row = min(t * MQ_KT + r, Lk - 1);
            const char* rowp = mbase + (size_t)row * MQF_ROW;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const char* src = rowp + coff[i][hh];
                const uint32_t ldst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(dst + (uint32_t)r * MQF_ROW + 1024u * hh));
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(src), "s"(ldst) : "memory");
            }
        }
    };
    auto wait_vm = [&](int tiles_younger) {                             // (wave-uniform) 4 DMAs per tile and wave
        if (tiles_younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (tiles_younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto partial_scores = [&](int t) {
        const char* base = mq_smem + (t & (MQF_NS - 1)) * MQF_TILE;
        f32x4 av[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = *reinterpret_cast<const f32x4*>(base + (sbase ^ (uint32_t)(j << 4)));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 sp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) sp[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sp[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], qf[j][e], sp[e], 0, 0, 0);
        const f32x4 s4 = (sp[0] + sp[1]) + (sp[2] + sp[3]);
        *reinterpret_cast<f32x4*>(scr + (t & 1) * 8192 + wave * 1024 + lane * 16) = s4;
    };
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // softmax update of tile t from the summed scores, then O^T += M^T P^T on this wave's 64 dims
    auto finish = [&](int t, const f32x4& s4, uint32_t m4) {
        const int key0 = t * MQ_KT + 4 * g;
        float s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = (key0 + i < Lk) && !((m4 >> (8 * i)) & 0xffu);
            s[i] = valid ? s4[i] : -INFINITY;
        }
        const float mx = mq_rows_max<true>(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
        const float m_new = fmaxf(m_run, mx);
        const float m_safe = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        float p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = __builtin_amdgcn_exp2f(s[i] - m_safe);
        l_run = l_run * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] *= alpha;
        }
        const char* base = mq_smem + (t & (MQF_NS - 1)) * MQF_TILE;
        f32x4 pv[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) pv[kk] = *reinterpret_cast<const f32x4*>(base + pbase[kk]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)                              // O^T[64 wave + 4 (4 g + i) + nb][n]
                acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[kk][nb], p[kk], acc[nb], 0, 0, 0);
    };

    int t0 = 0, t1 = ntiles;                                            // this block's key tiles
    if (nparts > 1) split_range(ntiles, part, nparts, t0, t1);
    const int npro = min(t1 - t0, MQF_NS - 1);
    for (int i = 0; i < npro; ++i) issue(t0 + i);
    wait_vm(npro - 1);
    __builtin_amdgcn_s_barrier();
    if (t0 < t1) partial_scores(t0);
    const bool first_half = wave < 4;                                    // (wave-uniform) which order this wave runs an iteration in
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) wait_vm(t + 2 < t1 ? 1 : 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + MQF_NS - 1 < t1) issue(t + MQF_NS - 1);
        f32x4 s4;
        {
            const char* sb = scr + (t & 1) * 8192 + lane * 16;
            f32x4 pp[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) pp[w] = *reinterpret_cast<const f32x4*>(sb + w * 1024);
            s4 = ((pp[0] + pp[1]) + (pp[2] + pp[3])) + ((pp[4] + pp[5]) + (pp[6] + pp[7]));
        }
        uint32_t m4 = 0;
        if constexpr (MASK) m4 = *reinterpret_cast<const uint32_t*>(mlds + t * MQ_KT + 4 * g);
        if (first_half) {
            if (t + 1 < t1) partial_scores(t + 1);
            finish(t, s4, m4);
        } else {
            finish(t, s4, m4);
            if (t + 1 < t1) partial_scores(t + 1);
        }
    }
    l_run += __shfl_xor(l_run, 16);
    l_run += __shfl_xor(l_run, 32);
    if (nparts > 1) {
        // (as dec_cross_mq_kernel) publish (O^T, m, l); the element's last range block to arrive merges them IN RANGE ORDER - a fixed
        // summation order whichever block arrives last: the token-exact path stays run-to-run identical
        char* const pbase = sp + 256 * (((size_t)gridDim.x / nparts * 4 + 255) / 256) + (size_t)b * nparts * MQ_SP_BYTES;
        const SplitOut so(pbase + (size_t)part * MQ_SP_BYTES, MQ_SP_BYTES);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) so.put(nb, 512, acc[nb]);
        so.put(4, 512, f32x4{m_run, l_run, 0.f, 0.f});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!split_arrive(reinterpret_cast<int*>(sp) + b, nparts, reinterpret_cast<int*>(mq_smem))) return;
        float m_all = -INFINITY;
        for (int pp = 0; pp < nparts; ++pp) m_all = fmaxf(m_all, split_get(pbase + (size_t)pp * MQ_SP_BYTES, 4, 512, MQ_SP_BYTES)[0]);
        const float ms = m_all == -INFINITY ? 0.f : m_all;
        l_run = 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pp = 0; pp < nparts; ++pp) {
            const char* ob = pbase + (size_t)pp * MQ_SP_BYTES;
            const f32x4 ml = split_get(ob, 4, 512, MQ_SP_BYTES);
            const float a1 = __builtin_amdgcn_exp2f(ml[0] - ms);
            l_run += ml[1] * a1;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb] += split_get(ob, nb, 512, MQ_SP_BYTES) * a1;
        }
    }
    if (n < H) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        float* orow = out + (size_t)n * MQ_D + 64 * wave + 16 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(orow + 4 * i) = f32x4{acc[0][i], acc[1][i], acc[2][i], acc[3][i]} * inv;
    }
}

// qt[b][h][j] = sl * sum_c q[b][h dh + c] * W_k[h dh + c][j]  (q = the step's cross-attention query rows, bias included; W_k = rows
// d .. 2d of the layer's bf16 in_proj weight, [d][d]).  A launch of its own behind the (LayerNorm-folded) query Linear: folding
// W_k,h^T W_q,h into that Linear instead makes it a [B][512] x [512][H 512] product - measured 18 us per layer on the 32 x 32-tile
// skinny kernel (four rounds of 1 024 blocks) against 6 + ~4 us for the two launches.  grid (B / 8, H), 512 threads: a thread owns
// one column j of 8 rows; the weights are read once per block (coalesced over j), the 8 x dh query values come from LDS.
constexpr int MQ_XR = 8;
template <int DH, typename T, typename XT = T>   // DH = head dim when it is 64 (every weight load of a thread issued before the first use: one round trip), else 0; XT: type of the layer-input rows
__global__ __launch_bounds__(512) void mq_expand_q_kernel(T* qt, const T* q, int ldq, const T* Wk, int B, int d, int H, float sl,
                                                          const XT* xrow = nullptr, T* xcache = nullptr, const int32_t* t_dev = nullptr, int Tmax = 0) {
    __shared__ __attribute__((aligned(16))) float qs[MQ_XR][MQ_D];                 // (dh <= d)
    const int h = blockIdx.y, r0 = blockIdx.x * MQ_XR, dh = DH ? DH : d / H, tid = threadIdx.x;
    if (xcache && h == 0) {                            // self-attention form: this step's layer-input rows join the row cache (row t of [B][Tmax][d])
        const int t = *t_dev;
        for (int e = tid; e < MQ_XR * d; e += 512) {
            const int r = e / d, j = e - r * d;
            if (r0 + r < B) xcache[((size_t)(r0 + r) * Tmax + t) * d + j] = (T)(float)xrow[(size_t)(r0 + r) * d + j];
        }
    }
    for (int e = tid; e < MQ_XR * dh; e += 512) {
        const int r = e / dh, c = e - r * dh;
        qs[r][c] = r0 + r < B ? (float)q[(size_t)(r0 + r) * ldq + h * dh + c] * sl : 0.f;
    }
    const int j = min(tid, d - 1);                    // this thread's column (d <= 512 = the block)
    const T* w = Wk + (size_t)h * dh * d + j;
    if constexpr (DH > 0) {
        float wf[DH];
#pragma unroll
        for (int c = 0; c < DH; ++c) wf[c] = (float)w[(size_t)c * d];
        __syncthreads();
#pragma unroll 1
        for (int r = 0; r < MQ_XR; ++r) {             // (rolled: unrolled, hipcc hoists all 8 x DH LDS reads and spills)
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < DH; c += 4) {
                const f32x4 q4 = *reinterpret_cast<const f32x4*>(&qs[r][c]);
                a += (q4[0] * wf[c] + q4[1] * wf[c + 1]) + (q4[2] * wf[c + 2] + q4[3] * wf[c + 3]);
            }
            if (tid < d && r0 + r < B) qt[((size_t)(r0 + r) * H + h) * d + j] = (T)a;
        }
    } else {
        __syncthreads();
        float a0[MQ_XR];
#pragma unroll
        for (int r = 0; r < MQ_XR; ++r) a0[r] = 0.f;
#pragma unroll 4
        for (int c = 0; c < dh; ++c) {
            const float f0 = (float)w[(size_t)c * d];
#pragma unroll
            for (int r = 0; r < MQ_XR; ++r) a0[r] += qs[r][c] * f0;
        }
        if (tid < d) {
#pragma unroll
            for (int r = 0; r < MQ_XR; ++r)
                if (r0 + r < B) qt[((size_t)(r0 + r) * H + h) * d + j] = (T)a0[r];
        }
    }
}
// The value side as a launch of its own (exact f32, dh = 64): ao[b][h dh + c] = b_v[h dh + c] + sum_j ctx[b][h][j] W_v[h dh + c][j], followed by
// the layer's ordinary out-projection.  In f32 the folded form (W~o = W_o,h W_v,h, one [B][H d] x [H d][d] product) costs 4 x the flops of the
// two steps on the 1/16-rate f32 matrix pipe: 27 us on the skinny kernel against ~8 + 7.  grid (B / 8, H), 512 threads: thread (c, jp) holds
// 64 weights of row h dh + c (columns 64 jp ..; read from the transposed copy WvT), the 8 context rows come from LDS (broadcast reads), the 8 column
// parts meet in LDS.
// WvT[h][j][c] = W_v[h 64 + c][j]: the value weights with the head's 64 output columns contiguous, so that the 64 lanes (c) of a wave of
// mq_contract_v_kernel read 256 contiguous bytes per load (from W_v itself every lane would start its own 2 KB-strided row).  Once per sequence.
template <typename T>
__global__ __launch_bounds__(256) void mq_transpose_v_kernel(T* WvT, const T* Wv, int d, int H) {
    const int64_t total = (int64_t)H * d * 64;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e & 63), j = (int)((e >> 6) % d), h = (int)(e / ((int64_t)64 * d));
        WvT[e] = Wv[(size_t)(h * 64 + c) * d + j];
    }
}
template <typename T>
__global__ __launch_bounds__(512) void mq_contract_v_kernel(T* ao, int ldo, const T* ctx, const T* WvT, const float* bv, int B, int d, int H) {
    __shared__ __attribute__((aligned(16))) float cs[MQ_XR][MQ_D];
    __shared__ float part[MQ_XR][8][64];
    const int h = blockIdx.y, r0 = blockIdx.x * MQ_XR, tid = threadIdx.x, c = tid & 63, jp = tid >> 6;
    float wf[64];
    {
        const T* w = WvT + ((size_t)h * d + 64 * jp) * 64 + c;
#pragma unroll
        for (int k = 0; k < 64; ++k) wf[k] = (float)w[(size_t)k * 64];
    }
    for (int e = tid; e < MQ_XR * MQ_D; e += 512) {
        const int r = e >> 9, j = e & (MQ_D - 1);
        cs[r][j] = r0 + r < B ? (float)ctx[((size_t)(r0 + r) * H + h) * MQ_D + j] : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < MQ_XR; ++r) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 64; k += 4) {
            const f32x4 q4 = *reinterpret_cast<const f32x4*>(&cs[r][64 * jp + k]);
            a += (q4[0] * wf[k] + q4[1] * wf[k + 1]) + (q4[2] * wf[k + 2] + q4[3] * wf[k + 3]);
        }
        part[r][jp][c] = a;
    }
    __syncthreads();
    const int r = tid >> 6;
    if (r0 + r < B) {
        float a = bv[h * 64 + c];
#pragma unroll
        for (int q = 0; q < 8; ++q) a += part[r][q][c];
        ao[(size_t)(r0 + r) * ldo + h * 64 + c] = (T)a;
    }
}

// W~o (bf16 [d][H d]) and b~o (f32 [d]):  W~o[n][h d + j] = sum_c W_o[n][h dh + c] W_v[h dh + c][j];  b~o = b_o + W_o b_v.  grid d blocks.
template <typename T>
__global__ __launch_bounds__(256) void mq_absorb_o_kernel(T* Wt, float* bt, const float* Wo, const float* bo, const float* Win,
                                                          const float* bin, int d, int H) {
    const int nrow = blockIdx.x, dh = d / H;
    const float* Wv = Win + (size_t)2 * d * d;
    const float* bv = bin + 2 * d;
    for (int col = threadIdx.x; col < H * d; col += 256) {
        const int h = col / d, j = col - h * d;
        float a = 0.f;
        for (int c = 0; c < dh; ++c) a += Wo[(size_t)nrow * d + h * dh + c] * Wv[(size_t)(h * dh + c) * d + j];
        Wt[(size_t)nrow * H * d + col] = (T)a;
    }
    if (threadIdx.x == 0) {
        float a = bo[nrow];
        for (int k = 0; k < d; ++k) a += Wo[(size_t)nrow * d + k] * bv[k];
        bt[nrow] = a;
    }
}

// range blocks per batch element for a launch of B elements with up to S keys: enough blocks to fill the chip, at least
// PLANK_DECODE_MQ_MINT (8) key tiles each; PLANK_DECODE_MQ_PARTS=<n> forces, 1 = never
int mq_parts(int B, int S) {
    static const int force = getenv("PLANK_DECODE_MQ_PARTS") ? atoi(getenv("PLANK_DECODE_MQ_PARTS")) : 0;
    static const int mint = getenv("PLANK_DECODE_MQ_MINT") ? atoi(getenv("PLANK_DECODE_MQ_MINT")) : 8;
    if (force > 0) return force > 64 ? 64 : force;
    const int ntiles = (S + MQ_KT - 1) / MQ_KT;
    int p = 256 / (B > 0 ? B : 1);
    if (p > ntiles / (mint > 0 ? mint : 1)) p = ntiles / (mint > 0 ? mint : 1);
    return p < 1 ? 1 : (p > 64 ? 64 : p);
}
int64_t mq_split_bytes(int B, int nparts) { return nparts > 1 ? (((int64_t)B * 4 + 255) / 256) * 256 + (int64_t)B * nparts * MQ_SP_BYTES : 0; }

int launch_cross_mq(bf16* ctx, const bf16* qt, const bf16* mem, const uint8_t* kpm, const int32_t* cu, int B, int S, int H, int d,
                    hipStream_t s, void* sp = nullptr, int64_t sp_bytes = 0, const int32_t* t_dev = nullptr) {
    if (d != MQ_D || H < 1 || H > MQ_MAXH || B <= 0 || S <= 0 || S > MQ_MAXS) return PA_ESHAPE;
    const int lds = MQ_NS * MQ_TILE + ((!cu && kpm) ? (S + 31) / 16 * 16 : 0);
    // PLANK_DECODE_MQ_NT=0: default-policy DMA instead of non-temporal (aux 2; measured in the step, B 256 x 1024: 0.958 vs 0.989 ms); PLANK_DECODE_MQ_SWAP=0: ds_bpermute row reductions instead of v_permlane*_swap
    static const int nt = getenv("PLANK_DECODE_MQ_NT") ? atoi(getenv("PLANK_DECODE_MQ_NT")) : 1;
    static const int swap = getenv("PLANK_DECODE_MQ_SWAP") ? atoi(getenv("PLANK_DECODE_MQ_SWAP")) : 1;
    static bool attr_done = false;
    constexpr int MAXLDS = MQ_NS * MQ_TILE + MQ_MAXS + 32;
    typedef void (*KernT)(bf16*, const bf16*, const bf16*, const uint8_t*, const int32_t*, int, int, int, char*, const int32_t*);
    static const KernT ks[8] = {dec_cross_mq_kernel<0, false, false>, dec_cross_mq_kernel<0, false, true>, dec_cross_mq_kernel<0, true, false>,
                                dec_cross_mq_kernel<0, true, true>,   dec_cross_mq_kernel<2, false, false>, dec_cross_mq_kernel<2, false, true>,
                                dec_cross_mq_kernel<2, true, false>,  dec_cross_mq_kernel<2, true, true>};
    if (!attr_done) {
        for (KernT k : ks) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
            if (e != hipSuccess) return (int)e;
        }
        attr_done = true;
    }
    const bool mask = !cu && kpm;
    int nparts = sp ? mq_parts(B, S) : 1;
    if (nparts > 1 && (mq_split_bytes(B, nparts) > sp_bytes || (reinterpret_cast<uintptr_t>(sp) & 255))) nparts = 1;
    PA_LAUNCH(ks[(nt ? 4 : 0) + (swap ? 2 : 0) + (mask ? 1 : 0)], dim3(B * nparts), dim3(256), lds, s, ctx, qt, mem, kpm, cu, S, H, nparts,
              static_cast<char*>(sp), t_dev);
    return 0;
}

int launch_cross_mq32(float* ctx, const float* qt, const float* mem, const uint8_t* kpm, const int32_t* cu, int B, int S, int H, int d,
                      hipStream_t s, const int32_t* t_dev = nullptr, void* sp = nullptr, int64_t sp_bytes = 0) {
    constexpr int MAXS32 = 16000;                 // (ring + partial-score slots + mask bytes within 160 KB)
    if (d != MQ_D || H < 1 || H > MQ_MAXH || B <= 0 || S <= 0 || S > MAXS32) return PA_ESHAPE;
    const bool mask = !cu && kpm;
    // PLANK_DECODE_MQ32_W8=0: the four-wave kernel (one wave per SIMD) instead of eight waves with skewed halves
    static const int w8 = getenv("PLANK_DECODE_MQ32_W8") ? atoi(getenv("PLANK_DECODE_MQ32_W8")) : 1;
    static bool attr_done = false;
    if (!attr_done) {
        constexpr int MAXLDS = MQF_NS * MQF_TILE + MQ8_SCR + MAXS32 + 32;
        const void* ks[4] = {reinterpret_cast<const void*>(dec_cross_mq32_kernel<false>), reinterpret_cast<const void*>(dec_cross_mq32_kernel<true>),
                             reinterpret_cast<const void*>(dec_cross_mq32w8_kernel<false>), reinterpret_cast<const void*>(dec_cross_mq32w8_kernel<true>)};
        for (const void* k : ks) {
            hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
            if (e != hipSuccess) return (int)e;
        }
        attr_done = true;
    }
    const int mbytes = mask ? (S + 31) / 16 * 16 : 0;
    if (w8) {
        const int lds = MQF_NS * MQF_TILE + MQ8_SCR + mbytes;
        int nparts = sp ? mq_parts(B, S) : 1;
        if (nparts > 1 && (mq_split_bytes(B, nparts) > sp_bytes || (reinterpret_cast<uintptr_t>(sp) & 255))) nparts = 1;
        if (mask) PA_LAUNCH(dec_cross_mq32w8_kernel<true>, dim3(B * nparts), dim3(512), lds, s, ctx, qt, mem, kpm, cu, S, H, t_dev, nparts, static_cast<char*>(sp));
        else PA_LAUNCH(dec_cross_mq32w8_kernel<false>, dim3(B * nparts), dim3(512), lds, s, ctx, qt, mem, kpm, cu, S, H, t_dev, nparts, static_cast<char*>(sp));
    } else {
        const int lds = MQF_NS * MQF_TILE + MQF_SCR + mbytes;
        if (mask) PA_LAUNCH(dec_cross_mq32_kernel<true>, dim3(B), dim3(256), lds, s, ctx, qt, mem, kpm, cu, S, H, t_dev);
        else PA_LAUNCH(dec_cross_mq32_kernel<false>, dim3(B), dim3(256), lds, s, ctx, qt, mem, kpm, cu, S, H, t_dev);
    }
    return 0;
}

}  // namespace
