// Included by attention.hip (inside its anonymous namespace, after mma_nat / mma_tr_nat and the first-generation f32 kernels).
//
// bf16x3 ("split") attention for the f32 parity path: reference plankassembly/models.py:60-69 (torch
// F.multi_head_attention_forward inside nn.TransformerEncoder/DecoderLayer) with f32 inputs, outputs, softmax statistics and
// dropout exactly as attn_fwd_kernel<float> / attn_bwd_dq_kernel<float> / attn_bwd_dkv_kernel<float> above - the same masks, the
// same deferred-rescale online softmax, the same lse / delta, the same dropout decisions - but every matrix product
// (S = Q K^T, O = P V, dP = dO V^T, dQ = dS K, dK = dS^T Q, dV = P^T dO) runs on the bf16 matrix pipe as
// hi*hi + hi*lo + lo*hi of the operands' bf16 hi / lo parts with f32 accumulation (2^-17 relative per product, see gemm.hip
// "bf16x3"): 12 v_mfma_f32_32x32x16_bf16 (384 cycles) per 32 x 32 x 64 product instead of 32 v_mfma_f32_32x32x2_f32 (2048).
//
// Structure = the first-generation kernels (128 owned rows per block, 64 streamed rows per step, global -> registers -> LDS,
// double buffered, one barrier per step) with two changes that the split makes possible:
//   * a streamed f32 tile is cut into hi / lo when it is written to LDS and kept as TWO natural [64][dh] bf16 images in the bf16
//     kernels' swizzle, so mma_nat<bf16> (A operand = rows) and mma_tr_nat (A operand = columns, ds_read_b64_tr_b16) serve both
//     uses of a tile: no transposed image, half the LDS of the f32 kernels' natural + transposed pair (64 KB per block for all
//     three kernels -> two blocks per CU; the f32 backward kernels need 98 / 132 KB and run one block per CU);
//   * a lane's own row operand (Q, dO, K, V rows in registers) is cut once per block.
template <int DH> struct X3L {
    static constexpr int RBN = BT<DH>::RBN;
    static constexpr int NAT = BT<DH>::NAT;                  // one 64-row bf16 image (8 KB at dh 64)
    static constexpr int TILE = 2 * NAT;                     // [hi image | lo image]
    static constexpr int STG = 2 * TILE;                     // a stage: two tiles (K | V, or Q | dO)
    static constexpr int AUX0 = 2 * STG;                     // both stages' tiles first, then the stages' aux records
    // aux record of a stage.  forward, dQ: [64 mask bytes][64 key-hash words (dropout)][flag: the tile contains a masked key];
    // dK / dV: [64 lse][64 delta][64 row-hash words] of the streamed queries
    static constexpr int AUX_HASH = 64, AUX_FLAG = 64 + 256, AUXS = 768;
    static constexpr int SHM = AUX0 + 2 * AUXS;
};

// four rows x one 16-byte f32 chunk (load_sub<float, DH>) -> hi / lo halves of a bf16 chunk in the two images of `tile`
template <int DH>
__device__ __forceinline__ void x3_store(const u32x4* regs, char* tile, int sb) {
    using A = AT<float, DH>;
    using B = BT<DH>;
    const int cb = sb % A::NCHR, rb = sb / A::NCHR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rb * 4 + i;
        const float x0 = __uint_as_float(regs[i][0]), x1 = __uint_as_float(regs[i][1]);
        const float x2 = __uint_as_float(regs[i][2]), x3 = __uint_as_float(regs[i][3]);
        u32x2 hi, lo;
        hi[0] = pack_bf16(x0, x1); hi[1] = pack_bf16(x2, x3);
        lo[0] = pack_bf16(x0 - bf16_lo(hi[0]), x1 - bf16_hi(hi[0]));
        lo[1] = pack_bf16(x2 - bf16_lo(hi[1]), x3 - bf16_hi(hi[1]));
        const int off = swz_off<B::RBN>(row, cb >> 1) + (cb & 1) * 8;
        *reinterpret_cast<u32x2*>(tile + off) = hi;
        *reinterpret_cast<u32x2*>(tile + B::NAT + off) = lo;
    }
}
// this lane's row operand, cut into hi / lo: NS = dh / 16 fragments of 8 bf16 each (elements 16 s + 8 half .. + 7)
template <int DH>
__device__ __forceinline__ void x3_row_regs(u32x4* hi, u32x4* lo, const float* base, int ld, int row, int nrows, int lane) {
    const int half = lane >> 5;
#pragma unroll
    for (int s = 0; s < BT<DH>::NS; ++s) {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (row < nrows) {
            const float* src = base + (size_t)row * ld + 16 * s + 8 * half;
            v0 = *reinterpret_cast<const f32x4*>(src); v1 = *reinterpret_cast<const f32x4*>(src + 4);
        }
        hi[s][0] = pack_bf16(v0[0], v0[1]); hi[s][1] = pack_bf16(v0[2], v0[3]);
        hi[s][2] = pack_bf16(v1[0], v1[1]); hi[s][3] = pack_bf16(v1[2], v1[3]);
        lo[s][0] = pack_bf16(v0[0] - bf16_lo(hi[s][0]), v0[1] - bf16_hi(hi[s][0]));
        lo[s][1] = pack_bf16(v0[2] - bf16_lo(hi[s][1]), v0[3] - bf16_hi(hi[s][1]));
        lo[s][2] = pack_bf16(v1[0] - bf16_lo(hi[s][2]), v1[1] - bf16_hi(hi[s][2]));
        lo[s][3] = pack_bf16(v1[2] - bf16_lo(hi[s][3]), v1[3] - bf16_hi(hi[s][3]));
    }
}
// acc (32 x 32) += TILE[row0 + (lane & 31)][:] x regs, three terms (the hi fragments serve two of them)
template <int DH>
__device__ __forceinline__ void x3_mma_nat(f32x16& acc, const char* tile, int row0, const u32x4* rh, const u32x4* rl, int lane) {
    using B = BT<DH>;
    const int row = row0 + (lane & 31), half = lane >> 5;
#pragma unroll
    for (int s = 0; s < B::NS; ++s) {
        const int off = swz_off<B::RBN>(row, 2 * s + half);
        const u32x4 ah = *reinterpret_cast<const u32x4*>(tile + off);
        const u32x4 al = *reinterpret_cast<const u32x4*>(tile + B::NAT + off);
        mma16B<bf16>(acc, ah, rh[s]);
        mma16B<bf16>(acc, al, rh[s]);
        mma16B<bf16>(acc, ah, rl[s]);
    }
}
// what bf16 rounding left of a 32 x 32 accumulator tile: pv - float(bf16(pv)), pairwise as mma_tr_nat packs it
__device__ __forceinline__ f32x16 x3_lo_part(const f32x16& pv) {
    f32x16 r;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint32_t u = pack_bf16(pv[2 * w], pv[2 * w + 1]);
        r[2 * w] = pv[2 * w] - bf16_lo(u); r[2 * w + 1] = pv[2 * w + 1] - bf16_hi(u);
    }
    return r;
}
// acc[dt] (32 d x 32) += TILE^T[d][32 rows at row0] x pv, three terms (mma_tr_nat rounds its pv argument to bf16 itself)
template <int DH>
__device__ __forceinline__ void x3_mma_tr(f32x16* acc, const char* tile, int row0, const f32x16& pv, int lane) {
    mma_tr_nat<DH>(acc, tile, row0, pv, lane);
    mma_tr_nat<DH>(acc, tile + BT<DH>::NAT, row0, pv, lane);
    const f32x16 pl = x3_lo_part(pv);
    mma_tr_nat<DH>(acc, tile, row0, pl, lane);
}

// Range-split backward (AttnP::parts_q / parts_kv > 1): a block covers 1 / parts of the streamed side and ADDS its 128 x dh partial
// result to rows that attn_delta_kernel zeroed (f32 atomics, fire and forget).  Why: a packed batch's backward launch lasts as
// long as its longest block - every block is resident from the start, two per CU - and the longest elements stream 16 tiles
// where the average block streams 8.  Built, correct, and MEASURED SLOWER (attention.hip run_bwd, PA_X3_PARTS): off by default.
template <int DH>
__device__ __forceinline__ void x3_add_rows(float* base, int ld, int row, int nrows, const f32x16* acc, int lane) {
    if (row >= nrows) return;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float* dst = base + (size_t)row * ld + dt * 32 + 8 * g + 4 * (lane >> 5);
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(dst + e, acc[dt][4 * g + e]);
        }
}

// Score arithmetic as in the tuned bf16 kernels (the first-generation f32 kernels spend ~25 VALU instructions per score): the key /
// row dropout hash words of a tile are computed ONCE by the 64 staging threads and read from LDS (one multiply + compare + select
// per score), the mask test only runs on tiles that contain a masked key or touch the causal diagonal, the row maximum is a
// v_max3_f32 chain, DROP is a template parameter and the survivors' 1 / (1 - p) is folded into the final normalisation (forward)
// or into one multiply per score (backward).
// stage aux of the forward / dQ kernels: mask byte, key hash word, "tile has a masked key" flag - written by wave 0
template <int DH, bool DROP>
__device__ __forceinline__ void x3_key_aux(char* aux, int tid, uint8_t mreg, uint32_t seed, int key) {
    using X = X3L<DH>;
    if (tid < BSTR) {                                           // (exactly wave 0)
        reinterpret_cast<uint8_t*>(aux)[tid] = mreg;
        if (DROP) reinterpret_cast<uint32_t*>(aux + X::AUX_HASH)[tid] = drop_key_hash(seed, (uint32_t)key);
        const unsigned long long any = __ballot(mreg != 0);
        if (tid == 0) *reinterpret_cast<uint32_t*>(aux + X::AUX_FLAG) = any ? 1u : 0u;
    }
}

// ---- forward (attn_fwd_kernel<float, DH> with split products) -------------------------------------------------------------
template <int DH, bool DROP>
__global__ __launch_bounds__(NTH, 2) void attnx_fwd_kernel(AttnP pin) {
    using A = AT<float, DH>;
    using X = X3L<DH>;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (q0 >= p.Lq) return;
    const float* Qp = reinterpret_cast<const float*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const float* Kp = reinterpret_cast<const float*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const float* Vp = reinterpret_cast<const float*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qw0 = q0 + wave * 32, qrow = qw0 + (lane & 31);

    u32x4 qh[BT<DH>::NS], ql[BT<DH>::NS];
    x3_row_regs<DH>(qh, ql, Qp, p.ldq, qrow, p.Lq, lane);

    int nsteps = (p.Lk + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);

    f32x16 oacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl = p.scale * LOG2E;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow)) : 0u;
    const uint32_t thr = p.drop_thr;

    u32x4 st[A::NITEM][4];
    uint8_t mreg = 0;
    int kreg = 0, st_row0 = 0;                   // st_row0: first row of the tile held in `st` (rows past Lk are zeroed in lstore)
    auto gload = [&](int step) {
        const int k0 = step * BSTR;
        st_row0 = k0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<float, DH>(st[j], Kp, p.ldk, k0, p.Lk, sb);
                else load_sub_raw<float, DH>(st[j], Vp, p.ldv, k0, p.Lk, sb);
            }
        }
        kreg = k0 + (tid & (BSTR - 1));
        // (every thread loads a byte, no guard: a load under `if` is a masked definition hipcc merges with a copy of the loaded register -
        //  a wait right behind it; without a mask the byte comes from the K rows and is ignored in lstore)
        mreg = (mp ? mp : reinterpret_cast<const uint8_t*>(Kp))[min(kreg, p.Lk - 1)];
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * X::STG;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                mask_sub<float, DH>(st[j], st_row0, p.Lk, item % A::NSB);
                x3_store<DH>(st[j], base + (item < A::NSB ? 0 : X::TILE), item % A::NSB);
            }
        }
        x3_key_aux<DH, DROP>(smem + X::AUX0 + buf * X::AUXS, tid, (kreg >= p.Lk) ? (uint8_t)1 : (mp ? mreg : (uint8_t)0), p.drop_seed, kreg);
    };

    if (nsteps > 0) { gload(0); lstore(0); }
    __syncthreads();

    auto body = [&](int step, const int S) {                    // S: the stage (0 / 1) this step's tiles are in
        if (step + 1 < nsteps) gload(step + 1);
        const char* aux = smem + X::AUX0 + S * X::AUXS;
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(aux);
        const int k0 = step * BSTR;
        // wave-uniform: this tile needs per-element mask tests (a masked key in it, or keys beyond this wave's first query row)
        const bool edge = (*reinterpret_cast<const uint32_t*>(aux + X::AUX_FLAG) != 0u) || (p.causal && k0 + BSTR - 1 > qw0);

        f32x16 sacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[0][r] = 0.f; sacc[1][r] = 0.f; }
        x3_mma_nat<DH>(sacc[0], smem + S * X::STG, 0, qh, ql, lane);
        x3_mma_nat<DH>(sacc[1], smem + S * X::STG, 32, qh, ql, lane);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kt][r] *= sl;
        if (edge) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ko = kt * 32 + 8 * g + 4 * half;
                    const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mk + ko);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && k0 + ko + e > qrow);
                        sacc[kt][4 * g + e] = masked ? -INFINITY : sacc[kt][4 * g + e];
                    }
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = max3f(mx, sacc[kt][r], sacc[kt][r + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(mx > m_run + RESCALE_THR)) {                 // deferred rescale, as attn_fwd_kernel
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
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x4 ck = {0u, 0u, 0u, 0u};
                if (DROP) ck = *reinterpret_cast<const u32x4*>(aux + X::AUX_HASH + 4 * (kt * 32 + 8 * g + 4 * half));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(sacc[kt][4 * g + e] - m_safe);
                    lsum += pe;
                    if (DROP) pe = drop_keep2(arow, ck[e], thr) ? pe : 0.f;        // (1 / (1 - p): in the final normalisation)
                    sacc[kt][4 * g + e] = pe;
                }
            }
        l_run += lsum;
        x3_mma_tr<DH>(oacc, smem + S * X::STG + X::TILE, 0, sacc[0], lane);
        x3_mma_tr<DH>(oacc, smem + S * X::STG + X::TILE, 32, sacc[1], lane);

        if (step + 1 < nsteps) lstore(S ^ 1);
        __syncthreads();
    };
    for (int step = 0; step < nsteps; ++step) body(step, step & 1);

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
    float* Op = reinterpret_cast<float*>(p.o) + (size_t)qoff * p.ldo + h * DH;
    store_rows<float, DH>(Op, p.ldo, qrow, p.Lq, oacc, inv, lane);
    if (half == 0 && qrow < p.Lq && p.lse)
        p.lse[((size_t)b * p.H + h) * pin.Lq + qrow] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : 0.f;
}

// ---- backward, dQ (attn_bwd_dq_kernel<float, DH> with split products) ----------------------------------------------------
template <int DH, bool DROP>
__global__ __launch_bounds__(NTH, 2) void attnx_bwd_dq_kernel(AttnP pin) {
    using A = AT<float, DH>;
    using X = X3L<DH>;
    constexpr int NS = BT<DH>::NS;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
    const int parts = pin.parts_q > 1 ? pin.parts_q : 1, part = blockIdx.x % parts, q0 = (blockIdx.x / parts) * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (q0 >= p.Lq) return;
    const float* Qp = reinterpret_cast<const float*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const float* Kp = reinterpret_cast<const float*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const float* Vp = reinterpret_cast<const float*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const float* dOp = reinterpret_cast<const float*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qw0 = q0 + wave * 32, qrow = qw0 + (lane & 31);

    u32x4 qh[NS], ql[NS], doh[NS], dol[NS];
    x3_row_regs<DH>(qh, ql, Qp, p.ldq, qrow, p.Lq, lane);
    x3_row_regs<DH>(doh, dol, dOp, p.lddo, qrow, p.Lq, lane);
    const size_t srow = ((size_t)b * p.H + h) * pin.Lq + qrow;
    const float lse2 = (qrow < p.Lq) ? p.lse[srow] * LOG2E : INFINITY;
    const float dlt = (qrow < p.Lq) ? p.delta[srow] : 0.f;

    int nsteps = (p.Lk + BSTR - 1) / BSTR;
    if (p.causal) nsteps = min(nsteps, (min(q0 + BOWN, p.Lq) + BSTR - 1) / BSTR);
    const int per = (nsteps + parts - 1) / parts, s_lo = part * per;          // this block's key steps [s_lo, nsteps)
    nsteps = min(nsteps, s_lo + per);
    // split launches: attn_delta_kernel zeroed the rows.  Unsplit: an element without keys (cu_k[b + 1] == cu_k[b]) must
    // still store its zero dQ rows - fall through to the store with an empty loop
    if (s_lo >= nsteps && parts > 1) return;

    f32x16 dqacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[dt][r] = 0.f;
    const float sl = p.scale * LOG2E;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow)) : 0u;
    const uint32_t thr = p.drop_thr;
    const float dsc = DROP ? p.drop_scale : 1.0f;

    u32x4 st[A::NITEM][4];
    uint8_t mreg = 0;
    int kreg = 0, st_row0 = 0;                   // st_row0: first row of the tile held in `st` (rows past Lk are zeroed in lstore)
    auto gload = [&](int step) {
        const int k0 = step * BSTR;
        st_row0 = k0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<float, DH>(st[j], Kp, p.ldk, k0, p.Lk, sb);
                else load_sub_raw<float, DH>(st[j], Vp, p.ldv, k0, p.Lk, sb);
            }
        }
        kreg = k0 + (tid & (BSTR - 1));
        // (every thread loads a byte, no guard: a load under `if` is a masked definition hipcc merges with a copy of the loaded register -
        //  a wait right behind it; without a mask the byte comes from the K rows and is ignored in lstore)
        mreg = (mp ? mp : reinterpret_cast<const uint8_t*>(Kp))[min(kreg, p.Lk - 1)];
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * X::STG;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                mask_sub<float, DH>(st[j], st_row0, p.Lk, item % A::NSB);
                x3_store<DH>(st[j], base + (item < A::NSB ? 0 : X::TILE), item % A::NSB);
            }
        }
        x3_key_aux<DH, DROP>(smem + X::AUX0 + buf * X::AUXS, tid, (kreg >= p.Lk) ? (uint8_t)1 : (mp ? mreg : (uint8_t)0), p.drop_seed, kreg);
    };

    if (s_lo < nsteps) { gload(s_lo); lstore(0); }                             // (block-uniform)
    __syncthreads();

    auto body = [&](int step, const int S) {                    // S: the stage (0 / 1) this step's tiles are in
        if (step + 1 < nsteps) gload(step + 1);
        const char* aux = smem + X::AUX0 + S * X::AUXS;
        const uint8_t* mk = reinterpret_cast<const uint8_t*>(aux);
        const int k0 = step * BSTR;
        const bool edge = (*reinterpret_cast<const uint32_t*>(aux + X::AUX_FLAG) != 0u) || (p.causal && k0 + BSTR - 1 > qw0);
        auto sub = [&](auto ktc) {
            constexpr int kt = decltype(ktc)::value;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            x3_mma_nat<DH>(sacc, smem + S * X::STG, kt * 32, qh, ql, lane);                      // S^T = K Q^T
            x3_mma_nat<DH>(dpacc, smem + S * X::STG + X::TILE, kt * 32, doh, dol, lane);         // dP^T = V dO^T
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ko = kt * 32 + 8 * g + 4 * half;
                u32x4 ck = {0u, 0u, 0u, 0u};
                if (DROP) ck = *reinterpret_cast<const u32x4*>(aux + X::AUX_HASH + 4 * ko);
                uint32_t m4 = 0u;
                if (edge) m4 = *reinterpret_cast<const uint32_t*>(mk + ko);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(fmaf(sacc[4 * g + e], sl, -lse2));
                    if (edge) {
                        const bool masked = ((m4 >> (8 * e)) & 0xffu) || (p.causal && k0 + ko + e > qrow);
                        pe = masked ? 0.f : pe;
                    }
                    float dp = dpacc[4 * g + e];
                    if (DROP) dp = drop_keep2(arow, ck[e], thr) ? dp * dsc : 0.f;
                    sacc[4 * g + e] = pe * (dp - dlt) * p.scale;          // dS^T
                }
            }
            x3_mma_tr<DH>(dqacc, smem + S * X::STG, kt * 32, sacc, lane);                        // dQ^T += K^T dS^T
        };
        sub(IC<0>{}); sub(IC<1>{});
        if (step + 1 < nsteps) lstore(S ^ 1);
        __syncthreads();
    };
    for (int step = s_lo; step < nsteps; ++step) body(step, (step - s_lo) & 1);
    float* dQp = reinterpret_cast<float*>(p.dq) + (size_t)qoff * p.lddq + h * DH;
    if (parts > 1) x3_add_rows<DH>(dQp, p.lddq, qrow, p.Lq, dqacc, lane);
    else store_rows<float, DH>(dQp, p.lddq, qrow, p.Lq, dqacc, 1.0f, lane);
}

// ---- backward, dK / dV (attn_bwd_dkv_kernel<float, DH> with split products) ----------------------------------------------
template <int DH, bool DROP, int OCC>      // OCC: blocks per CU the register allocation is made for (1: no spills, one wave per SIMD)
__global__ __launch_bounds__(NTH, OCC) void attnx_bwd_dkv_kernel(AttnP pin) {
    using A = AT<float, DH>;
    using X = X3L<DH>;
    constexpr int NS = BT<DH>::NS;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int b = pin.order ? pin.order[blockIdx.z] : (int)blockIdx.z, h = blockIdx.y;
    const int parts = pin.parts_kv > 1 ? pin.parts_kv : 1, part = blockIdx.x % parts, key0 = (blockIdx.x / parts) * BOWN;
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff);
    if (key0 >= p.Lk) return;
    const float* Qp = reinterpret_cast<const float*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const float* Kp = reinterpret_cast<const float*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const float* Vp = reinterpret_cast<const float*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const float* dOp = reinterpret_cast<const float*>(p.dout) + (size_t)qoff * p.lddo + h * DH;
    const int kw0 = key0 + wave * 32, krow = kw0 + (lane & 31);
    const bool kmasked = (krow >= p.Lk) || (p.kpm && p.kpm[(size_t)b * pin.Lk + krow]);

    u32x4 kh[NS], kl[NS], vh[NS], vl[NS];
    x3_row_regs<DH>(kh, kl, Kp, p.ldk, krow, p.Lk, lane);
    x3_row_regs<DH>(vh, vl, Vp, p.ldv, krow, p.Lk, lane);

    int nsteps = (p.Lq + BSTR - 1) / BSTR;
    int step0 = p.causal ? (key0 / BSTR) : 0;
    {   // this block's query steps: 1 / parts of [step0, nsteps)
        const int per = (nsteps - step0 + parts - 1) / parts;
        step0 += part * per;
        nsteps = min(nsteps, step0 + per);
        if (step0 >= nsteps && parts > 1) return;                              // (the rows were zeroed by attn_delta_kernel)
    }

    f32x16 dkacc[A::NDT], dvacc[A::NDT];
#pragma unroll
    for (int dt = 0; dt < A::NDT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[dt][r] = 0.f; dvacc[dt][r] = 0.f; }
    const float sl = p.scale * LOG2E;
    const uint32_t ckey = DROP ? drop_key_hash(p.drop_seed, (uint32_t)krow) : 0u;
    const uint32_t thr = p.drop_thr;
    const float dsc = DROP ? p.drop_scale : 1.0f;

    u32x4 st[A::NITEM][4];
    float lreg = 0.f, dreg = 0.f;
    uint32_t hreg = 0u;
    int st_row0 = 0;                             // first row of the tile held in `st` (rows past Lq are zeroed / neutralised in lstore)
    auto gload = [&](int step) {
        const int r0 = step * BSTR;
        st_row0 = r0;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                const int sb = item % A::NSB;
                if (item < A::NSB) load_sub_raw<float, DH>(st[j], Qp, p.ldq, r0, p.Lq, sb);
                else load_sub_raw<float, DH>(st[j], dOp, p.lddo, r0, p.Lq, sb);
            }
        }
        {   // (every thread, no guard - see the forward kernel's mask byte)
            const size_t srow0 = ((size_t)b * p.H + h) * pin.Lq;
            const size_t srow = srow0 + min(r0 + (tid & (BSTR - 1)), p.Lq - 1);
            lreg = p.lse[srow];                  // (raw; scaled / replaced for rows past Lq in lstore)
            dreg = p.delta[srow];
            if (DROP) hreg = drop_row_hash(p.drop_seed, (uint32_t)(srow0 + r0 + (tid & (BSTR - 1))));
        }
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * X::STG;
#pragma unroll
        for (int j = 0; j < A::NITEM; ++j) {
            const int item = tid + j * NTH;
            if (item < 2 * A::NSB) {
                mask_sub<float, DH>(st[j], st_row0, p.Lq, item % A::NSB);
                x3_store<DH>(st[j], base + (item < A::NSB ? 0 : X::TILE), item % A::NSB);
            }
        }
        if (tid < BSTR) {
            float* aux = reinterpret_cast<float*>(smem + X::AUX0 + buf * X::AUXS);
            const bool in = st_row0 + tid < p.Lq;
            aux[tid] = in ? lreg * LOG2E : INFINITY; aux[64 + tid] = in ? dreg : 0.f;
            if (DROP) reinterpret_cast<uint32_t*>(aux)[128 + tid] = hreg;
        }
    };

    if (step0 < nsteps) { gload(step0); lstore(0); }
    __syncthreads();

    auto body = [&](int step, const int S) {                    // S: the stage (0 / 1) this step's tiles are in
        // (the next tile's global loads are issued between the two 32-row sub-tiles: their 32 staging registers are then live for
        // half of the step - the kernel sits at the 256-register limit of two blocks per CU)
        const float* aux = reinterpret_cast<const float*>(smem + X::AUX0 + S * X::AUXS);
        const int r0 = step * BSTR;
        const bool diag = p.causal && kw0 + 31 > r0;              // wave-uniform: some (query, key) pair of this tile is above the diagonal
        auto sub = [&](auto qtc) {
            constexpr int qt = decltype(qtc)::value;
            f32x16 sacc, dpacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; dpacc[r] = 0.f; }
            x3_mma_nat<DH>(sacc, smem + S * X::STG, qt * 32, kh, kl, lane);                      // S[q][key]: rows q (registers), column key (lane)
            x3_mma_nat<DH>(dpacc, smem + S * X::STG + X::TILE, qt * 32, vh, vl, lane);           // dP[q][key]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qo = qt * 32 + 8 * g + 4 * half;
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(aux + qo);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(aux + 64 + qo);
                u32x4 h4 = {0u, 0u, 0u, 0u};
                if (DROP) h4 = *reinterpret_cast<const u32x4*>(aux + 128 + qo);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = fast_exp2(fmaf(sacc[4 * g + e], sl, -l4[e]));     // (rows past Lq: lse = +inf -> 0)
                    pe = kmasked ? 0.f : pe;
                    if (diag) pe = (krow > r0 + qo + e) ? 0.f : pe;
                    float dp = dpacc[4 * g + e];
                    float pd = pe;
                    if (DROP) {
                        const bool keep = drop_keep2(h4[e], ckey, thr);
                        dp = keep ? dp * dsc : 0.f;
                        pd = keep ? pe * dsc : 0.f;
                    }
                    sacc[4 * g + e] = pd;                                   // dropped P  -> dV
                    dpacc[4 * g + e] = pe * (dp - d4[e]) * p.scale;         // dS         -> dK
                }
            }
            x3_mma_tr<DH>(dvacc, smem + S * X::STG + X::TILE, qt * 32, sacc, lane);              // dV^T += dO^T P
            x3_mma_tr<DH>(dkacc, smem + S * X::STG, qt * 32, dpacc, lane);                       // dK^T += Q^T dS
        };
        sub(IC<0>{});
        if (step + 1 < nsteps) gload(step + 1);
        sub(IC<1>{});
        if (step + 1 < nsteps) lstore(S ^ 1);
        __syncthreads();
    };
    for (int step = step0; step < nsteps; ++step) body(step, (step - step0) & 1);
    float* dKp = reinterpret_cast<float*>(p.dk) + (size_t)koff * p.lddk + h * DH;
    float* dVp = reinterpret_cast<float*>(p.dv) + (size_t)koff * p.lddv + h * DH;
    if (parts > 1) {
        x3_add_rows<DH>(dKp, p.lddk, krow, p.Lk, dkacc, lane);
        x3_add_rows<DH>(dVp, p.lddv, krow, p.Lk, dvacc, lane);
    } else {
        store_rows<float, DH>(dKp, p.lddk, krow, p.Lk, dkacc, 1.0f, lane);
        store_rows<float, DH>(dVp, p.lddv, krow, p.Lk, dvacc, 1.0f, lane);
    }
}
