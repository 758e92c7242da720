// Included by attention.hip (inside its anonymous namespace, after the v3 / v4 helpers).
//
// v5 forward, dh = 64, bf16, not causal: the encoder self-attention of reference plankassembly/models.py:60-63,206 (torch
// F.multi_head_attention_forward with a key-padding mask and attention-probability dropout 0.2).
//
// Why another kernel.  v3 / v4 run a wave's 32-key chunk as three strictly serial phases - [K fragments from LDS -> S^T
// MFMAs] -> [softmax + dropout VALU] -> [V^T fragments from LDS -> O^T MFMAs] - and rely on 16 resident waves per CU to
// overlap them; measured, the matrix pipe is 12-20 % busy and 45 % of the wave cycles are parked in s_waitcnt.  Here ONE
// wave's instruction stream is software-pipelined across chunks instead (cdna_hip_programming.md, "Fused attention prefill":
// two MFMA phases per key tile with the softmax VALU placed in the MFMA shadows):
//   * while chunk c's 16 scores per lane are exponentiated / summed / dropped, the four S^T MFMAs of chunk c + 1 and the
//     four O^T MFMAs of chunk c are issued BETWEEN the groups of four scores (hand-placed, fenced with sched_barrier), so the
//     matrix pipe works under the VALU stream of the same wave;
//   * every LDS operand of a chunk body (K fragments of chunk c + 1, the 16 key-hash words, V^T fragments of chunk c) is
//     requested at the top of the body and waited for with counted lgkmcnt just before its first consumer;
//   * K / V tiles travel through a 4-stage ring by LDS-DMA issued THREE tiles ahead, one barrier per 64-key tile;
//   * query rows are pre-multiplied by scale * log2(e) and the S^T accumulator starts at -(reference point), so the MFMA
//     delivers the exponent (no FMA per score); the row maximum is only formed on the steps that move the reference point.
// What bounds it (profiles/r04_attn_issue_bound.txt): instruction ISSUE.  A SIMD issues about one instruction per four cycles
// whatever its type; v3 at the padded benchmark shape executes 12.2 instructions per score and lane (7.9 VALU, 3.0 SALU incl.
// s_nop / s_waitcnt, 0.9 LDS, 0.4 MFMA) with its SIMDs issuing 90 % of the time - the matrix pipe idles because the wave
// slots are spent on everything else.  So this kernel is first of all an instruction diet (no max / FMA / accumulator
// initialisation per score, dropout compares in SGPR pairs without hazard nops, packed row sums, no register copies at the
// chunk boundary) and second as many waves per SIMD as its registers allow (OCC = 3 with a 3-stage ring, or 2 with 4 stages).
// Forward / backward consistency: like attn4_fwd_kernel this kernel scores bf16(q * scale * log2 e) . k, while the
// backward kernels recompute (q . k) * scale * log2 e from the unscaled rows: the two probabilities differ by one extra bf16 rounding
// of q (2^-9 relative per element - the size of the input quantisation), pinned by
// tests/test_kernels_gpu.py::test_attention_backward_probabilities_sum_to_one_per_row.
// Dropout decisions, lse, masks: identical functions to v3 / v4 (pa_device.h drop_keep2; tests/dropout_masks.py).
#define PA_SB() __builtin_amdgcn_sched_barrier(0)
template <int NS> struct L5 {
    // [stage 0 .. NS-1: K tile | V tile] (<= 64 KB: every tile address stays inside the 16-bit DS offset field), then the
    // stages' aux records [64 mask bytes][64 key-hash words (key_slot order)], then 64 B for the mask scan
    static constexpr int NAT = 64 * 128;                    // one 64-row tile of K or V
    static constexpr int STG = 2 * NAT;
    static constexpr int AUX0 = NS * STG, AUXS = 768;
    static constexpr int SHM = AUX0 + NS * AUXS + 64;
};

// Dropout of four probabilities: keep <=> low 32 bits of arow * C[key] >= thr (pa_device.h drop_keep2).  One asm block per four
// scores: the compare results go to four SGPR pairs and each v_cndmask sits three instructions behind its v_cmp (a VALU write of
// an SGPR needs two wait states before a VALU reads it on gfx950; through VCC the compiler pads every cmp / cndmask pair with an
// s_nop, and every instruction - s_nop included - is an issue slot of a kernel that is bound by instruction issue).
template <int G> __device__ __forceinline__ void drop_keep4(f32x16& pr, uint32_t arow, const u32x4& c, uint32_t thr) {
    float p0 = pr[4 * G], p1 = pr[4 * G + 1], p2 = pr[4 * G + 2], p3 = pr[4 * G + 3];
    uint32_t t0, t1, t2, t3;
    uint64_t m0, m1, m2, m3;
    asm("v_mul_u32_u24 %4, %12, %13\n\tv_mul_u32_u24 %5, %12, %14\n\tv_mul_u32_u24 %6, %12, %15\n\tv_mul_u32_u24 %7, %12, %16\n\t"
        "v_cmp_ge_u32 %8, %4, %17\n\tv_cmp_ge_u32 %9, %5, %17\n\tv_cmp_ge_u32 %10, %6, %17\n\tv_cmp_ge_u32 %11, %7, %17\n\t"
        "v_cndmask_b32 %0, 0, %0, %8\n\tv_cndmask_b32 %1, 0, %1, %9\n\tv_cndmask_b32 %2, 0, %2, %10\n\tv_cndmask_b32 %3, 0, %3, %11"
        : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(arow), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "s"(thr));
    pr[4 * G] = p0; pr[4 * G + 1] = p1; pr[4 * G + 2] = p2; pr[4 * G + 3] = p3;
}

template <bool DROP, int NS, int OCC>
__global__ __launch_bounds__(NTH, OCC) void attn5_fwd_kernel(AttnP pin) {
    using L = L5<NS>;
    constexpr int DH = 64, RBN = 128, NAT = L::NAT, STG = L::STG, AUX0 = L::AUX0, AUXS = L::AUXS;
    constexpr int AHEAD = NS - 1;                                      // tiles of DMA in flight ahead of the one being computed
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_, h, b, off_ = 0, len_ = -1;
    int part = 0, nparts = 1, slot = 0;                                // balanced == 2: this block's range of the key tiles
    if (pin.balanced == 2) {
        SplitUnit su;
        if (!decode_unit_split(pin, pin.cu_q, su)) return;
        tile_ = su.tile; h = su.h; b = su.b; off_ = su.off; len_ = su.len; part = su.part; nparts = su.nparts; slot = su.slot;
    }
    else if (pin.balanced) { if (!decode_block_balanced(pin, pin.cu_q, tile_, h, b, off_, len_)) return; }
    else { decode_block((pin.Lq + BOWN - 1) / BOWN, pin.H, pin.B, tile_, h, b); b = dispatch_batch(pin.order, b); }
    int qoff, koff;
    const AttnP p = batch_view(pin, b, qoff, koff, off_, len_);
    const int q0 = tile_ * BOWN;
    if (q0 >= p.Lq) return;
    const bf16* Qp = reinterpret_cast<const bf16*>(p.q) + (size_t)qoff * p.ldq + h * DH;
    const bf16* Kp = reinterpret_cast<const bf16*>(p.k) + (size_t)koff * p.ldk + h * DH;
    const bf16* Vp = reinterpret_cast<const bf16*>(p.v) + (size_t)koff * p.ldv + h * DH;
    const uint8_t* mp = p.kpm ? p.kpm + (size_t)b * pin.Lk : nullptr;
    const int qw0 = q0 + wave * 32, qrow = qw0 + (lane & 31);
    const bool wave_on = qw0 < p.Lq;                                   // wave-uniform

    u32x4 qreg[4];
    load_row_regs<bf16, DH>(qreg, Qp, p.ldq, qrow, p.Lq, lane);
    const float sl = p.scale * LOG2E;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int w = 0; w < 4; ++w) qreg[s][w] = pack_bf16(bf16_lo(qreg[s][w]) * sl, bf16_hi(qreg[s][w]) * sl);

    // K / V tiles by LDS-DMA through ONE buffer descriptor per matrix (the whole sample: rows past its end read as zeros through
    // the hardware range check) and per-lane byte offsets that advance by one tile per issue() - four v_add per tile.  (v3 / v4
    // build a descriptor per DMA instruction, base advanced and size reduced in scalar 64-bit arithmetic: ~18 instructions
    // each, ~100 per 64-key tile, a fifth of this kernel's issue slots when it was measured that way.)
    const TileSrc srcK = tile_src(Kp, p.ldk, p.Lk, DH), srcV = tile_src(Vp, p.ldv, p.Lk, DH);
    auto whole = [](const TileSrc& ts) {
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(ts.base), 0,
                                                 (int)(ts.bytes < 0x7fffffff ? ts.bytes : 0x7fffffff), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsK = whole(srcK), rsV = whole(srcV);
    int vk0 = tile_voff<DH>(p.ldk, tid), vv0 = tile_voff<DH>(p.ldv, tid);
    int vk1 = vk0 + 32 * p.ldk * 2, vv1 = vv0 + 32 * p.ldv * 2;                       // second 32 rows of a tile
    const int stepK = BSTR * p.ldk * 2, stepV = BSTR * p.ldv * 2;
    int t_lo = 0, t_hi = (p.Lk + BSTR - 1) / BSTR;                     // key tiles of this block: all of the element's, or its range
    if (nparts > 1) split_range(t_hi, part, nparts, t_lo, t_hi);
    const int kbase = t_lo * BSTR;
    vk0 += t_lo * stepK; vk1 += t_lo * stepK; vv0 += t_lo * stepV; vv1 += t_lo * stepV;
    LdsBase<DH> lb;
    const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    lb.init(smem_base, lane);
    const uint32_t cbase = smem_base + AUX0 + half * 128;       // this half-wave's 32 key-hash words of a tile (key_slot)

    // issue() calls come in tile order (t = 0, 1, 2, ...): the per-lane offsets are running values
    auto issue = [&](int t, int stage, int kfirst_) {
        char* base = smem + stage * STG + wave * 1024;
        char* aux = smem + AUX0 + stage * AUXS;
        const int k0 = kbase + t * BSTR;
        const bool tile_masked = k0 + BSTR > kfirst_;                  // block-uniform
        uint8_t mb = 0;
        if (tile_masked && tid < BSTR) {                               // (loaded before the DMA is issued: its wait must not cover the tiles)
            const int key = k0 + tid;
            mb = (key >= p.Lk) ? 1 : (mp ? mp[key] : 0);
        }
        using lds_t = __attribute__((address_space(3))) void*;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_t)(base), 16, vk0, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, (lds_t)(base + 4096), 16, vk1, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_t)(base + NAT), 16, vv0, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lds_t)(base + NAT + 4096), 16, vv1, 0, 0, 0);
        vk0 += stepK; vk1 += stepK; vv0 += stepV; vv1 += stepV;
        if (tid < BSTR) {
            if (tile_masked) reinterpret_cast<uint8_t*>(aux)[tid] = mb;
            if (DROP) reinterpret_cast<uint32_t*>(aux + 64)[key_slot(tid)] = drop_key_hash(p.drop_seed, (uint32_t)(k0 + tid));
        }
    };
    const int nt_all = t_hi - t_lo;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t)
        if (t < nt_all) issue(t, t, 0);
    int kfirst = p.Lk, klast = p.Lk;
    if (mp) scan_key_mask(mp, p.Lk, tid, reinterpret_cast<int*>(smem + AUX0 + NS * AUXS), kfirst, klast);
    const int ntiles = min((klast + BSTR - 1) / BSTR, t_hi) - t_lo;
    const int nchunks = (min(klast, t_hi * BSTR) - kbase + 31) / 32;

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const uint32_t arow = DROP ? drop_row_hash(p.drop_seed, (uint32_t)(((size_t)b * p.H + h) * pin.Lq + qrow)) : 0u;
    tile_barrier();                                                    // the first AHEAD tiles have landed

    // S^T of chunk 0.  The score accumulators ping-pong by chunk parity (s_even: chunks 0, 2, ..; s_odd: 1, 3, ..): while one
    // is consumed by the softmax, the MFMAs of the next chunk fill the other - no register copies at the chunk boundary.
    f32x16 s_even, s_odd;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    s_even = zero16; s_odd = zero16;
    // raw S^T (reference point 0) of the 32 keys at LDS offset `koff` (tile + chunk): prologue and slow path
    auto raw_scores = [&](f32x16& dst, auto offc) {
        constexpr int OFF = decltype(offc)::value;
        u32x4 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) PA_DS128(kf[s], lb.nat[s], OFF);
        wait_lds(kf);
        dst = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&kf[0]), *reinterpret_cast<const bf16x8*>(&qreg[0]), zero16, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 4; ++s) mma16B<bf16>(dst, kf[s], qreg[s]);
    };
    if (wave_on && nchunks > 0) raw_scores(s_even, IC<0>{});

    // -(reference point) of this lane's row, added to every score BY THE MATRIX PIPE: one more k-step of the S^T contraction
    // whose A fragment is constant (1, 1, 1, 0, ...) and whose B fragment holds -m as three bf16 terms (24 bits) in the lanes of
    // half-wave 0.  No accumulator initialisation per chunk (the chain starts from the inline constant 0), no FMA per score,
    // and a move of the reference point rewrites four registers.  (A 16-register C operand holding -m was tried first: the
    // compiler re-materialises such a tuple with 16 + 8 moves at every chunk boundary once a branch may redefine it.)
    const u32x4 bias_a = half == 0 ? u32x4{0x3f803f80u, 0x00003f80u, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
    u32x4 bias_b = {0u, 0u, 0u, 0u};
    auto set_reference = [&](float neg_m) {                 // bias_b <- neg_m = b0 + b1 + b2
        const float b0 = bf16_hi(__float_as_uint(neg_m) + 0x8000u) , r1 = neg_m - b0;
        const float b1 = bf16_hi(__float_as_uint(r1) + 0x8000u), r2 = r1 - b1;
        const uint32_t w0 = (__float_as_uint(b0) >> 16) | (__float_as_uint(b1) & 0xffff0000u);
        const uint32_t w1 = pack_bf16(r2, 0.f);
        bias_b = half == 0 ? u32x4{w0, w1, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
    };
    // per lane: the chunk sum above which the slow path is taken - below zero until the row has a reference point (EVERY chunk
    // takes the slow path until then: a first chunk whose scaled scores all sit below -126 exponentiates to 0 and a `> 0` test
    // would never give the row a reference - ADVICE r4), 2^RESCALE_THR after
    float slow_thr = -1.f;
    const float big = __builtin_amdgcn_exp2f(RESCALE_THR);
    const uint32_t thr_s = p.drop_thr;

    // sacc: on entry the scores of chunk c relative to the reference point, on exit dead; snext: on exit the scores of chunk c + 1
    auto chunk = [&](int c, auto stagec, auto ktc, f32x16& sacc, f32x16& snext) {
        constexpr int STAGE = decltype(stagec)::value, KT = decltype(ktc)::value;
        constexpr int SOFF = STAGE * STG;
        constexpr int KNEXT = KT == 0 ? SOFF + 32 * RBN : ((STAGE + 1) % NS) * STG;   // K rows of chunk c + 1
        constexpr int VOFF = SOFF + NAT + KT * 32 * RBN;
        constexpr int HOFF = STAGE * AUXS + 64 + KT * 64;
        const int k0 = kbase + c * 32;
        // ---- every LDS operand of this body, requested up front (in order of first use) ----
        u32x4 kf[4], cq[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) PA_DS128(kf[s], lb.nat[s], KNEXT);
        if (DROP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) PA_DS128(cq[g], cbase, HOFF + g * 16);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                u32x2 x0, x1;
                if (dt == 0) { PA_DSTR(x0, lb.trA, VOFF + u * 16 * RBN); PA_DSTR(x1, lb.trB, VOFF + u * 16 * RBN + 8 * RBN); }
                else { PA_DSTR(x0, lb.trB, VOFF + u * 16 * RBN); PA_DSTR(x1, lb.trA, VOFF + u * 16 * RBN + 8 * RBN); }
                vf[dt * 2 + u][0] = x0[0]; vf[dt * 2 + u][1] = x0[1]; vf[dt * 2 + u][2] = x1[0]; vf[dt * 2 + u][3] = x1[1];
            }
        // ---- masks (edge tiles only) ----
        const bool key_masked = k0 + 32 > kfirst;                      // wave-uniform; the tile's mask bytes exist (issue() wrote them)
        auto apply_mask = [&]() {
            const uint8_t* mk = reinterpret_cast<const uint8_t*>(smem + AUX0 + STAGE * AUXS);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mk + KT * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc[4 * g + e] = ((m4 >> (8 * e)) & 0xffu) ? -INFINITY : sacc[4 * g + e];
            }
        };
        if (key_masked) apply_mask();
        // ---- exponentials of chunk c, in place, with the first two S^T MFMAs of chunk c + 1 ----
        PA_SB();
        if (DROP) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
        else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
        snext = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&kf[0]), *reinterpret_cast<const bf16x8*>(&qreg[0]), zero16, 0, 0, 0);
        PA_SB();
#pragma unroll
        for (int r = 0; r < 8; ++r) sacc[r] = fast_exp2(sacc[r]);
        PA_SB();
        mma16B<bf16>(snext, kf[1], qreg[1]);
        PA_SB();
#pragma unroll
        for (int r = 8; r < 16; ++r) sacc[r] = fast_exp2(sacc[r]);
        // (sums after all the exponentials: a transcendental result needs a wait state before a plain VALU reads it, and two
        // independent packed chains keep the adds from waiting on each other)
        f32x2 ca = f32x2{sacc[0], sacc[1]} + f32x2{sacc[4], sacc[5]}, cb = f32x2{sacc[2], sacc[3]} + f32x2{sacc[6], sacc[7]};
        ca += f32x2{sacc[8], sacc[9]}; cb += f32x2{sacc[10], sacc[11]};
        ca += f32x2{sacc[12], sacc[13]}; cb += f32x2{sacc[14], sacc[15]};
        ca += cb;
        float csum = ca[0] + ca[1];
        // No row maximum on the common path: the reference point only has to move when a probability could leave the range
        // bf16 / the f32 sums are comfortable with, and a chunk whose 16 probabilities sum to at most 2^RESCALE_THR has none
        // above it.  (A row that has not seen an unmasked key yet has no reference point: it takes the slow path on its first.)
        if (__any(csum > slow_thr)) {
            // slow path (a row's first keys; afterwards only when its scores outgrow the reference by 2^RESCALE_THR): the chunk's
            // raw scores are formed again - the exponentials above overwrote them - and the reference point becomes the maximum
            raw_scores(sacc, IC<SOFF + KT * 32 * RBN>{});
            if (key_masked) apply_mask();
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(mx, fmaxf(sacc[r], sacc[r + 1]));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float ms = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - ms);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            csum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] = fast_exp2(sacc[r] - ms); csum += sacc[r]; }
            m_run = m_new;
            slow_thr = (m_new == -INFINITY) ? -1.f : big;
            set_reference(-ms);                                        // chunk c + 1 onwards: its bias k-step is issued below
        }
        l_run += csum;
        // ---- dropout + pack of chunk c between the remaining MFMAs ----
        u32x4 pb[2];
        auto drop4 = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (DROP) drop_keep4<g>(sacc, arow, cq[g], thr_s);
        };
        auto pack = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
#pragma unroll
            for (int w = 0; w < 4; ++w) pb[u][w] = pack_bf16(sacc[8 * u + 2 * w], sacc[8 * u + 2 * w + 1]);
        };
        PA_SB();
        mma16B<bf16>(snext, kf[2], qreg[2]);
        PA_SB();
        if (DROP) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cq[0]), "+v"(cq[1]), "+v"(cq[2]), "+v"(cq[3]));
        drop4(IC<0>{}); drop4(IC<1>{});
        pack(IC<0>{});
        PA_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vf[0]), "+v"(vf[1]), "+v"(vf[2]), "+v"(vf[3]));
        mma16B<bf16>(oacc[0], vf[0], pb[0]);
        mma16B<bf16>(snext, kf[3], qreg[3]);
        PA_SB();
        drop4(IC<2>{});
        PA_SB();
        mma16B<bf16>(oacc[1], vf[2], pb[0]);
        PA_SB();
        drop4(IC<3>{});
        pack(IC<1>{});
        PA_SB();
        mma16B<bf16>(oacc[0], vf[1], pb[1]);
        mma16B<bf16>(snext, bias_a, bias_b);                           // S^T(c + 1) -= reference point (decided above, at the latest)
        mma16B<bf16>(oacc[1], vf[3], pb[1]);
        PA_SB();
    };
    auto tile = [&](int t, auto stagec) {
        constexpr int STAGE = decltype(stagec)::value;
        if (t > 0) {
            // Tile t + 1 must have landed (chunk 2t + 1 reads its first K rows) and every wave must have left tile t - 1, whose
            // stage the next DMA overwrites.  In flight, oldest first: tiles t + 1 .. t + AHEAD - 1 (four 1-KB pieces per wave
            // each): the counted wait leaves the younger ones travelling across the barrier.
            // (s_barrier by hand: __syncthreads() makes hipcc drain vmcnt to 0 first.  lgkmcnt(0): the aux words this wave wrote.)
            if (AHEAD >= 3 && t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (t + AHEAD < ntiles) issue(t + AHEAD, (STAGE + AHEAD) % NS, kfirst);
        if (!wave_on) return;
        chunk(2 * t, stagec, IC<0>{}, s_even, s_odd);
        if (2 * t + 1 < nchunks) chunk(2 * t + 1, stagec, IC<1>{}, s_odd, s_even);
    };
    for (int t = 0; t < ntiles; t += NS) {
        tile(t, IC<0>{});
        if (t + 1 < ntiles) tile(t + 1, IC<1>{});
        if (t + 2 < ntiles) tile(t + 2, IC<2>{});
        if (NS > 3 && t + 3 < ntiles) tile(t + 3, IC<(NS > 3 ? 3 : 0)>{});
    }
    float l_tot = l_run + __shfl_xor(l_run, 32);
    if (nparts > 1) {
        // range block: publish (O^T, m, l); the last of the tile's range blocks to arrive rescales all of them to the common
        // reference point and stores (the same merge attn4_fwd_kernel<., 2> does through LDS)
        char* const pbase = pin.sp_part + (size_t)slot * pin.sp_pmax * SP_BYTES;
        const SplitOut so(pbase + (size_t)part * SP_BYTES, SP_BYTES);
#pragma unroll
        for (int j = 0; j < 8; ++j) so.put(j, NTH, f32x4{oacc[j >> 2][4 * (j & 3)], oacc[j >> 2][4 * (j & 3) + 1], oacc[j >> 2][4 * (j & 3) + 2], oacc[j >> 2][4 * (j & 3) + 3]});
        so.put(8, NTH, f32x4{m_run, l_tot, 0.f, 0.f});
        if (!split_arrive(pin.sp_tick + slot, nparts, reinterpret_cast<int*>(smem + AUX0 + NS * AUXS))) return;
        for (int pp = 0; pp < nparts; ++pp) {
            if (pp == part) continue;
            const char* ob = pbase + (size_t)pp * SP_BYTES;
            const f32x4 ml = split_get(ob, 8, NTH, SP_BYTES);
            const float m_new = fmaxf(m_run, ml[0]);
            const float ms = (m_new == -INFINITY) ? 0.f : m_new;
            const float a0 = fast_exp2(m_run - ms), a1 = fast_exp2(ml[0] - ms);
            l_tot = l_tot * a0 + ml[1] * a1;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 x = split_get(ob, j, NTH, SP_BYTES);
#pragma unroll
                for (int e = 0; e < 4; ++e) oacc[j >> 2][4 * (j & 3) + e] = oacc[j >> 2][4 * (j & 3) + e] * a0 + x[e] * a1;
            }
            m_run = m_new;
        }
    }
    const float inv = l_tot > 0.f ? (DROP ? p.drop_scale : 1.0f) / l_tot : 0.f;
    bf16* Op = reinterpret_cast<bf16*>(p.o) + (size_t)qoff * p.ldo + h * DH;
    store_rows<bf16, DH>(Op, p.ldo, qrow, p.Lq, oacc, inv, lane);
    if (half == 0 && qrow < p.Lq && p.lse)
        p.lse[((size_t)b * p.H + h) * pin.Lq + qrow] = l_tot > 0.f ? (m_run + log2f(l_tot)) * LN2 : 0.f;
}
