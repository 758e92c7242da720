// Eight-wave big-tile ring kernel (included by gemm.hip after gemm3w_kernel; uses GemmP, Unit, static_for, mma16B).
// OPT-IN (PA_GEMM_BIG=1): measured level with the four-wave kernels at this model's K = 512 ... 1536, never ahead
// (profiles/r04_step_floor_probes.txt, sections 6-8).
//
// Why it was built: the K loop of the four-wave 128 x 128 kernels takes ~1 130 cycles per 64-deep K tile against 512 cycles of
// MFMA per wave, and exactly the same 1 130 cycles when the whole activation operand is one L2-resident row (tools/gemm_trace.py
// ALIAS=1) - the loop is not waiting for memory latency.  With one wave per SIMD nothing else can issue while that wave walks its
// ~100 non-MFMA instructions per item; here a block is EIGHT waves - two per SIMD, 256 registers each - on a (32 FM WM) x (32 FN WN)
// tile, and an operand row staged in LDS meets twice as many rows of the other operand:
//   <WM 4, WN 2, FM 2, FN 4>  256 x 256 tile, wave tile 64 x 128: 8 MFMAs per 6 fragment reads, 64 KB per K tile for 4x the flops of
//                             a 128 x 128 tile (32 KB).  K loop 2 853 cycles per item = 0.72 of the MFMA peak inside the loop
//   <WM 4, WN 2, FM 2, FN 2>  256 x 128 tile, wave tile 64 x 64 (N = d_model: 256-wide tiles would leave 3/4 of the CUs idle)
//   (<WM 2, WN 4, FM 4, FN 2>, the same 256 x 256 tile with 128 x 64 wave tiles, first streamed its K tiles three times slower -
//   6 732 cycles per item: hipcc had left the epilogue's fragment loops rolled, the accumulators were indexed dynamically and 28
//   bytes per lane went to scratch.  With the loops as static_for it ties with <4, 2, 2, 4>: 25.7 vs 25.3 us at 8704 x 1536 x 512.)
//   Four waves stacked along M with whole-width wave tiles (tools/ubench/gemm8_lat.hip cfg 6 / 7: <4, 1, 2, 7> 256 x 224 and
//   <4, 1, 2, 5> 256 x 160, the tile shapes hipBLASLt picks so that M ~ 8 700 is ONE round of 238 blocks; unit order plain_order 2)
//   do not pay either: 256 x 160 21.5 us at 8704 x 1024 x 512 (default kernels 22.6, hipBLASLt 17.3), 256 x 224 37 us (84 bytes of
//   scratch per lane, 14 MFMAs per k-step on one wave per SIMD).  profiles/r04_step_floor_probes.txt section 12.
// What was learned: a CU's direct-to-LDS DMA is processed at ~36 cycles per 1 KB instruction (27-29 B/clk) whatever the ring depth,
// so only flops per DMA byte help the loop - and at K = 512 the 256 x 256 tile's epilogue (10.4 k cycles: eight staged 32 x 32 passes
// per wave) and first-item latency (4.2 k) cost what its K loop (22.8 k) saves.
// bf16, both operands k-contiguous, K % 64 == 0, no split-K.  An ITEM is one BK-deep K tile (BK = 32: 64-byte LDS rows, or 64);
// an NSTG-stage LDS ring of items is filled by direct-to-LDS DMA and the items of consecutive units form one stream, as in
// gemm3w_kernel.  ONE barrier per item: before the MFMAs of an item's last k-step every wave waits (counted vmcnt) for its
// share of the next item, the block meets at the barrier - now the next item has landed for everybody AND everybody has read
// the last fragments of the current item - and the current item's own stage is refilled with the item NSTG ahead, so NSTG - 1
// items are in flight behind the one being multiplied.
// Timing ablations for tools/ubench/gemm8_lat.hip (-DPA_G8_ABL=<bits>, wrong results): 1 no MFMA, 2 no DMA, 4 no fragment reads,
// 8 no output stores, 16 non-temporal stores, 32 write-through stores, 64 return at entry (the launch alone).
#ifdef PA_GEMM8_TRACE
// debug build only (tools/ubench/gemm8_lat.hip -DPA_GEMM8_TRACE): per-block cycle stamps
__device__ unsigned long long pa_gemm8_trace[512 * 8];
#define PA_TR8(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) pa_gemm8_trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PA_TR8(i) do { } while (0)
#endif
template <int WM, int WN, int FM, int FN, int BK, int NSTG, int WPS = 1>     // WPS: waves per SIMD the register budget must allow
__global__ __launch_bounds__(64 * WM * WN, WPS) void gemm8_kernel(GemmP p) {
    using T = bf16;
    PA_TR8(0);
#if defined(PA_G8_ABL) && (PA_G8_ABL & 64)             // probe: the launch alone (same registers / LDS, no work)
    if (p.alpha != 123.f) return;
#endif
    constexpr int NT8 = 64 * WM * WN;
    constexpr int TBM = 32 * FM * WM, TBN = 32 * FN * WN;
    constexpr int RB = BK * 2, NCH = RB / 16, RPB = 256 / RB, KS = BK / 16;
    static_assert(BK == 32 || BK == 64, "K tile depth");
    constexpr int A_BYTES = TBM * RB, B_BYTES = TBN * RB;
    constexpr int NLA = TBM * NCH / NT8, NLB = TBN * NCH / NT8, NLD = NLA + NLB;      // 16-byte DMA chunks per thread per item
    constexpr int STAGE = A_BYTES + B_BYTES, EPI = 32 * 32 * 4, NW = WM * WN;
    static_assert(NSTG * STAGE + NW * EPI <= 160 * 1024, "LDS budget");
    static_assert(NSTG >= 2 && NSTG <= 5, "ring depth");
    static_assert(TBM * NCH % NT8 == 0 && TBN * NCH % NT8 == 0, "whole DMA chunks per thread");
    __shared__ __attribute__((aligned(256))) char smem[NSTG * STAGE + NW * EPI];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5;
    constexpr int esz = 2;
    f32x16 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint32_t offA[NLA], offB[NLB];
    const char* kA = nullptr; const char* kB = nullptr;
    auto setup = [&](const Unit& un) {
        kA = reinterpret_cast<const char*>(p.A) + (size_t)un.b * p.sA * esz;
        kB = reinterpret_cast<const char*>(p.B) + (size_t)un.b * p.sB * esz;
        const int m0 = un.tile_m * TBM, n0 = un.tile_n * TBN;
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int pidx = tid + i * NT8, row = pidx / NCH, ch = ((pidx % NCH) ^ (row / RPB)) & (NCH - 1);
            offA[i] = (uint32_t)min(m0 + row, p.M - 1) * (uint32_t)(p.lda * esz) + ch * 16;
        }
#pragma unroll
        for (int i = 0; i < NLB; ++i) {
            const int pidx = tid + i * NT8, row = pidx / NCH, ch = ((pidx % NCH) ^ (row / RPB)) & (NCH - 1);
            offB[i] = (uint32_t)min(n0 + row, p.N - 1) * (uint32_t)(p.ldb * esz) + ch * 16;
        }
    };
    // chunk c of an item's DMA: c < NLA -> A chunk c, else B chunk c - NLA
    auto fetch_c = [&](int stage, auto C_) {
        constexpr int c = decltype(C_)::value;
        char* lx = smem + stage * STAGE + (c < NLA ? 0 : A_BYTES);
        constexpr int i = c < NLA ? c : c - NLA;
        const char* src = (c < NLA) ? kA + offA[i] : kB + offB[i];
#if !(defined(PA_G8_ABL) && (PA_G8_ABL & 2))          // timing ablation (wrong results): no DMA
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
            (__attribute__((address_space(3))) void*)(lx + (i * NT8 + wave * 64) * 16), 16, 0, 0);
#else
        (void)src; (void)lx;
#endif
    };
    auto fetch_done = [&]() { kA += BK * esz; kB += BK * esz; };
    auto fetch_range = [&](int stage, auto LO_, auto HI_) {
        constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value;
        static_for<LO, HI>([&](auto I_) { fetch_c(stage, I_); });
    };
    auto fetch = [&](int stage) {
        fetch_range(stage, std::integral_constant<int, 0>{}, std::integral_constant<int, NLD>{});
        fetch_done();
    };

    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int sw = ((lane & 31) / RPB) & (NCH - 1);
    uint32_t xs[KS];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_) xs[s_] = (uint32_t)((((2 * s_ + half) ^ sw) & (NCH - 1)) << 4);
    const uint32_t fa_off = (wm * 32 * FM + (lane & 31)) * RB, fb_off = (wn * 32 * FN + (lane & 31)) * RB + A_BYTES;
#define PA_RD128P(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
    // fragment W of k-step s: W < FM -> A row group W, else B row group W - FM
    auto rd1 = [&](u32x4 (&f)[FM + FN], uint32_t st, int s_, auto W_) {
        constexpr int W = decltype(W_)::value;
#if !(defined(PA_G8_ABL) && (PA_G8_ABL & 4))          // timing ablation: no fragment reads
        if constexpr (W < FM) { const uint32_t ad = st + fa_off + xs[s_]; PA_RD128P(f[W], ad, W * 32 * RB); }
        else { const uint32_t ad = st + fb_off + xs[s_]; PA_RD128P(f[W], ad, (W - FM) * 32 * RB); }
#else
        (void)st; (void)s_;
#endif
    };
    auto frag = [&](u32x4 (&f)[FM + FN], uint32_t st, int s_) {
        static_for<0, FM + FN>([&](auto I_) { rd1(f, st, s_, I_); });
    };
    auto wait_frag = [&](u32x4 (&f)[FM + FN]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < FM + FN; ++i) asm volatile("" : "+v"(f[i]));
    };

    // ---- epilogue: one 32 x 32 accumulator tile at a time through a 4 KB per-wave staging slice (as gemm3w_kernel) ------
    auto epilogue = [&](const Unit& un) {
        char* stage = smem + NSTG * STAGE + wave * EPI;
        const size_t cbase = (size_t)un.b * p.sC;
        const int chunk = lane & 7, rsub = lane >> 3;
        const bool out_f32 = p.out_dtype == PA_F32;
        const bool has_bias = p.bias != nullptr, has_aux = p.aux != nullptr, has_res = p.R != nullptr, has_drop = p.drop_thr != 0;
        constexpr int NIT = 4;
        // (static_for, not #pragma unroll: with FN = 7 hipcc leaves the loop rolled and the accumulators go to scratch)
        static_for<0, FN>([&](auto FN_) {
            constexpr int fn = decltype(FN_)::value;
            const int nw = un.tile_n * TBN + (wn * FN + fn) * 32;
            const int n = nw + chunk * 4;
            const bool fast = p.vec_ok && (nw + 32 <= p.N);
            f32x4 bias = {0.f, 0.f, 0.f, 0.f};
            if (has_bias) {
                const float* bp = p.bias + (size_t)un.b * p.sBias;
                if (fast) bias = *reinterpret_cast<const f32x4*>(bp + n);
                else { for (int e = 0; e < 4; ++e) if (n + e < p.N) bias[e] = bp[n + e]; }
            }
            static_for<0, FM>([&](auto FM_) {
                constexpr int fm = decltype(FM_)::value;
                const int mw = un.tile_m * TBM + (wm * FM + fm) * 32;
                if (mw >= p.M || nw >= p.N) {                          // (wave-uniform) sub-tile entirely outside the matrix
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;
                    return;
                }
                {
                    const int lrow = lane & 31;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[fn][fm][4 * g4 + e];
                        const int ch = 2 * g4 + half;
                        *reinterpret_cast<f32x4*>(stage + lrow * 128 + ((ch ^ (lrow & 7)) << 4)) = v;
                    }
                }
                f32x4 x[NIT];
                const int mp = mw + rsub;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int lr = it * 8 + rsub;
                    x[it] = *reinterpret_cast<const f32x4*>(stage + lr * 128 + ((chunk ^ (lr & 7)) << 4));
                }
                if (fast) {
                    // residual / gate rows stay packed (bf16 pairs) until they are used: 2 registers per row instead of 4
                    u32x2 res[NIT], gate[NIT];
                    f32x4 resf[NIT];
                    auto widen = [](const u32x2& u) { f32x4 r; r[0] = bf16_lo(u[0]); r[1] = bf16_hi(u[0]); r[2] = bf16_lo(u[1]); r[3] = bf16_hi(u[1]); return r; };
                    if (has_res) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const size_t ro = (size_t)un.b * p.sR + (size_t)min(mp + it * 8, p.M - 1) * p.ldr + n;
                            if (out_f32) resf[it] = ld4<float>(reinterpret_cast<const float*>(p.R) + ro);
                            else res[it] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16*>(p.R) + ro);
                        }
                    }
                    if (has_aux) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const size_t ao = (size_t)un.b * p.sAux + (size_t)min(mp + it * 8, p.M - 1) * p.ldaux + n;
                            gate[it] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const T*>(p.aux) + ao);
                        }
                    }
#pragma unroll
                    for (int it = 0; it < NIT; ++it)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x[it][e] = x[it][e] * p.alpha + bias[e];
                    if (p.relu) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it)
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[it][e] = fmaxf(x[it][e], 0.f);
                    }
                    if (has_aux) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const f32x4 g = widen(gate[it]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[it][e] = g[e] > 0.f ? x[it][e] * p.aux_scale : 0.f;
                        }
                    }
                    if (has_drop) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                x[it][e] = drop_keep_rc(p.drop_seed, (uint32_t)(un.b * p.M + mp + it * 8), (uint32_t)(n + e), p.drop_thr) ? x[it][e] * p.drop_scale : 0.f;
                            }
                    }
                    if (has_res) {
#pragma unroll
                        for (int it = 0; it < NIT; ++it) {
                            const f32x4 rr = out_f32 ? resf[it] : widen(res[it]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) x[it][e] += rr[e];
                        }
                    }
                    {
                        const bool interior = mw + 32 <= p.M;                          // wave-uniform
                        const size_t co0 = cbase + (size_t)mp * p.ldc + n, rstep = (size_t)8 * p.ldc;
                        if (out_f32) {
                            float* cp = reinterpret_cast<float*>(p.C) + co0;
                            if (interior) {
#pragma unroll
                                for (int it = 0; it < NIT; ++it) *reinterpret_cast<f32x4*>(cp + it * rstep) = x[it];
                            } else {
#pragma unroll
                                for (int it = 0; it < NIT; ++it) if (mp + it * 8 < p.M) *reinterpret_cast<f32x4*>(cp + it * rstep) = x[it];
                            }
                        } else {
                            bf16* cp = reinterpret_cast<bf16*>(p.C) + co0;
                            if (interior) {
#if defined(PA_G8_ABL) && (PA_G8_ABL & 8)             // timing ablation: no output stores
                                if (p.alpha == 123.f)
#endif
#pragma unroll
                                for (int it = 0; it < NIT; ++it) {
#if defined(PA_G8_ABL) && (PA_G8_ABL & 16)            // probe: non-temporal stores
                                    u32x2 u_; u_[0] = pack_bf16(x[it][0], x[it][1]); u_[1] = pack_bf16(x[it][2], x[it][3]);
                                    __builtin_nontemporal_store(u_, reinterpret_cast<u32x2*>(cp + it * rstep));
#elif defined(PA_G8_ABL) && (PA_G8_ABL & 32)          // probe: write-through (agent-scope relaxed atomic) stores
                                    u32x2 u_; u_[0] = pack_bf16(x[it][0], x[it][1]); u_[1] = pack_bf16(x[it][2], x[it][3]);
                                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(cp + it * rstep), ((unsigned long long)u_[1] << 32) | u_[0],
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                                    st4<bf16>(cp + it * rstep, x[it]);
#endif
                                }
                            } else {
#pragma unroll
                                for (int it = 0; it < NIT; ++it) if (mp + it * 8 < p.M) st4<bf16>(cp + it * rstep, x[it]);
                            }
                        }
                    }
                } else {
#pragma unroll 1
                    for (int it = 0; it < NIT; ++it) {
                        const int m = mp + it * 8;
                        if (m >= p.M) continue;
                        for (int e = 0; e < 4; ++e) {
                            if (n + e >= p.N) continue;
                            float y = x[it][e] * p.alpha + bias[e];
                            if (p.relu) y = fmaxf(y, 0.f);
                            if (has_aux) {
                                const float g = ld1(reinterpret_cast<const T*>(p.aux) + (size_t)un.b * p.sAux + (size_t)m * p.ldaux + n + e);
                                y = g > 0.f ? y * p.aux_scale : 0.f;
                            }
                            if (has_drop) {
                                y = drop_keep_rc(p.drop_seed, (uint32_t)(un.b * p.M + m), (uint32_t)(n + e), p.drop_thr) ? y * p.drop_scale : 0.f;
                            }
                            if (has_res) {
                                const size_t ro = (size_t)un.b * p.sR + (size_t)m * p.ldr + n + e;
                                y += out_f32 ? reinterpret_cast<const float*>(p.R)[ro] : (float)reinterpret_cast<const bf16*>(p.R)[ro];
                            }
                            const size_t co = cbase + (size_t)m * p.ldc + n + e;
                            if (out_f32) reinterpret_cast<float*>(p.C)[co] = y;
                            else reinterpret_cast<bf16*>(p.C)[co] = (bf16)y;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[fn][fm][r] = 0.f;
            });
        });
    };

    // ---- item stream: units in XCD-interleaved order over the row tiles; DMA cursor up to NSTG items ahead ------------------
    const int ustride = gridDim.x, nt = p.K / BK;
    auto unit_of = [&](int u, Unit& un) -> bool {
        if (p.plain_order == 2) {
            // one-round order: block u sits on XCD u & 7; XCD x takes the x-th eighth of the row-major tile list, so a row panel of A
            // is fetched by one or two L2s and no row-tile padding pushes the launch past 256 units (units = 8 * per * batch)
            const int tiles = p.tiles_m * p.tiles_n, per = (tiles + 7) >> 3, per_b = per << 3;
            un.b = u / per_b; un.z = un.b;
            const int r = u - un.b * per_b, idx = (r & 7) * per + (r >> 3);
            un.tile_m = idx / p.tiles_n; un.tile_n = idx - un.tile_m * p.tiles_n;
            un.t_begin = 0; un.t_end = nt;
            return idx < tiles;
        }
        const int per_b = p.tiles_m_pad * p.tiles_n;
        un.b = u / per_b; un.z = un.b;
        const int r = u - un.b * per_b;
        if (p.plain_order) { un.tile_m = r / p.tiles_n; un.tile_n = r - un.tile_m * p.tiles_n; }
        else {
            const int xcd = r & 7, i = r >> 3, q = i / p.tiles_n;
            un.tile_n = i - q * p.tiles_n; un.tile_m = q * 8 + xcd;
        }
        un.t_begin = 0; un.t_end = nt;
        return un.tile_m < p.tiles_m;
    };
    auto seek = [&](int u, Unit& un) -> int { while (u < p.units && !unit_of(u, un)) u += ustride; return u; };
    Unit cun, dun;
    int cc_u = seek(blockIdx.x, cun), cc_t = 0;
    if (cc_u >= p.units) return;
    int cd_u = cc_u, cd_t = 0;
    dun = cun;
    setup(dun);
    int sd = 0, pending = 0;                    // sd: stage the next DMA item goes to; pending: items issued and not yet multiplied
    auto issue = [&]() {
        fetch(sd);
        sd = sd + 1 == NSTG ? 0 : sd + 1;
        ++pending;
        if (++cd_t >= nt) { cd_u = seek(cd_u + ustride, dun); cd_t = 0; if (cd_u < p.units) setup(dun); }
    };
    // wait until at most `younger` whole items (NLD DMA instructions each) are still outstanding
    auto wait_items = [&](int younger) {
        if (NSTG >= 5 && younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * NLD) : "memory");
        else if (NSTG >= 4 && younger == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NLD) : "memory");
        else if (NSTG >= 3 && younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    PA_TR8(1);
#pragma unroll 1
    for (int k = 0; k < NSTG; ++k) if (cd_u < p.units) issue();
    wait_items(pending - 1);
    __builtin_amdgcn_s_barrier();
    PA_TR8(2);
    int sc = 0;
    u32x4 F0[FM + FN], F1[FM + FN];
    frag(F0, lds0, 0);
    // the MFMAs of one k-step.  The scheduling fence keeps them in front of whatever follows in program order: without it
    // hipcc sinks most of them below the next wait_frag (an asm volatile it may not reorder against other asm, but MFMA
    // builtins are free to move), so the wave would sit in s_waitcnt lgkmcnt(0) with an empty matrix pipe.
    auto mma_all = [&](u32x4 (&f)[FM + FN]) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
#if !(defined(PA_G8_ABL) && (PA_G8_ABL & 1))          // timing ablation: no MFMA
            for (int fm = 0; fm < FM; ++fm) mma16B<T>(acc[fn][fm], f[FM + fn], f[fm]);
#else
            for (int fm = 0; fm < FM; ++fm) asm volatile("" : "+v"(acc[fn][fm]) : "v"(f[FM + fn]), "v"(f[fm]));
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    while (true) {
        const uint32_t st = lds0 + sc * STAGE;
        // k-steps 0 .. KS-2: multiply the fragments on hand while the next k-step's are read
        if constexpr (KS == 4) {
            wait_frag(F0); frag(F1, st, 1); mma_all(F0);
            wait_frag(F1); frag(F0, st, 2); mma_all(F1);
            wait_frag(F0); frag(F1, st, 3); mma_all(F0);
        } else {
            wait_frag(F0); frag(F1, st, 1); mma_all(F0);
        }
        wait_frag(F1);                          // every LDS read of this item has returned
        const bool has_next = pending >= 2;
        const bool unit_end = cc_t + 1 >= nt;
        const uint32_t stn = lds0 + (sc + 1 == NSTG ? 0 : sc + 1) * STAGE;
        if (has_next) {
            wait_items(pending - 2);            // my share of the next item has landed
            __builtin_amdgcn_s_barrier();       // ... everybody's has, and everybody is done reading this item's stage
            if (cd_u < p.units) issue();        // refill it (sd == sc here: the ring is full whenever a next item exists)
            if (!unit_end) frag(F0, stn, 0);    // (at a unit boundary the epilogue comes first: its registers overlay the fragments')
        }
        mma_all(F1);
        sc = sc + 1 == NSTG ? 0 : sc + 1;
        --pending;
        if (unit_end) {
            PA_TR8(3);
            epilogue(cun);
            PA_TR8(4);
            cc_t = 0;
            if (has_next) { cc_u = seek(cc_u + ustride, cun); frag(F0, stn, 0); }
        } else ++cc_t;
        if (!has_next) break;
    }
#undef PA_RD128P
}
