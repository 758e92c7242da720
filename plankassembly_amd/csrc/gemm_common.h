// Pieces shared by the GEMM kernels of gemm.hip and gemm8.h (and by the stand-alone micro-benchmarks under tools/ubench that
// compile ONE kernel family without the rest of gemm.hip): the kernel argument block, the work-unit record, a compile-time loop.
#pragma once
#include <type_traits>
#include <utility>
#include "pa_device.h"
#include "../../include/plank_hip.h"

namespace {

struct GemmP {
    const void* A; const void* B; void* C;
    const float* bias; const void* R; const void* aux;
    int M, N, K;
    int lda, ldb, ldc, ldr, ldaux;
    long long sA, sB, sC, sR, sAux, sBias;
    int batch;
    float alpha; int relu; float aux_scale;
    uint32_t drop_thr; float drop_scale; uint32_t drop_seed;
    int out_dtype;
    int splitk, tiles_per_slice;   // split-K: C is the f32 slab workspace, plain store
    int tiles_m, tiles_n, tiles_m_pad, units;   // tiles_m_pad == tiles_m: plain row-major unit order (no XCD interleave)
    int vec_ok;                    // epilogue may use 4-element vector accesses on C / R / aux / bias
    int plain_order;               // units enumerate (tile_m, tile_n) row-major instead of the XCD interleave
    int dbg;                       // ablation bits (PA_GEMM_DBG): 1 no MFMA, 2 no ds_read, 4 no loads, 8 no epilogue
    // "A = LayerNorm(Z)" folded into the product (gemm3s_kernel only, pa_gemm_norm_a): A holds the raw rows Z, B the weight
    // pre-multiplied by gamma, ln_u[n] = sum_k B[n][k], bias[n] = b[n] + sum_k W[n][k] beta[k]; the kernel computes the row
    // statistics itself and writes  rstd_m (acc - mean_m u_n) + bias_n.  ln_y: where to materialise LayerNorm(Z) (or null).
    const float* ln_u; const float* ln_gamma; const float* ln_beta; void* ln_y; int ldy; float ln_eps;
    // f32 residual stream beside bf16 matrix operands (gemm_skinny_kernel only; the bf16 greedy-decode step): c_lp = bf16 copy of
    // an f32 output (the next Linear's MFMA operand); ln_zf = the f32 rows Z behind the bf16 operand A = bf16(Z): the row
    // statistics and the materialised LayerNorm(Z) come from them, ln_y_f32: ln_y is f32
    void* c_lp; int ldc_lp;
    const float* ln_zf; int ldzf; int ln_y_f32;
};

struct Unit { int tile_m, tile_n, b, z, t_begin, t_end; };

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [LO, HI)
template <int LO, int... I, typename F>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, LO + I>{}), ...); }
template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<LO>(f, std::make_integer_sequence<int, (HI > LO ? HI - LO : 0)>{}); }


}  // namespace
