// sunscreen_amd/csrc/behzcore.hpp -- per-coefficient BEHZ base conversions (SEAL RNSTool), shared by the
// coefficient-parallel kernels of kernels.hip and the head / tail kernels of kernels_split.hip so that both
// pipelines execute the identical arithmetic.
#pragma once
#include "devarith.hpp"

namespace hipbfv {

// fastbconv_m_tilde + sm_mrq for one coefficient: x[i] = residue mod q_i  ->  out[j] = residue mod Bsk_j
// (Evaluator_Multiply steps 1-2, seal_fhe/src/evaluator_base.rs:198-212 -> SEAL bfv_multiply).
template <int KMAX>
__device__ __forceinline__ void behz_extend_coeff(const DevCtx* __restrict__ ctx, const u64 (&x)[KMAX], u64 (&out)[KMAX + 2]) {
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK;
  u64 y[KMAX];
  u32 rm = 0;
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    if ((u32)i < K) {
      y[i] = mul_shoup(x[i], ctx->ext_scale[i], ctx->mod[i].q);
      rm += (u32)y[i] * ctx->q_to_mtilde[i];
    }
  }
  rm *= ctx->neg_inv_q_mod_mtilde;  // r_mtilde = -x/q mod 2^32
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) {
    if ((u32)j < S) {
      const DevMod& pm = ctx->mod[KK + j];
      u128 acc = 0;
#pragma unroll
      for (int i = 0; i < KMAX; i++)
        if ((u32)i < K) acc += (u128)y[i] * ctx->q_to_bsk[j][i];
      u64 rc = rm;
      if (rm >= 0x80000000u) rc += pm.q - 0x100000000ull;  // centred representative
      acc += (u128)rc * ctx->q_mod_bsk[j];
      out[j] = mul_shoup(reduce128_fast(acc, pm), ctx->inv_mtilde_mod_bsk[j], pm.q);
    }
  }
}

// fast_floor (q u Bsk -> Bsk) then Shenoy-Kumaresan (Bsk -> q) for one coefficient.
// y[i]  = x * t * (q/q_i)^{-1} mod q_i (the scaled inverse transform delivers exactly this),
// xb[j] = x * t mod Bsk_j;  out[i] = floor(t*x/q) mod q_i with SEAL's alpha_sk correction.
template <int KMAX>
__device__ __forceinline__ void behz_floor_sk_coeff(const DevCtx* __restrict__ ctx, const u64 (&y)[KMAX], const u64 (&xb)[KMAX + 2],
                                                    u64 (&out)[KMAX]) {
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, nB = ctx->nB;
  u64 yb[KMAX + 1];
  u64 fl_msk = 0;
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) {
    if ((u32)j < S) {
      const DevMod& pm = ctx->mod[KK + j];
      u128 acc = 0;
#pragma unroll
      for (int i = 0; i < KMAX; i++)
        if ((u32)i < K) acc += (u128)y[i] * ctx->q_to_bsk[j][i];
      const u64 conv = reduce128_fast(acc, pm);
      const u64 fl = mul_shoup(xb[j] + pm.q - conv, ctx->inv_q_mod_bsk[j], pm.q);
      if ((u32)j < nB) {
        if (j < KMAX + 1) yb[j < KMAX + 1 ? j : 0] = mul_shoup(fl, ctx->inv_punct_B[j], pm.q);
      } else {
        fl_msk = fl;
      }
    }
  }
  const DevMod& msk = ctx->mod[KK + nB];
  u128 acc = 0;
#pragma unroll
  for (int j = 0; j < KMAX + 1; j++)
    if ((u32)j < nB) acc += (u128)yb[j] * ctx->B_to_msk[j];
  const u64 alpha = mul_shoup(reduce128_fast(acc, msk) + msk.q - fl_msk, ctx->inv_B_mod_msk, msk.q);
  const bool neg = alpha > (msk.q >> 1);
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    if ((u32)i < K) {
      const DevMod& qm = ctx->mod[i];
      u128 a = 0;
#pragma unroll
      for (int j = 0; j < KMAX + 1; j++)
        if ((u32)j < nB) a += (u128)yb[j] * ctx->B_to_q[i][j];
      if (neg)
        a += (u128)(msk.q - alpha) * ctx->B_mod_q[i];
      else
        a += (u128)alpha * (qm.q - ctx->B_mod_q[i]);
      out[i] = reduce128(a, qm);
    }
  }
}

}  // namespace hipbfv
