// sunscreen_amd/csrc/devctx.hpp -- plain-old-data tables shared by host code and HIP kernels.
//
// One DevCtx describes one BFV context (SEALContext_Create: seal_fhe/src/context.rs:63-80):
// the key-level primes (data primes then the special prime), the BEHZ auxiliary base
// Bsk = B u {m_sk}, and every per-modulus constant the kernels need.  It lives in device
// global memory; kernels receive a `const DevCtx*` (uniform loads -> SGPRs).
#pragma once
#include <cstddef>
#include <cstdint>

namespace hipbfv {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int kMaxKey = 17;   // key-level primes (n=32768 default: 16)
constexpr int kMaxBsk = 18;   // |B| + 1
constexpr int kMaxMod = kMaxKey + kMaxBsk + 1;  // + the plain modulus (BatchEncoder transforms)

// A constant multiplicand with its Shoup quotient floor(w * 2^64 / q).
struct MulOp {
  u64 w;
  u64 wq;
};

// The same constant for the FP64 arithmetic path: w and w/q as doubles (exact integer w < 2^50).
struct MulOpD {
  double w;
  double wq;
};

// flags in DevMod::split_fwd_mask / split_inv_mask beside the per-pass reduce bits (context.cpp plan_f64_split)
constexpr u32 kPlanStoreReduce = 1u << 30;  // fwd: the head's outputs / inv: the middle kernel's outputs exceed the 48-bit packed range
                                            // unless reduced in front of the store (packed rows only; 8-byte rows never reduce there)
constexpr u32 kPlanScaleReduce = 1u << 29;  // inv: the tail's scaling product must be reduced before one conditional add makes it canonical

struct DevMod {
  u64 q;
  u64 q2;       // 2q
  u64 bar_lo;   // floor(2^128 / q), low word
  u64 bar_hi;   // floor(2^128 / q), high word
  MulOp ninv;   // n^{-1} mod q
  // FP64 path (primes < 2^50 whose range simulation succeeds, see context.cpp): residues are held as
  // exact integers in doubles; *_reduce_mask bit p = "reduce every value mod q at the start of pass p".
  u32 use_f64;
  u32 fwd_reduce_mask;   // pass structure with 16 elements per thread (stand-alone transforms)
  u32 inv_reduce_mask;
  // split transforms (nttshape.hpp): bit p = reduce at the start of middle pass p; bit 8 = reduce at the start of the tail;
  // kPlanStoreReduce / kPlanScaleReduce below
  u32 split_fwd_mask;
  u32 split_inv_mask;
  u32 split_ok;          // the FP64 range plan of the split structure succeeded
  // pseudo-Mersenne form q = 2^61 - pm_c with pm_c < 2^28 (every SEAL auxiliary prime): 128-bit values are
  // reduced by folding at bit 64 (2^64 = 8*pm_c) and at bit 61 instead of a two-word Barrett; 0 = not applicable
  u32 pm_c;
  u32 pad_[1];

  double qd;     // (double) q
  double qinv;   // 1.0 / q
  MulOpD ninv_d; // n^{-1} mod q
};

struct DevCtx {
  u32 n, logn;
  u32 K;    // data-level primes
  u32 KK;   // key-level primes (K + 1, or 1 when there is no special prime)
  u32 nB;   // |B|
  u32 S;    // |Bsk| = nB + 1
  u32 P;    // KK + S moduli in `mod`
  u32 pad0;
  u64 t;    // plain modulus

  // moduli: [0, KK) key-level primes, [KK, KK+S) Bsk primes (B..., m_sk)
  DevMod mod[kMaxMod];
  // twiddles: tw_fwd[m*n + k] = psi_m^{bitrev(k)}, tw_inv[m*n + k] = psi_m^{-bitrev(k)} (k >= 1);
  // for moduli with use_f64 the 16-byte entries hold MulOpD instead of MulOp
  const MulOp* tw_fwd;
  const MulOp* tw_inv;

  // ---- BEHZ multiply (SEAL RNSTool) ----
  MulOp ext_scale[kMaxKey];            // m_tilde * (q/q_i)^{-1} mod q_i
  u64 q_to_bsk[kMaxBsk][kMaxKey];      // (q/q_i) mod Bsk_j
  u32 q_to_mtilde[kMaxKey];            // (q/q_i) mod 2^32
  u32 neg_inv_q_mod_mtilde;            // -(q^{-1}) mod 2^32
  u32 pad1;
  u64 q_mod_bsk[kMaxBsk];              // q mod Bsk_j
  MulOp inv_mtilde_mod_bsk[kMaxBsk];   // m_tilde^{-1} mod Bsk_j
  MulOp intt_scale_q[kMaxKey];         // n^{-1} * t * (q/q_i)^{-1} mod q_i  (INTT epilogue before fast_floor)
  MulOpD intt_scale_q_d[kMaxKey];      // same constant for the FP64 path
  MulOp intt_scale_bsk[kMaxBsk];       // n^{-1} * t mod Bsk_j
  MulOp inv_q_mod_bsk[kMaxBsk];        // q^{-1} mod Bsk_j
  MulOp inv_punct_B[kMaxBsk];          // (B/B_j)^{-1} mod B_j
  u64 B_to_q[kMaxKey][kMaxBsk];        // (B/B_j) mod q_i
  u64 B_to_msk[kMaxBsk];               // (B/B_j) mod m_sk
  MulOp inv_B_mod_msk;                 // B^{-1} mod m_sk
  u64 B_mod_q[kMaxKey];                // B mod q_i

  // FP64 copies of the conversion constants, valid when aux_f64 != 0: every data prime and every Bsk prime
  // takes the FP64 split path, the auxiliary base is the library's own (context.cpp), and the head / tail
  // kernels run the BEHZ conversions in exact double arithmetic
  u32 aux_f64;
  // mixed base (context.cpp): integer-policy data primes beside FP64-policy auxiliary primes.  aux_f64 == 0, but the FP64
  // copies of the Bsk-side constants below are valid and the conversions whose TARGET is an auxiliary prime run in exact FP64:
  // a data residue y < 2^62 enters them as two halves, y = yh * 2^30 + yl, with the constant q_to_bsk_hi_d for the high one
  u32 aux_mixed;
  // 48-bit packed intermediates in the split pipelines (kernels_split.hip nat_load/nat_store): every modulus involved is an
  // FP64-policy prime below 2^48.  pack_ks: the KK key primes; pack_mul: aux_f64 and the K data + S auxiliary primes.
  // pack_mul == 2 (r04): PER ROW -- every auxiliary prime is below 2^48 and some data primes are (the SEAL default set of
  // N = 16384 has three 48-bit and five 49-bit data primes beside nine 48-bit auxiliary primes -- ten 45-bit ones when this was
  // measured: 13 of 18 rows travel as 6 bytes);
  // mul_row_packed[r] says which rows of ext / D (r < K: data prime r, else auxiliary prime r - K)
  unsigned char pack_ks, pack_mul, pad3[2];
  MulOpD ext_scale_d[kMaxKey];
  double q_to_bsk_d[kMaxBsk][kMaxKey];
  double q_to_bsk_hi_d[kMaxBsk][kMaxKey];  // 2^30 * (q/q_i) mod Bsk_j (mixed base)
  double q_mod_bsk_d[kMaxBsk];
  MulOpD inv_mtilde_mod_bsk_d[kMaxBsk];
  MulOpD intt_scale_bsk_d[kMaxBsk];
  MulOpD inv_q_mod_bsk_d[kMaxBsk];
  MulOpD inv_punct_B_d[kMaxBsk];
  double B_to_q_d[kMaxKey][kMaxBsk];
  double B_to_msk_d[kMaxBsk];
  MulOpD inv_B_mod_msk_d;
  double B_mod_q_d[kMaxKey];
  // exact base-conversion sums (griddot.hpp): 1.5 * 2^(52+g), valid when conv_grid != 0 (context.cpp proves the bounds
  // for this context's moduli; otherwise the head / tail kernels reduce every product of a sum on its own)
  double conv_magic;
  u32 conv_grid;
  u32 pad4;

  // split multiply: residue indices (0..K+S-1) handled by the FP64 middle kernel with 8-byte rows (mid_res_d), with 48-bit packed
  // rows (mid_res_dp) and by the integer middle kernel (mid_res_i)
  unsigned char mid_res_d[kMaxMod];
  unsigned char mid_res_dp[kMaxMod];
  unsigned char mid_res_i[kMaxMod];
  unsigned char mul_row_packed[kMaxMod];
  u32 mid_nd, mid_ndp, mid_ni;
  u32 mul_row_mask;  // bit r = mul_row_packed[r], r < 32 (the head / tail kernels test it: a scalar load, not a byte load)
  // split key switch: key-prime indices (0..KK-1) handled by the FP64 / integer middle kernel; ks_split_ok: every key prime
  // has a policy the split kernels implement (FP64 with a split range plan, or integer with Shoup twiddle tables)
  // pack_ks == 2 (r06): PER ROW, like pack_mul == 2 -- every key prime FP64-policy and some below 2^48: the rows T[.][I][.] of those
  // primes travel as 6 bytes (ks_res_dp / ks_row_mask bit I), the others (ks_res_d) and every accumulator row as doubles
  unsigned char ks_res_d[kMaxKey + 3];
  unsigned char ks_res_dp[kMaxKey + 3];
  unsigned char ks_res_i[kMaxKey + 3];
  u32 ks_nd, ks_ndp, ks_ni, ks_split_ok;
  u32 ks_row_mask, pad5;  // bit I: the rows of key prime I are 48-bit packed (pack_ks == 1: every bit set)

  // ---- key switching (special prime = mod[KK-1]) ----
  u64 qsp_half;                        // q_sp >> 1
  u64 qsp_half_mod_q[kMaxKey];         // (q_sp >> 1) mod q_i
  MulOp inv_qsp_mod_q[kMaxKey];        // q_sp^{-1} mod q_i
  MulOpD inv_qsp_mod_q_d[kMaxKey];     // the same constants for the FP64 mod-down (moddown_d.hpp): valid for FP64-policy q_i
  double qsp_half_mod_q_d[kMaxKey];

  // ---- the steps either side of the evaluator (kernels_client.hip): BatchEncoder, Decryptor, Encryptor ----
  u32 batching;                        // t is a prime == 1 (mod 2N): mod[t_mod] = t with NTT tables
  u32 t_mod;                           // = KK + S
  DevMod tm;                           // Barrett constants of t (valid for every t)
  DevMod gamma;                        // SEAL's gamma: decrypt_scale_and_round works in the base {t, gamma}
  MulOp dec_scale_q[kMaxKey];          // t * gamma * (q/q_i)^{-1} mod q_i
  u64 q_to_t[kMaxKey];                 // (q/q_i) mod t
  u64 q_to_gamma[kMaxKey];             // (q/q_i) mod gamma
  MulOp neg_inv_q_mod_t;               // -(q^{-1}) mod t
  MulOp neg_inv_q_mod_gamma;           // -(q^{-1}) mod gamma
  MulOp inv_gamma_mod_t;               // gamma^{-1} mod t

  // ---- Evaluator_ModSwitchToNext: divide-and-round by the LAST DATA prime q_{K-1} (valid when K >= 2) ----
  u64 ms_half;                         // q_{K-1} >> 1
  u64 ms_half_mod_q[kMaxKey];          // (q_{K-1} >> 1) mod q_i, i < K-1
  MulOp ms_inv_last_mod_q[kMaxKey];    // q_{K-1}^{-1} mod q_i, i < K-1

  // ---- plaintext lifting / scaling ----
  u64 q_div_t_mod_q[kMaxKey];          // floor(q/t) mod q_i
  u64 q_mod_t;                         // q mod t
  u64 t_half_up;                       // (t + 1) >> 1
  u32 fast_plain_lift;                 // t < every q_i
  u32 pad2;
};

// the residue lists are read through aligned 32-bit scalar loads (kernels_split.hip residue_of)
static_assert(offsetof(DevCtx, mid_res_d) % 4 == 0 && offsetof(DevCtx, mid_res_dp) % 4 == 0 && offsetof(DevCtx, mid_res_i) % 4 == 0 &&
                  offsetof(DevCtx, ks_res_d) % 4 == 0 && offsetof(DevCtx, ks_res_dp) % 4 == 0 && offsetof(DevCtx, ks_res_i) % 4 == 0,
              "residue lists must start at 4-byte-aligned offsets");

}  // namespace hipbfv
