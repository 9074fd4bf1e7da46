// sunscreen_amd/csrc/moddown_d.hpp -- the last step of the hybrid key switch (SEAL Evaluator::switch_key_inplace, "mod-down" by the
// special prime p; bound by seal_fhe/src/evaluator_base.rs:214-240 and :300-407), per output value, in exact FP64.
//
//   out = base + (a - ((tl mod q) - (floor(p/2) mod q))) * p^-1        (mod q)
//
// with a = the key-switch accumulator's residue mod the data prime q, tl = (accumulator residue mod p + floor(p/2)) mod p -- the
// canonical integer in [0, p), whose VALUE is reduced mod q -- and base = what the result is added to (c0 / c1 of the product, a
// fused addend).  The tail kernels did this in 64-bit integer arithmetic (reduce64, two sub_mod, mul_shoup, add_mod: ~55 VALU
// instructions per output value beside ~10 to make `a` canonical first) although every quantity is an exact integer below 2^52 and
// the butterflies that produce `a` run in FP64.  Here: one reduction, two additions, one constant product, one reduction, one
// conditional add (~20 instructions), on any representative of a.  Every step is exact integer arithmetic mod q (bounds below),
// so the canonical result is the same integer the 64-bit path gives: tests/native/moddown_check.cpp compares the two on random
// and extreme operands for primes of 36 ... 50 bits (CPU; the arithmetic is IEEE double fma / add / rint on both sides).
//
// Used by mulrelin_tail_kernel and the all-FP64 arm of ks_tail_kernel (kernels_split.hip) since round 5: whole GPU suite green under it,
// interleaved A/B profiles/r05_s1_ab_*.txt (headline +1.4 %, ks_tail -7.7 %, n = 16384 +0.6 %, chi_sq +1.3 %).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define HIPBFV_MD_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define HIPBFV_MD_HD inline
#endif

namespace hipbfv {

// v - rint(v / q) * q for an integer |v| < 2^52: |result| <= q/2 (+1), exact
HIPBFV_MD_HD double md_reduce(double v, double q, double qinv) { return fma(-rint(v * qinv), q, v); }

// y * w mod q for a constant w in [0, q), wq = w / q rounded; |y| < 2^52, q < 2^50: |result| <= q * (0.5 + |y| * 2^-51), exact
HIPBFV_MD_HD double md_mul_const(double y, double w, double wq, double q) {
  const double qf = rint(y * wq);
  const double xh = y * w;
  const double xl = fma(y, w, -xh);
  return fma(-qf, q, xh) + xl;
}

// q, qinv: the data prime and 1/q;  w, wq: p^-1 mod q and its quotient by q;  half: floor(p/2) mod q;  p_above_q: p > q
// s:    any representative of a with |s| <= 2q  (ArithD::mul_const(v, n^-1) leaves |s| <= q * (0.5 + |v| * 1.5 * 2^-52), |v| < 2^52)
// tl:   the canonical integer (accumulator mod p + floor(p/2)) mod p, in [0, p), p < 2^50
// base: sum of the canonical addends, 0 <= base < 2q
// returns the canonical result in [0, q)
// Bounds (q < 2^50): |tk| <= q/2 (+1) when reduced, else 0 <= tk < p < q;  |d| < 2q + q + q = 4q < 2^52;  |dd| <= 2.5q (md_mul_const
// at |y| < 2^52);  |dd + base| < 4.5q < 2^53: an exact integer, reduced exactly.
HIPBFV_MD_HD double mod_down_d(double q, double qinv, double w, double wq, double half, bool p_above_q, double s, double tl, double base) {
  const double tk = p_above_q ? md_reduce(tl, q, qinv) : tl;
  const double d = (s - tk) + half;
  const double dd = md_mul_const(d, w, wq, q);
  const double r = md_reduce(dd + base, q, qinv);
  return r < 0.0 ? r + q : r;
}

}  // namespace hipbfv
