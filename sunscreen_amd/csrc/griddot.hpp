// sunscreen_amd/csrc/griddot.hpp -- exact sums of products in FP64 with ONE modular reduction per sum.
//
// BEHZ's base conversions (SEAL RNSTool fastbconv / fastbconv_sk, behind Evaluator_Multiply,
// seal_fhe/src/evaluator_base.rs:198-212) are short dot products  sum_i y_i * c_i  mod p  with y_i, c_i < 2^50:
// each product has up to 100 bits.  Reducing every product on its own (ArithD::mul_var) costs 7 DP instructions per
// term.  GridDot keeps the exact sum instead:
//
//   acc  starts at the magic M = 1.5 * 2^(52+g).  While |partial sum| < 2^(51+g), M + partial sum stays inside the
//        binade [2^(52+g), 2^(53+g)), whose doubles are exactly the multiples of 2^g: a1 = fma(y, c, acc) is the exact
//        value y*c + acc rounded to that grid, acc - a1 is exact (two grid points of one binade), and
//   err  += fma(y, c, acc - a1) is the (exactly representable) rounding error of that step, |.| <= 2^(g-1).
//
// So  sum = (acc - M) + err  exactly, with H = acc - M a multiple of 2^g: one quotient estimate reduces H
// (H - rint(H/p)*p is a small integer, exact in one fma), and err is added on top.  4 DP instructions per term + 5 per
// sum instead of 7 per term: it pays from about 8 terms (the floor of the 8-prime multiply, behzcore.hpp; measured there).
// plan_grid_dot() (host) picks g for a context and proves the bounds; contexts that fail keep the per-term reduction.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define HIPBFV_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define HIPBFV_HD inline
#endif

namespace hipbfv {

struct GridDot {
  double acc, err;
  // the sum starts with its first term (saves the 0 + e of a zero-initialised err)
  HIPBFV_HD GridDot(double magic, double y, double c) {
    acc = fma(y, c, magic);
    err = fma(y, c, magic - acc);
  }
  HIPBFV_HD void add(double y, double c) {
    const double a1 = fma(y, c, acc);
    err += fma(y, c, acc - a1);
    acc = a1;
  }
  // the grid part H (a multiple of 2^g, |H| < 2^(51+g)) and the exact remainder
  HIPBFV_HD double high(double magic) const { return acc - magic; }
  HIPBFV_HD double low() const { return err; }
};

// The accumulator form for sums that grow across a rolled loop: (acc, err) live in the caller's arrays.
HIPBFV_HD void grid_dot_add(double& acc, double& err, double y, double c) {
  const double a1 = fma(y, c, acc);
  err += fma(y, c, acc - a1);
  acc = a1;
}

// Host-side plan.  ymax / cmax: exclusive bounds on |y_i|, |c_i| (the largest modulus involved); terms: the longest sum;
// pmin / pmax: smallest / largest modulus a sum is reduced by.  On success *magic = 1.5 * 2^(52+g) and every value
// `reduce(H) + err` is an exact integer below 2^51 in magnitude (callers reduce it once more before using it as a
// multiplicand).  Conditions:
//   (1) terms * ymax * cmax < 2^(51+g)                  -- partial sums stay in the binade of the magic
//   (2) terms * 2^(g-1) < 2^51                          -- err is an exact integer sum
//   (3) |H - rint(H * (1/p)) * p| + |err| < 2^51        -- the reduced value is exact (quotient estimate off by at most
//                                                          0.5 + |H/p| * 2^-51: two roundings of relative size 2^-53
//                                                          each, generously doubled)
inline bool plan_grid_dot(long double ymax, long double cmax, unsigned terms, long double pmin, long double pmax, double* magic, int* g_out = nullptr) {
  if (terms == 0 || ymax <= 0 || cmax <= 0 || pmin < 2) return false;
  const long double bound = (long double)terms * ymax * cmax;
  int g = 0;
  while (std::ldexp(1.0L, 51 + g) <= bound) g++;
  if (52 + g > 1000) return false;
  const long double errmax = g == 0 ? 0.0L : (long double)terms * std::ldexp(1.0L, g - 1);
  const long double two51 = std::ldexp(1.0L, 51);
  if (errmax >= two51) return false;
  const long double hq = std::ldexp(1.0L, 51 + g) / pmin;           // |H / p|
  const long double qerr = 0.5L + hq * std::ldexp(1.0L, -51) + 1.0L;  // + 1: margin
  const long double rmax = qerr * pmax;
  if (rmax + errmax >= two51) return false;
  *magic = (double)std::ldexp(1.5L, 52 + g);
  if (g_out) *g_out = g;
  return true;
}

}  // namespace hipbfv
