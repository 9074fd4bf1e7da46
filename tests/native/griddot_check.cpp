// tests/native/griddot_check.cpp -- host check of sunscreen_amd/csrc/griddot.hpp against 128-bit integer arithmetic.
// The same GridDot code runs in the mul_head / mul_tail kernels (IEEE doubles, fma): exactness is a property of the
// arithmetic, so it is proved here on the CPU for the bounds plan_grid_dot() accepts.  Prints "ok <cases>" or a failure.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "griddot.hpp"

typedef unsigned long long u64;
typedef __int128 i128;
using hipbfv::GridDot;

static double reduce_d(double v, double p, double pinv) { return std::fma(-std::rint(v * pinv), p, v); }

static long long mod_i128(i128 v, long long p) {
  i128 r = v % p;
  if (r < 0) r += p;
  return (long long)r;
}

struct Case {
  int ybits, cbits, terms, pbits;
};

int main() {
  std::mt19937_64 rng(0x5EA1u);
  // (source modulus bits, target modulus bits, terms, bits of the modulus the sum is reduced by)
  const Case cases[] = {{44, 46, 5, 46}, {46, 44, 5, 44}, {44, 46, 4, 46}, {49, 45, 9, 45}, {45, 49, 10, 49}, {37, 40, 3, 40},
                        {50, 48, 5, 48}, {48, 50, 5, 50}, {50, 48, 9, 48}, {36, 36, 2, 36}, {20, 20, 8, 20}, {49, 45, 18, 45},
                        {49, 47, 8, 48} /* r04: 49-bit data primes, CENTRED constants of 48-bit auxiliary primes */};
  long long total = 0;
  int planned = 0;
  for (const Case& cs : cases) {
    const long double ymax = std::ldexp(1.0L, cs.ybits), cmax = std::ldexp(1.0L, cs.cbits);
    const long double pmin = std::ldexp(1.0L, cs.pbits - 1), pmax = std::ldexp(1.0L, cs.pbits);
    double magic;
    int g;
    if (!hipbfv::plan_grid_dot(ymax, cmax, (unsigned)cs.terms, pmin, pmax, &magic, &g)) {
      std::printf("# plan rejects y<2^%d c<2^%d terms=%d p~2^%d (falls back to per-term reduction)\n", cs.ybits, cs.cbits, cs.terms, cs.pbits);
      continue;
    }
    planned++;
    for (int trial = 0; trial < 60000; trial++) {
      // a modulus of pbits bits (odd), operands of every kind: random, extreme, signed
      u64 p = ((rng() >> (64 - cs.pbits)) | (1ull << (cs.pbits - 1)) | 1ull);
      if (trial % 7 == 0) p = (1ull << cs.pbits) - 1;
      if (trial % 11 == 0) p = (1ull << (cs.pbits - 1)) + 1;
      const double pd = (double)p, pinv = 1.0 / pd;
      const int T = 1 + (int)(rng() % cs.terms);
      std::vector<long long> y(T), c(T);
      for (int i = 0; i < T; i++) {
        const int mode = trial % 5;
        u64 yv = rng() >> (64 - cs.ybits), cv = rng() >> (64 - cs.cbits);
        if (mode == 1) yv = (1ull << cs.ybits) - 1, cv = (1ull << cs.cbits) - 1;   // all maximal: the bound of the plan
        if (mode == 2 && (i & 1)) yv = 0;
        if (mode == 3) yv = (1ull << cs.ybits) - 1 - (rng() & 7), cv = (1ull << cs.cbits) - 1 - (rng() & 7);
        y[i] = (long long)yv;
        c[i] = (long long)cv;
        if (mode == 4 && (rng() & 1)) y[i] = -y[i];  // signed operands (the centred r_mtilde and alpha_sk terms)
        if ((mode == 4 || mode == 3) && (rng() & 1)) c[i] = -c[i];  // centred constants (context.cpp, r04)
      }
      GridDot gd(magic, (double)y[0], (double)c[0]);
      i128 exact = (i128)y[0] * c[0];
      for (int i = 1; i < T; i++) {
        gd.add((double)y[i], (double)c[i]);
        exact += (i128)y[i] * c[i];
      }
      // (a) the two parts are the exact sum
      const double H = gd.high(magic), E = gd.low();
      if ((i128)H + (i128)E != exact) {  // H is a multiple of 2^g below 2^(51+g): the conversion to i128 is exact
        std::printf("FAIL exact-sum y<2^%d c<2^%d T=%d trial=%d\n", cs.ybits, cs.cbits, T, trial);
        return 1;
      }
      // (b) reduce(H) + err is an exact small integer congruent to the sum
      const double v = reduce_d(H, pd, pinv) + E;
      if (std::fabs(v) >= 2251799813685248.0 || v != std::rint(v) || mod_i128((i128)v, (long long)p) != mod_i128(exact, (long long)p)) {
        std::printf("FAIL reduce y<2^%d c<2^%d T=%d trial=%d v=%.1f\n", cs.ybits, cs.cbits, T, trial, v);
        return 1;
      }
      // (c) and so is the canonical value formed the way the kernels form it
      double r = reduce_d(v, pd, pinv);
      if (r < 0) r += pd;
      if ((long long)r != mod_i128(exact, (long long)p)) {
        std::printf("FAIL canonical y<2^%d c<2^%d T=%d trial=%d\n", cs.ybits, cs.cbits, T, trial);
        return 1;
      }
      // accumulator form (sums grown across a loop) gives the same two parts
      double acc = magic, err = 0.0;
      for (int i = 0; i < T; i++) hipbfv::grid_dot_add(acc, err, (double)y[i], (double)c[i]);
      if ((i128)(acc - magic) + (i128)err != exact) {
        std::printf("FAIL accumulator form\n");
        return 1;
      }
      total++;
    }
  }
  std::printf("ok %lld planned=%d\n", total, planned);
  return 0;
}
