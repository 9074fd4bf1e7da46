// tests/native/moddown_check.cpp -- host check of sunscreen_amd/csrc/moddown_d.hpp against 128-bit integer arithmetic: the FP64 form
// of the key switch's mod-down step must give, value for value, the canonical residue the 64-bit integer path of the tail kernels
// gives (reduce64 / sub_mod / mul_shoup / add_mod: plain modular arithmetic, restated here with __int128).  IEEE doubles, fma, rint
// on both the host and the device: exactness is a property of the arithmetic.  Prints "ok <cases>" or a failure.
#include <cstdint>
#include <cstdio>
#include <random>

#include "moddown_d.hpp"

typedef unsigned long long u64;
typedef __int128 i128;

static u64 mulmod(u64 a, u64 b, u64 m) { return (u64)((unsigned __int128)a * b % m); }
static u64 powmod(u64 a, u64 e, u64 m) {
  u64 r = 1;
  for (a %= m; e; e >>= 1, a = mulmod(a, a, m))
    if (e & 1) r = mulmod(r, a, m);
  return r;
}
static bool is_prime(u64 n) {
  if (n < 4) return n > 1;
  if (!(n & 1)) return false;
  u64 d = n - 1;
  int s = 0;
  while (!(d & 1)) d >>= 1, s++;
  for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    if (a % n == 0) continue;
    u64 x = powmod(a, d, n);
    if (x == 1 || x == n - 1) continue;
    bool comp = true;
    for (int i = 1; i < s && comp; i++) {
      x = mulmod(x, x, n);
      if (x == n - 1) comp = false;
    }
    if (comp) return false;
  }
  return true;
}
static u64 prime_below(u64 v) {
  v |= 1;
  while (!is_prime(v)) v -= 2;
  return v;
}

int main() {
  std::mt19937_64 rng(0xD0D0u);
  long long total = 0;
  for (int qbits = 36; qbits <= 50; qbits += 2) {
    for (int pbits = 36; pbits <= 50; pbits += 7) {
      for (int which = 0; which < 3; which++) {
        // primes at the top, the bottom and somewhere inside their size class
        const u64 qtop = (1ull << qbits) - 1, qbot = (1ull << (qbits - 1)) + 1;
        const u64 q = prime_below(which == 0 ? qtop : which == 1 ? qbot + 2000 : qbot + (rng() % (qtop - qbot)));
        u64 p = prime_below(which == 0 ? (1ull << pbits) - 1 : (1ull << (pbits - 1)) + 1 + (rng() % ((1ull << (pbits - 1)) - 2)));
        if (p == q) p = prime_below(p - 2);
        const u64 w = powmod(p % q, q - 2, q);  // p^-1 mod q
        const u64 half = (p >> 1) % q;
        const double qd = (double)q, qinv = 1.0 / qd, wd = (double)w, wq = (double)((long double)w / (long double)q), hd = (double)half;
        for (int trial = 0; trial < 40000; trial++) {
          // s: a representative of a, |s| <= 2q; tl in [0, p); base in [0, 2q)
          const int mode = trial % 8;
          long long s = (long long)(rng() % (4 * q + 1)) - (long long)(2 * q);
          u64 tl = rng() % p, base = rng() % (2 * q - 1);
          if (mode == 1) s = (long long)(2 * q), tl = p - 1, base = 2 * q - 2;
          if (mode == 2) s = -(long long)(2 * q), tl = 0, base = 0;
          if (mode == 3) s = (long long)(q / 2) + 1, tl = p - 1 - (rng() & 3), base = q - 1;
          if (mode == 4) s = -(long long)(q / 2) - 1, tl = rng() & 3, base = 2 * q - 2 - (rng() & 3);
          if (mode == 5) s = 0, tl = p / 2, base = q;
          if (mode == 6) s = (long long)(2 * q) - (long long)(rng() & 7), tl = 0, base = 2 * q - 2;  // the largest dd + base
          const double got = hipbfv::mod_down_d(qd, qinv, wd, wq, hd, p > q, (double)s, (double)tl, (double)base);
          // the integer path: a canonical, tk = tl mod q, tk -= half, d = (a - tk) * p^-1, out = base + d
          const u64 a = (u64)(((i128)s % (i128)q + (i128)q) % (i128)q);
          u64 tk = tl % q;
          tk = (tk + q - half) % q;
          u64 d = (a + q - tk) % q;
          d = mulmod(d, w, q);
          const u64 want = (u64)(((unsigned __int128)base + d) % q);
          if (got < 0.0 || got >= qd || got != (double)(long long)got || (u64)got != want) {
            std::printf("FAIL q=%llu p=%llu s=%lld tl=%llu base=%llu got=%.1f want=%llu\n", q, p, s, tl, base, got, want);
            return 1;
          }
          total++;
        }
        // md_mul_const on its own, over the range the inverse transforms hand it (|y| < 2^52): exact and within its bound
        for (int trial = 0; trial < 20000; trial++) {
          long long y = (long long)(rng() >> 12) - (1ll << 51);
          if (trial % 5 == 0) y = (trial & 1) ? (1ll << 52) - 1 : -((1ll << 52) - 1);
          const double r = hipbfv::md_mul_const((double)y, wd, wq, qd);
          const i128 exact = (i128)y * (i128)w;
          if (r != (double)(long long)r || (i128)((exact - (i128)(long long)r) % (i128)q) != 0 || std::fabs(r) > qd * 2.5) {
            std::printf("FAIL mul_const q=%llu y=%lld r=%.1f\n", q, y, r);
            return 1;
          }
          total++;
        }
      }
    }
  }
  std::printf("ok %lld\n", total);
  return 0;
}
