// sunscreen_amd/csrc/context.cpp -- see context.hpp.
#include "context.hpp"
#include "griddot.hpp"
#include "nttshape.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace hipbfv {

namespace {

typedef unsigned __int128 u128;

inline u64 mulm(u64 a, u64 b, u64 q) { return (u64)((u128)a * b % q); }

u64 powm(u64 b, u64 e, u64 q) {
  u64 r = 1 % q;
  b %= q;
  for (; e; e >>= 1) {
    if (e & 1) r = mulm(r, b, q);
    b = mulm(b, b, q);
  }
  return r;
}

// modular inverse for any modulus coprime to a (q may be composite, e.g. 2^32 or a non-prime t)
bool invm(u64 a, u64 q, u64* out) {
  __int128 r0 = q, r1 = a % q, s0 = 0, s1 = 1;
  while (r1 != 0) {
    __int128 k = r0 / r1, tmp = r0 - k * r1;
    r0 = r1;
    r1 = tmp;
    tmp = s0 - k * s1;
    s0 = s1;
    s1 = tmp;
  }
  if (r0 != 1) return false;
  if (s0 < 0) s0 += q;
  *out = (u64)s0;
  return true;
}

inline MulOp make_mulop(u64 w, u64 q) {
  MulOp m;
  m.w = w;
  m.wq = (u64)(((u128)w << 64) / q);
  return m;
}

inline u32 bit_reverse(u32 v, int bits) {
  u32 r = 0;
  for (int i = 0; i < bits; i++) r |= ((v >> i) & 1u) << (bits - 1 - i);
  return r;
}

// product of a list of moduli reduced mod m, optionally skipping one index
u64 prod_mod(const std::vector<u64>& base, u64 m, int skip = -1) {
  u64 p = 1 % m;
  for (size_t i = 0; i < base.size(); i++)
    if ((int)i != skip) p = mulm(p, base[i] % m, m);
  return p;
}

// little-endian multi-word integer, just enough for q = prod(q_i), floor(q/t), q mod m
struct BigUint {
  std::vector<u64> w{0};
  void mul(u64 v) {
    u64 carry = 0;
    for (auto& x : w) {
      u128 p = (u128)x * v + carry;
      x = (u64)p;
      carry = (u64)(p >> 64);
    }
    if (carry) w.push_back(carry);
  }
  u64 divmod(u64 d) {  // *this /= d, returns remainder
    u64 rem = 0;
    for (size_t i = w.size(); i-- > 0;) {
      u128 cur = ((u128)rem << 64) | w[i];
      w[i] = (u64)(cur / d);
      rem = (u64)(cur % d);
    }
    while (w.size() > 1 && w.back() == 0) w.pop_back();
    return rem;
  }
  u64 mod(u64 d) const {
    u64 rem = 0;
    for (size_t i = w.size(); i-- > 0;) rem = (u64)((((u128)rem << 64) | w[i]) % d);
    return rem;
  }
  int bits() const { return (int)(64 * (w.size() - 1)) + (w.back() ? 64 - __builtin_clzll(w.back()) : 0); }
};

inline MulOpD make_mulop_d(u64 w, u64 q) {
  MulOpD m;
  m.w = (double)w;
  m.wq = (double)w / (double)q;
  return m;
}

// FP64 arithmetic path: residues are exact integers held in doubles.  A twiddle product
// T = Y*W - rint(Y*(W/q))*q is computed exactly (error-free product via fma) and satisfies
// |T| <= q*(0.5 + |Y|*2^-52); butterflies only add and subtract, so magnitudes grow per stage and
// every value must stay below 2^53.  This simulates the worst-case growth (in units of q) through the
// kernel's pass structure and decides at which pass starts all values must be reduced mod q.
// A twiddle product estimates its quotient from the ROUNDED product and the rounded 1/q (ArithD::mul_tw): relative error
// 3 * 2^-53 instead of the 2 * 2^-53 of a precomputed W/q, i.e. |T| <= q*(0.5 + |Y| * kTwEps * 2^-52).
constexpr double kTwEps = 1.5;
// ArithD::reduce(v) = v - rint(fl(v * fl(1/q))) * q: the quotient v/q = M is estimated with two roundings, off by M * 2^-52
// (no factor q: v is exact), so what is left is within q * (0.5 + M * 2^-52).  (Rounds 1-2 priced a reduction like a twiddle
// product, 0.5 + M * q * 2^-52: safe, but with 1.5 x that the 48- and 49-bit primes of N = 16384 ran out of range.)
inline double reduced(double M) { return 0.5 + M * 1.01 / 4503599627370496.0; }
bool plan_f64_path(u64 q, int logn, int ept, u32* fwd_mask, u32* inv_mask) {
  if (q >= (1ull << 50)) return false;
  const double limit = 0.98 * 9007199254740992.0 / (double)q;  // 2^53 / q with a margin
  const double eps = kTwEps * (double)q / 4503599627370496.0;  // q * 1.5 * 2^-52
  const int np = ntt_num_passes(logn, ept);
  // forward (Cooley-Tukey): X' = X + T, Y' = X - T
  {
    double M = 1.0;  // canonical input in [0,q)
    u32 mask = 0;
    for (int p = 0; p < np; p++) {
      const int r = ntt_pass_radix(logn, p, ept);
      auto run = [&](double m, bool* ok) {
        *ok = true;
        for (int s = 0; s < r; s++) {
          m = m + 0.5 + m * eps;
          if (m > limit) *ok = false;
        }
        return m;
      };
      bool ok;
      double m = run(M, &ok);
      if (!ok) {
        mask |= 1u << p;
        m = run(reduced(M), &ok);
        if (!ok) {  // bit p+16: reduce twice at the start of pass p
          mask |= 1u << (p + 16);
          m = run(reduced(reduced(M)), &ok);
          if (!ok) return false;
        }
      }
      M = m;
    }
    *fwd_mask = mask;
  }
  // inverse (Gentleman-Sande): X' = X + Y, Y' = (X - Y) * W ; inverse pass p uses forward pass np-1-p's window
  {
    double M = 1.0;
    u32 mask = 0;
    for (int p = 0; p < np; p++) {
      const int r = ntt_pass_radix(logn, np - 1 - p, ept);
      auto run = [&](double m, bool* ok) {
        *ok = true;
        for (int s = 0; s < r; s++) {
          const double d = 2.0 * m;  // |X - Y| and |X + Y|
          if (d > limit) *ok = false;
          const double t = 0.5 + d * eps;
          m = d > t ? d : t;
        }
        return m;
      };
      bool ok;
      double m = run(M, &ok);
      if (!ok) {
        mask |= 1u << p;
        m = run(reduced(M), &ok);
        if (!ok) {  // bit p+16: reduce twice at the start of pass p
          mask |= 1u << (p + 16);
          m = run(reduced(reduced(M)), &ok);
          if (!ok) return false;
        }
      }
      M = m;
    }
    *inv_mask = mask;
  }
  return true;
}

// Same growth simulation for the split (head / middle / tail) structure of nttshape.hpp.
//
// [r05] Besides the per-pass reduce bits the plan says where the kernels may SKIP a reduction they used to do unconditionally
// (devctx.hpp kPlan*): the one in front of every 48-bit packed store (the packed form holds |v| < 2^47: a few q for the 42 ... 44-bit
// primes of the default sets, so the head's three stages and a middle kernel whose last pass starts reduced fit as they are)
// and the one between the tail's scaling product and its canonicalisation.  Two changes make that pay:
//   * the inverse starts from the bound its inputs actually have -- a tensor term a * b of two forward outputs (ArithD::mul_var:
//     q * (0.5 + |a * b| / q * 1.5 * 2^-52)) or a reduced accumulator -- instead of a flat 2.5 q, which moves the one reduction the
//     middle passes need to their LAST pass for the default primes;
//   * when the middle passes would end above the packed range, the reduction goes to the start of the last pass if it is not there
//     yet (same count for unpacked rows: it replaces the tail's), and only a prime too wide even for that keeps the store-side one.
// The masks are simulated for rows that travel as 8-byte doubles (no reduction at a store); a packed row that does reduce at a
// store only starts the next kernel smaller, so one plan serves both.  tests/test_fp64_range_plan_cpu.py replays the plan, the
// flags and the entry bound against an independent model.
bool plan_f64_split(u64 q, int logn, u32* fwd_mask, u32* inv_mask) {
  if (q >= (1ull << 50) || logn < 12 || logn > 15) return false;
  const double limit = 0.98 * 9007199254740992.0 / (double)q;
  const double pack_room = 0.98 * 140737488355328.0 / (double)q;  // 2^47 / q with the same margin
  const bool packable = q < (1ull << 48);
  static_assert(split_inv_passes(12) >= 1 && split_inv_passes(13) >= 1 && split_inv_passes(14) >= 1 && split_inv_passes(15) >= 1,
                "the forced reduction below addresses the last inverse middle pass");
  const double eps = kTwEps * (double)q / 4503599627370496.0;
  double Mf = 1.0;
  {  // forward: head does head_log(logn) stages from canonical input, then the middle passes
    double M = 1.0;
    for (int s = 0; s < head_log(logn); s++) {
      M = M + 0.5 + M * eps;
      if (M > limit) return false;
    }
    u32 mask = 0;
    if (packable && M > pack_room) mask |= kPlanStoreReduce;
    for (int p = 0; p < split_fwd_passes(logn); p++) {
      const int r = split_fwd_radix(logn, p);
      auto run = [&](double m, bool* ok) {
        *ok = true;
        for (int s = 0; s < r; s++) {
          m = m + 0.5 + m * eps;
          if (m > limit) *ok = false;
        }
        return m;
      };
      bool ok;
      double m = run(M, &ok);
      if (!ok) {
        mask |= 1u << p;
        m = run(reduced(M), &ok);
        if (!ok) {  // reduce twice: the second reduction starts from a value already close to q/2
          mask |= 1u << (p + 16);
          m = run(reduced(reduced(M)), &ok);
          if (!ok) return false;
        }
      }
      M = m;
    }
    // the middle kernels multiply these values by key / operand residues: |a*b| must stay below 2^105
    if (M * (double)q * (double)q * M > 4.0e31) return false;
    *fwd_mask = mask;
    Mf = M;
  }
  {  // inverse: middle passes from products / accumulators, then tail_log(logn) stages in the tail
    // entry bound: mul_var(a, b) of two forward outputs, |a|, |b| <= Mf q (the MAC accumulators and the two-term tensor sum are
    // reduced on the way in: 0.5 + tiny); N = 32768 has only the stand-alone two-kernel inverse, which enters with canonical residues
    double M = 0.5 + 1.02 * kTwEps * Mf * Mf * (double)q / 4503599627370496.0;
    if (logn == 15) M = 1.0;  // (no tensor product or MAC in split form at this degree: only ntt_midinv_kernel, from canonical words)
    if (M > limit) return false;
    u32 mask = 0;
    auto run = [&](double m, int r, bool* ok) {
      *ok = true;
      for (int s = 0; s < r; s++) {
        const double d = 2.0 * m;
        if (d > limit) *ok = false;
        const double t = 0.5 + d * eps;
        m = d > t ? d : t;
      }
      return m;
    };
    const int np = split_inv_passes(logn);
    double M_last_in = M;  // what the last middle pass starts from, before any reduction there
    for (int p = 0; p <= np; p++) {
      const bool tail = p == np;
      const int r = tail ? tail_log(logn) : split_inv_radix(logn, p);
      // (only rows that CAN travel packed -- primes below 2^48 -- need their middle-kernel outputs inside the packed range; a wider
      // prime's rows are 8-byte doubles in every configuration and keep whatever reductions the growth itself asks for: ADVICE r05)
      if (tail && packable && M > pack_room) {
        // the middle kernel's outputs do not fit a packed row: reduce at the start of its last pass if that is not planned yet ...
        const int pl = np - 1;
        bool ok = true;
        if (!((mask >> pl) & 1u)) {
          mask |= 1u << pl;
          M = run(reduced(M_last_in), split_inv_radix(logn, pl), &ok);
          if (!ok) return false;  // (cannot happen: the unreduced run was in range)
        }
        // ... and a prime too wide even for that keeps the reduction in front of the store
        if (M > pack_room) mask |= kPlanStoreReduce;
      }
      if (p == np - 1) M_last_in = M;
      bool ok;
      double m = run(M, r, &ok);
      if (!ok) {
        mask |= tail ? (1u << 8) : (1u << p);
        m = run(reduced(M), r, &ok);
        if (!ok) {
          mask |= tail ? (1u << 24) : (1u << (p + 16));
          m = run(reduced(reduced(M)), r, &ok);
          if (!ok) return false;
        }
      }
      M = m;
    }
    // the tail multiplies by n^-1 (or a fused BEHZ scaling): ArithD::mul_const leaves q * (0.5 + |v| * 2^-52); canonical after ONE
    // conditional add while that stays below q -- otherwise the product is reduced first
    if (M * (double)q * 1.02 / 4503599627370496.0 > 0.45) mask |= kPlanScaleReduce;
    *inv_mask = mask;
  }
  return true;
}

}  // namespace

void debug_f64_plan(u64 q, int logn, u32 out[6]) {
  for (int i = 0; i < 6; i++) out[i] = 0;
  out[0] = plan_f64_path(q, logn, 16, &out[1], &out[2]) ? 1u : 0u;
  out[3] = out[0] && plan_f64_split(q, logn, &out[4], &out[5]) ? 1u : 0u;
}

bool is_prime_u64(u64 v) {
  if (v < 2) return false;
  for (u64 p : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    if (v == p) return true;
    if (v % p == 0) return false;
  }
  u64 d = v - 1;
  int s = 0;
  while ((d & 1) == 0) d >>= 1, s++;
  for (u64 a : {2ull, 3ull, 5ull, 7ull, 11ull, 13ull, 17ull, 19ull, 23ull, 29ull, 31ull, 37ull}) {
    u64 x = powm(a, d, v);
    if (x == 1 || x == v - 1) continue;
    bool witness = true;
    for (int r = 1; r < s && witness; r++) {
      x = mulm(x, x, v);
      if (x == v - 1) witness = false;
    }
    if (witness) return false;
  }
  return true;
}

std::vector<u64> find_primes(u64 factor, int bits, size_t count) {
  std::vector<u64> out;
  if (bits < 2 || bits > 62 || factor == 0) return out;
  const u64 floor_v = 1ull << (bits - 1);
  u64 cand = ((1ull << bits) - 1) / factor * factor + 1;
  while (out.size() < count && cand > floor_v) {
    if (is_prime_u64(cand)) out.push_back(cand);
    if (cand < factor) break;
    cand -= factor;
  }
  return out;
}

u64 minimal_primitive_root(u64 two_n, u64 q) {
  if ((q - 1) % two_n) return 0;
  const u64 cof = (q - 1) / two_n;
  u64 gen = 0;
  for (u64 g = 2; g < 4096 && !gen; g++) {
    u64 c = powm(g, cof, q);
    if (powm(c, two_n >> 1, q) == q - 1) gen = c;
  }
  if (!gen) return 0;
  // every primitive 2n-th root is an odd power of gen: take the numerically smallest
  const u64 step = mulm(gen, gen, q);
  u64 best = gen, cur = gen;
  for (u64 i = 1; i < (two_n >> 1); i++) {
    cur = mulm(cur, step, q);
    best = std::min(best, cur);
  }
  return best;
}

// SEAL's default BFV coefficient moduli (util/globals.cpp); n <= 8192 @128 pinned by
// logproof/src/rings.rs:36-125, (1024,{192,256}) by seal_fhe/src/encryption_parameters.rs:340-365.
std::vector<u64> default_coeff_modulus(u64 n, int sec) {
  if (sec == 128) {
    switch (n) {
      case 1024: return {0x7e00001};
      case 2048: return {0x3fffffff000001};
      case 4096: return {0xffffee001, 0xffffc4001, 0x1ffffe0001};
      case 8192: return {0x7fffffd8001, 0x7fffffc8001, 0xfffffffc001, 0xffffff6c001, 0xfffffebc001};
      case 16384:
        return {0xfffffffd8001,  0xfffffffa0001,  0xfffffff00001,  0x1fffffff68001, 0x1fffffff50001,
                0x1ffffffee8001, 0x1ffffffea0001, 0x1ffffffe88001, 0x1ffffffe48001};
      case 32768:
        return {0x7fffffffe90001, 0x7fffffffbf0001, 0x7fffffffbd0001, 0x7fffffffba0001,
                0x7fffffffaa0001, 0x7fffffffa50001, 0x7fffffff9f0001, 0x7fffffff7e0001,
                0x7fffffff770001, 0x7fffffff380001, 0x7fffffff330001, 0x7fffffff2d0001,
                0x7fffffff170001, 0x7fffffff150001, 0x7ffffffef00001, 0xfffffffff70001};
      default: return {};
    }
  }
  if (sec == 192 && n == 1024) return {0x7f001};
  if (sec == 256 && n == 1024) return {0x3001};
  return {};
}

// SEAL CoeffModulus::MaxBitCount (HomomorphicEncryption.org security standard tables)
int max_coeff_bit_count(u64 n, int sec) {
  static const int t128[] = {27, 54, 109, 218, 438, 881};
  static const int t192[] = {19, 37, 75, 152, 305, 611};
  static const int t256[] = {14, 29, 58, 118, 237, 476};
  int idx = -1;
  for (int i = 0; i < 6; i++)
    if (n == (1024ull << i)) idx = i;
  if (idx < 0) return 0;
  if (sec == 128) return t128[idx];
  if (sec == 192) return t192[idx];
  if (sec == 256) return t256[idx];
  return 0;
}

Context::~Context() {
  if (dev_) (void)hipFree(dev_);
  if (tw_fwd_) (void)hipFree(tw_fwd_);
  if (tw_inv_) (void)hipFree(tw_inv_);
  if (batch_map_) (void)hipFree(batch_map_);
}

Context* Context::create(u32 n, const std::vector<u64>& key_primes, u64 t, int device, std::string* err) {
  auto fail = [&](const char* m) -> Context* {
    if (err) *err = m;
    return nullptr;
  };
  if (n < 1024 || n > 32768 || (n & (n - 1))) return fail("poly_modulus_degree must be a power of two in [1024, 32768]");
  if (key_primes.empty() || key_primes.size() > (size_t)kMaxKey) return fail("invalid coefficient modulus count");
  if (t < 2 || t >= (1ull << 60)) return fail("invalid plain modulus");
  const u64 two_n = 2ull * n;
  for (size_t i = 0; i < key_primes.size(); i++) {
    u64 q = key_primes[i];
    if (q < 2 || q >= (1ull << 60) || !is_prime_u64(q) || (q - 1) % two_n) return fail("coefficient modulus primes must be < 2^60 and == 1 mod 2n");
    for (size_t j = 0; j < i; j++)
      if (key_primes[j] == q) return fail("coefficient modulus primes must be distinct");
  }
  std::unique_ptr<Context> c(new Context());
  c->device_ = device;
  c->key_primes_ = key_primes;
  DevCtx& h = c->host_;
  h.n = n;
  h.logn = 31 - __builtin_clz(n);
  h.KK = (u32)key_primes.size();
  h.K = h.KK > 1 ? h.KK - 1 : 1;
  h.t = t;
  const u32 K = h.K, KK = h.KK;
  std::vector<u64> q(key_primes.begin(), key_primes.begin() + K);

  // q as a big integer
  BigUint Q;
  Q.w[0] = 1;
  for (u64 p : q) Q.mul(p);
  const int q_bits = Q.bits();
  const int t_bits = 64 - __builtin_clzll(t);
  for (u64 p : q)
    if (t % p == 0) return fail("plain modulus must be coprime to the coefficient modulus");

  // auxiliary base.  SEAL (RNSTool::initialize) takes |B| = K 61-bit primes, one more when
  // 32 + bits(t) + bits(q) >= 61*(K+1), plus m_sk.  The BEHZ result does not depend on WHICH auxiliary primes
  // are used: every value that passes through Bsk is an integer determined by the q-residues alone (the
  // q -> Bsk conversions add a multiple of q that depends only on the base q; the Bsk -> q conversion is made
  // exact by the m_sk correction), provided B*m_sk exceeds the same bound 2^(32 + bits(t) + bits(q)) SEAL sizes
  // its base for.  When every data prime takes the FP64 path the library therefore picks its own base of
  // primes below 2^48 satisfying that bound, so that the auxiliary residues run on the FP64 pipe as well
  // (bit-identical results; tests compare against the oracle, which keeps SEAL's base).
  const int need_bits = 32 + t_bits + q_bits;
  // [r04] What the base has to hold, derived instead of copied.  Every row computes its residues exactly whatever the size of
  // the integers behind them; the base size enters in ONE place, the Shenoy-Kumaresan step.  sm_mrq leaves operands
  // |x'| <= q/2 * (1 + 2K/m~); a tensor coefficient is a sum of at most m = min(size_a, size_b) <= 8 negacyclic products
  // (Evaluator::multiply accepts size_a + size_b <= 16), so |c'| <= m * N * q^2/4 * (1 + eps) <= 2 N q^2 (1 + eps).  With
  // T = t * c', fast_floor delivers F = floor(T/q) - a', a' in [0, K), so |F| <= 2 t N q (1 + eps) + K + 1, and the conversion
  // Bsk -> q recovers F exactly iff its correction alpha = e - floor(F/B), e in [0, nB), is below m_sk/2 in magnitude:
  // |F| < B * (m_sk/2 - nB - 1).  B * m_sk >= 8 * 2^bits(t) * N * 2^bits(q) = 2^(bits(t) + log2 N + bits(q) + 3) satisfies that
  // with a factor of two to spare (four for the 2 x 2 multiply the programs issue).  SEAL's 2^(32 + bits(t) + bits(q)) is the
  // same rule with 32 bits reserved for m * N (rnstool.cpp's comment); at N = 16384 the 15 bits between the two are one
  // auxiliary prime fewer (17 rows in the multiply instead of 18), at 3 x 54 bits four 49-bit primes instead of five.
  // tests/test_behz_base_bound_cpu.py replays floor + Shenoy-Kumaresan in exact integers at the edge of the bound for the
  // bases this code picks.  (HIPBFV_SEAL_AUX=1 selects SEAL's base AND SEAL's sizing: the oracle's.)
  const int own_need_bits = t_bits + (int)h.logn + q_bits + 3;
  std::vector<u64> B;
  u64 m_sk = 0;
  bool own_base = false, data_f64 = true;
  {
    bool want = true;
    if (const char* env = std::getenv("HIPBFV_SEAL_AUX")) want = env[0] != '1';
    if (const char* env = std::getenv("HIPBFV_NO_F64"))
      if (env[0] == '1') want = false;
    auto fp64_ok = [&](u64 p) {
      u32 a, b;
      if (!plan_f64_path(p, (int)h.logn, 16, &a, &b)) return false;
      if (h.logn >= 12 && h.logn <= 15 && !plan_f64_split(p, (int)h.logn, &a, &b)) return false;
      return true;
    };
    for (u64 p : q) data_f64 = data_f64 && fp64_ok(p);
    // Integer-policy data primes (wider than 50 bits: the north star's 3 x 54-bit set) still gain from FP64 AUXILIARY residues
    // where the split multiply runs them: more than half of mul_mid's rows move from the integer butterfly (28 instructions)
    // to the FP64 one (8).  The conversions stay in integers -- they are generic in the auxiliary primes -- so only the head /
    // middle / tail kernels' per-row policy changes (mixed base: own_base without aux_f64).
    if (!data_f64) want = want && K <= 4 && h.logn >= 12 && h.logn <= 14;
    // the fewest primes, and for that count the smallest size, whose ACTUAL product reaches 2^own_need_bits (find_primes
    // returns the largest primes of a size first, so `cnt` primes of `bits` bits are worth almost cnt * bits bits).  48 bits is
    // the ceiling: the packed intermediates and the FP64 rows of the mixed base are written for auxiliary primes below 2^48.
    const int max_bits = 48;
    for (u32 cnt = 2; want && !own_base && cnt <= (u32)kMaxBsk; cnt++) {
      for (int bits = std::max(36, (own_need_bits + (int)cnt - 1) / (int)cnt); bits <= max_bits && !own_base; bits++) {
        std::vector<u64> cand = find_primes(two_n, bits, cnt + key_primes.size());
        std::vector<u64> pick;
        for (u64 p : cand) {
          if (pick.size() == cnt) break;
          if (std::find(key_primes.begin(), key_primes.end(), p) != key_primes.end() || t % p == 0 || !fp64_ok(p)) continue;
          pick.push_back(p);
        }
        if (pick.size() != cnt) continue;
        BigUint prod;
        prod.w[0] = 1;
        for (u64 p : pick) prod.mul(p);
        if (prod.bits() <= own_need_bits) continue;  // product < 2^own_need_bits
        own_base = true;
        m_sk = pick[0];
        B.assign(pick.begin() + 1, pick.end());
      }
    }
  }
  // the split multiply (kernels_split.hip) is instantiated for at most 4 data and 6 auxiliary primes: a large plain
  // modulus can make the own base longer than that, and then SEAL's shorter 61-bit base keeps the faster pipeline
  if (own_base && K <= 4 && h.logn >= 12 && h.logn <= 14 && B.size() + 1 > 6) own_base = false;
  if (own_base) {
    h.nB = (u32)B.size();
  } else {
    h.nB = K;
    if (need_bits >= 61 * (int)K + 61) h.nB++;
    std::vector<u64> aux = find_primes(two_n, 61, h.nB + 2);
    if (aux.size() != h.nB + 2) return fail("cannot find auxiliary primes");
    m_sk = aux[0];
    B.assign(aux.begin() + 2, aux.end());
  }
  h.aux_f64 = own_base && data_f64 ? 1u : 0u;
  h.aux_mixed = own_base && !data_f64 ? 1u : 0u;
  h.S = h.nB + 1;
  h.P = KK + h.S;
  if (h.S > (u32)kMaxBsk) return fail("too many primes");
  std::vector<u64> Bsk = B;
  Bsk.push_back(m_sk);
  const u64 m_tilde = 1ull << 32;

  // per-modulus constants and NTT tables
  std::vector<u64> all;
  all.insert(all.end(), key_primes.begin(), key_primes.end());
  all.insert(all.end(), Bsk.begin(), Bsk.end());
  // the plain modulus joins the table when it supports batching (BatchEncoder transforms over Z_t)
  c->batching_ = is_prime_u64(t) && (t - 1) % two_n == 0;
  h.batching = c->batching_ ? 1u : 0u;
  h.t_mod = h.P;
  if (c->batching_) all.push_back(t);
  const u32 nmods = (u32)all.size();
  auto fill_basic = [&](DevMod& dm, u64 p) {
    dm.q = p;
    dm.q2 = p << 1;
    u128 ratio = (~(u128)0) / p;  // p is odd: floor((2^128-1)/p) == floor(2^128/p)
    dm.bar_lo = (u64)ratio;
    dm.bar_hi = (u64)(ratio >> 64);
    dm.pm_c = 0;
    if ((p >> 60) == 1 && ((1ull << 61) - p) < (1ull << 28)) dm.pm_c = (u32)((1ull << 61) - p);
    dm.qd = (double)p;
    dm.qinv = 1.0 / (double)p;
  };
  std::vector<MulOp> twf((size_t)nmods * n), twi((size_t)nmods * n);
  for (u32 m = 0; m < nmods; m++) {
    const u64 p = all[m];
    DevMod& dm = h.mod[m];
    fill_basic(dm, p);
    u64 ninv;
    if (!invm(n, p, &ninv)) return fail("n not invertible");
    dm.ninv = make_mulop(ninv, p);
    dm.ninv_d = make_mulop_d(ninv, p);
    dm.use_f64 = plan_f64_path(p, (int)h.logn, 16, &dm.fwd_reduce_mask, &dm.inv_reduce_mask) ? 1u : 0u;
    dm.split_ok = dm.use_f64 && plan_f64_split(p, (int)h.logn, &dm.split_fwd_mask, &dm.split_inv_mask) ? 1u : 0u;
    if (const char* env = std::getenv("HIPBFV_NO_F64"))
      if (env[0] == '1') dm.use_f64 = 0, dm.split_ok = 0;
    const u64 psi = minimal_primitive_root(two_n, p);
    if (!psi) return fail("no primitive root");
    u64 ipsi;
    invm(psi, p, &ipsi);
    u64 pw = 1, ipw = 1;
    for (u32 i = 0; i < n; i++) {
      const u32 k = bit_reverse(i, h.logn);
      if (dm.use_f64) {
        // FP64 policy: the modulus' region holds n doubles (the twiddle alone, ArithD::Tw) in its first half
        const double f = (double)pw, b = (double)ipw;
        std::memcpy(reinterpret_cast<double*>(&twf[(size_t)m * n]) + k, &f, sizeof(double));
        std::memcpy(reinterpret_cast<double*>(&twi[(size_t)m * n]) + k, &b, sizeof(double));
      } else {
        twf[(size_t)m * n + k] = make_mulop(pw, p);
        twi[(size_t)m * n + k] = make_mulop(ipw, p);
      }
      pw = mulm(pw, psi, p);
      ipw = mulm(ipw, ipsi, p);
    }
  }

  // ---- BEHZ constants ----
  for (u32 i = 0; i < K; i++) {
    u64 inv_punct;
    if (!invm(prod_mod(q, q[i], (int)i), q[i], &inv_punct)) return fail("base not coprime");
    h.ext_scale[i] = make_mulop(mulm(m_tilde % q[i], inv_punct, q[i]), q[i]);
    h.q_to_mtilde[i] = (u32)prod_mod(q, m_tilde, (int)i);
    const u64 nt = mulm(h.mod[i].ninv.w, t % q[i], q[i]);
    h.intt_scale_q[i] = make_mulop(mulm(nt, inv_punct, q[i]), q[i]);
    h.intt_scale_q_d[i] = make_mulop_d(mulm(nt, inv_punct, q[i]), q[i]);
    h.B_mod_q[i] = prod_mod(B, q[i]);
    for (u32 j = 0; j < h.nB; j++) h.B_to_q[i][j] = prod_mod(B, q[i], (int)j);
  }
  {
    u64 inv;
    if (!invm(prod_mod(q, m_tilde), m_tilde, &inv)) return fail("q not invertible mod m_tilde");
    h.neg_inv_q_mod_mtilde = (u32)((m_tilde - inv) & (m_tilde - 1));
  }
  for (u32 j = 0; j < h.S; j++) {
    const u64 p = Bsk[j];
    for (u32 i = 0; i < K; i++) h.q_to_bsk[j][i] = prod_mod(q, p, (int)i);
    h.q_mod_bsk[j] = prod_mod(q, p);
    u64 inv;
    invm(m_tilde % p, p, &inv);
    h.inv_mtilde_mod_bsk[j] = make_mulop(inv, p);
    invm(h.q_mod_bsk[j], p, &inv);
    h.inv_q_mod_bsk[j] = make_mulop(inv, p);
    h.intt_scale_bsk[j] = make_mulop(mulm(h.mod[KK + j].ninv.w, t % p, p), p);
    h.intt_scale_bsk_d[j] = make_mulop_d(h.intt_scale_bsk[j].w, p);  // meaningful for FP64-path moduli only
  }
  for (u32 j = 0; j < h.nB; j++) {
    u64 inv;
    invm(prod_mod(B, B[j], (int)j), B[j], &inv);
    h.inv_punct_B[j] = make_mulop(inv, B[j]);
    h.B_to_msk[j] = prod_mod(B, m_sk, (int)j);
  }
  {
    u64 inv;
    invm(prod_mod(B, m_sk), m_sk, &inv);
    h.inv_B_mod_msk = make_mulop(inv, m_sk);
  }

  if (h.aux_f64) {
    for (u32 i = 0; i < K; i++) {
      h.ext_scale_d[i] = make_mulop_d(h.ext_scale[i].w, q[i]);
      h.B_mod_q_d[i] = (double)h.B_mod_q[i];
      for (u32 j = 0; j < h.nB; j++) h.B_to_q_d[i][j] = (double)h.B_to_q[i][j];
    }
    for (u32 j = 0; j < h.S; j++) {
      const u64 p = Bsk[j];
      for (u32 i = 0; i < K; i++) h.q_to_bsk_d[j][i] = (double)h.q_to_bsk[j][i];
      h.q_mod_bsk_d[j] = (double)h.q_mod_bsk[j];
      h.inv_mtilde_mod_bsk_d[j] = make_mulop_d(h.inv_mtilde_mod_bsk[j].w, p);
      h.inv_q_mod_bsk_d[j] = make_mulop_d(h.inv_q_mod_bsk[j].w, p);
    }
    for (u32 j = 0; j < h.nB; j++) {
      h.inv_punct_B_d[j] = make_mulop_d(h.inv_punct_B[j].w, B[j]);
      h.B_to_msk_d[j] = (double)h.B_to_msk[j];
    }
    h.inv_B_mod_msk_d = make_mulop_d(h.inv_B_mod_msk.w, m_sk);
    // Exact sums for the q -> Bsk conversion of the multiply's floor (griddot.hpp; used by the 8-prime tail kernel, where
    // it pays): every term is a residue mod some q_i times a constant below the target Bsk_j, K terms per sum (the plan
    // allows K + 1, the length of the extension's sum with its r_mtilde term).  One grid serves every target; the sums
    // are reduced by the Bsk primes.
    long double qmax = 0, bmax = 0, bmin = 1e30L;
    for (u32 i = 0; i < K; i++) qmax = std::max(qmax, (long double)q[i]);
    for (u64 p : Bsk) bmax = std::max(bmax, (long double)p), bmin = std::min(bmin, (long double)p);
    double magic = 0;
    bool ok = plan_grid_dot(qmax, bmax, K + 1, bmin, bmax, &magic);
    // [r04] 49-bit data primes beside the 48-bit auxiliary primes of the derived base bound (n = 16384: nine of them) are one
    // bit over that plan.  The floor's sums have exactly K terms, and a constant c mod Bsk_j may as well be its CENTRED
    // representative c - Bsk_j (|c| <= Bsk_j / 2): every consumer of q_to_bsk_d forms y * c with a sign-symmetric reduction
    // (ArithD::mul_var, GridDot) and the sums are reduced before use, so the canonical results are the same; the plan then
    // closes with one bit to spare (tests/native/griddot_check.cpp: 49 x 47 bits, 8 terms, signed constants).
    bool centred = false;
    if (!ok) ok = centred = plan_grid_dot(qmax, std::floor(bmax / 2) + 1, K, bmin, bmax, &magic);
    if (const char* env = std::getenv("HIPBFV_NO_GRID"))
      if (env[0] == '1') ok = false;
    h.conv_grid = ok ? (centred ? 2u : 1u) : 0u;  // 2: the plan holds for the floor's K terms with centred constants only
    h.conv_magic = ok ? magic : 0.0;
    if (ok && centred)
      for (u32 j = 0; j < h.S; j++)
        for (u32 i = 0; i < K; i++)
          if (h.q_to_bsk[j][i] > Bsk[j] / 2) h.q_to_bsk_d[j][i] = -(double)(Bsk[j] - h.q_to_bsk[j][i]);
  }

  if (h.aux_mixed) {  // the Bsk-side constants in FP64 form (every auxiliary prime is FP64-capable by selection)
    for (u32 j = 0; j < h.S; j++) {
      const u64 p = Bsk[j];
      for (u32 i = 0; i < K; i++) {
        h.q_to_bsk_d[j][i] = (double)h.q_to_bsk[j][i];
        h.q_to_bsk_hi_d[j][i] = (double)mulm(h.q_to_bsk[j][i], (1ull << 30) % p, p);
      }
      h.q_mod_bsk_d[j] = (double)h.q_mod_bsk[j];
      h.inv_mtilde_mod_bsk_d[j] = make_mulop_d(h.inv_mtilde_mod_bsk[j].w, p);
      h.inv_q_mod_bsk_d[j] = make_mulop_d(h.inv_q_mod_bsk[j].w, p);
    }
    for (u32 j = 0; j < h.nB; j++) {
      h.inv_punct_B_d[j] = make_mulop_d(h.inv_punct_B[j].w, B[j]);
      h.B_to_msk_d[j] = (double)h.B_to_msk[j];
    }
    h.inv_B_mod_msk_d = make_mulop_d(h.inv_B_mod_msk.w, m_sk);
    // the mixed sums are formed exactly on one grid and reduced once (griddot.hpp): 2K + 1 terms, each a half of a data
    // residue (below 2^32; the r_mtilde term below 2^31) times a constant below the target auxiliary prime
    long double bmax = 0, bmin = 1e30L;
    for (u64 p : Bsk) bmax = std::max(bmax, (long double)p), bmin = std::min(bmin, (long double)p);
    double magic = 0;
    if (!plan_grid_dot(4294967296.0L, bmax, 2 * K + 1, bmin, bmax, &magic)) return fail("internal: no exact-sum grid for the mixed auxiliary base");
    h.conv_magic = magic;
  }


  // key switching: a key prime takes the FP64 policy when its split range plan exists, the integer policy when its
  // twiddle tables are Shoup pairs (use_f64 == 0); an FP64-table prime WITHOUT a split plan has neither form
  h.ks_split_ok = 1;
  for (u32 i = 0; i < KK; i++)
    if (h.mod[i].use_f64 && !h.mod[i].split_ok) h.ks_split_ok = 0;

  {
    // N = 4096: the multiply's middle kernel loses more on the doubled load count than its head gains (measured -1.4 %)
    bool ks = true, mul = h.aux_f64 != 0 && h.logn >= 13;
    const u64 lim = 1ull << 48;
    for (u32 i = 0; i < KK; i++) ks = ks && h.mod[i].use_f64 && h.mod[i].split_ok && h.mod[i].q < lim;
    for (u32 r = 0; r < K + h.S; r++) {
      const DevMod& dm = h.mod[r < K ? r : KK + (r - K)];
      mul = mul && dm.use_f64 && dm.split_ok && dm.q < lim;
    }
    if (const char* env = std::getenv("HIPBFV_NO_PACK")) {  // "1": neither pipeline, "mul" / "ks": not that one
      if (env[0] == '1') ks = mul = false;
      if (env[0] == 'm') mul = false;
      if (env[0] == 'k') ks = false;
    }
    h.pack_ks = ks ? 1 : 0;
    h.pack_mul = mul ? 1 : 0;
    // r04, per-row packing of the multiply's intermediates: every row FP64-policy and every AUXILIARY
    // prime below 2^48 (the library's own base always is); data rows pack when their prime is below 2^48.  The 8-prime head /
    // tail instantiations take the per-row flags (kneed > 4).  History: opt-in in r04 (slower then: the heads / tails paid more for
    // the packing than the rows saved); r05's plan-aware stores made it +0.6 % on mul+relin but -1.8 % on chi_sq, whose packed
    // SQUARING middle kernel spilt 94 instructions to scratch.  r06 sequenced that kernel like MODE 3 (no scratch) and it wins on
    // every n = 16384 workload (interleaved, one box, profiles/r06_s9_ab_packrows.txt): mul+relin 59.05 -> 59.77 K ops/s (+1.2 %),
    // chi_sq 10.36 -> 10.45 K programs/s (+0.8 %), dot_prod +0.2 % -- the DEFAULT since; HIPBFV_PACK_ROWS=0 restores 8-byte rows.
    const u32 kneed = std::max(K, h.S > 2 ? h.S - 2 : 0u);
    bool part = !mul && h.aux_f64 != 0 && h.logn >= 13 && kneed > 4;
    for (u32 r = 0; r < K + h.S && part; r++) {
      const DevMod& dm = h.mod[r < K ? r : KK + (r - K)];
      part = dm.use_f64 && dm.split_ok && (r < K || dm.q < lim);
    }
    if (const char* env = std::getenv("HIPBFV_NO_PACK"))
      if (env[0] == '1' || env[0] == 'm') part = false;
    const char* rows_env = std::getenv("HIPBFV_PACK_ROWS");
    if (part && !(rows_env && rows_env[0] == '0')) h.pack_mul = 2;  // (the auxiliary rows alone are S of the K + S rows)
  }
  if (h.pack_mul == 2 && K + h.S > 32) return fail("internal: per-row packing covers at most 32 rows (DevCtx::mul_row_mask)");
  // r06, the same per row for the key switch's intermediates: every key prime FP64-policy, some below 2^48 (N = 16384 default set:
  // the rows of 3 of the 9 key primes).  Instantiated beside the per-row multiply only (the 8-prime fused kernels), so it follows
  // pack_mul == 2; HIPBFV_NO_PACK=ks / HIPBFV_PACK_ROWS=0 leave the 8-byte rows.
  if (!h.pack_ks && h.pack_mul == 2) {
    bool all = true, some = false;
    for (u32 i = 0; i < KK; i++) {
      all = all && h.mod[i].use_f64 && h.mod[i].split_ok;
      some = some || h.mod[i].q < (1ull << 48);
    }
    const char* env = std::getenv("HIPBFV_NO_PACK");
    if (all && some && !(env && env[0] == 'k')) h.pack_ks = 2;
  }
  // key switching: a key prime takes the FP64 policy when its split range plan exists (8-byte rows of T: ks_res_d, packed rows: ks_res_dp),
  // the integer policy when its twiddle tables are Shoup pairs (use_f64 == 0)
  h.ks_nd = h.ks_ndp = h.ks_ni = 0;
  h.ks_row_mask = 0;
  for (u32 i = 0; i < KK; i++) {
    if (h.mod[i].use_f64 && h.mod[i].split_ok) {
      if (h.pack_ks == 1 || (h.pack_ks == 2 && h.mod[i].q < (1ull << 48))) {
        h.ks_res_dp[h.ks_ndp++] = (unsigned char)i;
        h.ks_row_mask |= 1u << i;
      } else {
        h.ks_res_d[h.ks_nd++] = (unsigned char)i;
      }
    } else if (!h.mod[i].use_f64) {
      h.ks_res_i[h.ks_ni++] = (unsigned char)i;
    }
  }
  h.mid_nd = h.mid_ndp = h.mid_ni = 0;
  h.mul_row_mask = 0;
  for (u32 r = 0; r < K + h.S; r++) {
    const u32 m = r < K ? r : KK + (r - K);
    h.mul_row_packed[r] = (h.pack_mul == 1 || (h.pack_mul == 2 && h.mod[m].q < (1ull << 48))) ? 1 : 0;
    if (h.mul_row_packed[r] && r < 32) h.mul_row_mask |= 1u << r;
    if (h.mod[m].use_f64 && h.mod[m].split_ok) {
      if (h.mul_row_packed[r])
        h.mid_res_dp[h.mid_ndp++] = (unsigned char)r;
      else
        h.mid_res_d[h.mid_nd++] = (unsigned char)r;
    } else {
      h.mid_res_i[h.mid_ni++] = (unsigned char)r;
    }
  }

  // ---- key switching ----
  if (KK > 1) {
    const u64 qsp = key_primes[KK - 1];
    h.qsp_half = qsp >> 1;
    for (u32 i = 0; i < K; i++) {
      h.qsp_half_mod_q[i] = h.qsp_half % q[i];
      u64 inv;
      invm(qsp % q[i], q[i], &inv);
      h.inv_qsp_mod_q[i] = make_mulop(inv, q[i]);
      h.inv_qsp_mod_q_d[i] = make_mulop_d(inv, q[i]);
      h.qsp_half_mod_q_d[i] = (double)h.qsp_half_mod_q[i];
    }
  }

  // ---- modulus switching to the next level: divide-and-round by the last data prime ----
  if (K >= 2) {
    const u64 ql = q[K - 1];
    h.ms_half = ql >> 1;
    for (u32 i = 0; i + 1 < K; i++) {
      h.ms_half_mod_q[i] = h.ms_half % q[i];
      u64 inv;
      if (!invm(ql % q[i], q[i], &inv)) return fail("coefficient modulus primes are not distinct");
      h.ms_inv_last_mod_q[i] = make_mulop(inv, q[i]);
    }
  }

  // ---- plaintext scaling ----
  {
    BigUint Qt = Q;
    h.q_mod_t = Qt.divmod(t);  // Qt = floor(q/t)
    h.t_half_up = (t + 1) >> 1;
    h.fast_plain_lift = 1;
    for (u32 i = 0; i < K; i++) {
      h.q_div_t_mod_q[i] = Qt.mod(q[i]);
      if (t >= q[i]) h.fast_plain_lift = 0;
    }
  }
  // ---- Decryptor: SEAL RNSTool::decrypt_scale_and_round in the base {t, gamma} ----
  {
    fill_basic(h.tm, t);
    if (((~(u128)0) % t) == t - 1) {  // t divides 2^128 (a power of two): floor(2^128/t) is one more than floor((2^128-1)/t)
      const u128 ratio = (~(u128)0) / t + 1;
      h.tm.bar_lo = (u64)ratio;
      h.tm.bar_hi = (u64)(ratio >> 64);
    }
    const std::vector<u64> g2 = find_primes(two_n, 61, 2);  // SEAL: [m_sk, gamma, B...] -> gamma is the second
    if (g2.size() != 2) return fail("cannot find gamma");
    const u64 gamma = g2[1];
    fill_basic(h.gamma, gamma);
    for (u32 i = 0; i < K; i++) {
      u64 inv_punct;
      invm(prod_mod(q, q[i], (int)i), q[i], &inv_punct);
      const u64 tg = mulm(t % q[i], gamma % q[i], q[i]);
      h.dec_scale_q[i] = make_mulop(mulm(tg, inv_punct, q[i]), q[i]);
      h.q_to_t[i] = prod_mod(q, t, (int)i);
      h.q_to_gamma[i] = prod_mod(q, gamma, (int)i);
    }
    u64 inv;
    if (!invm(prod_mod(q, t), t, &inv)) return fail("q not invertible mod t");
    h.neg_inv_q_mod_t = make_mulop((t - inv) % t, t);
    invm(prod_mod(q, gamma), gamma, &inv);
    h.neg_inv_q_mod_gamma = make_mulop(gamma - inv, gamma);
    if (!invm(gamma % t, t, &inv)) return fail("gamma not invertible mod t");
    h.inv_gamma_mod_t = make_mulop(inv, t);
  }
  // BatchEncoder matrix_reps_index_map (SEAL batchencoder.cpp; seal_fhe/src/encoder.rs:75-190)
  std::vector<u32> bmap;
  if (c->batching_) {
    bmap.resize(n);
    const u32 row = n >> 1, m2 = n << 1;
    u64 pos = 1;
    for (u32 i = 0; i < row; i++) {
      bmap[i] = bit_reverse((u32)((pos - 1) >> 1), h.logn);
      bmap[row | i] = bit_reverse((u32)((m2 - pos - 1) >> 1), h.logn);
      pos = (pos * 3) & (m2 - 1);
    }
  }

  // ---- upload ----
  if (device < 0) return c.release();  // host-only view (hipbfv_debug_aux_base): the tables above, nothing on a device
  if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
  const size_t tw_bytes = twf.size() * sizeof(MulOp);
  if (hipMalloc((void**)&c->tw_fwd_, tw_bytes) != hipSuccess || hipMalloc((void**)&c->tw_inv_, tw_bytes) != hipSuccess ||
      hipMalloc((void**)&c->dev_, sizeof(DevCtx)) != hipSuccess)
    return fail("hipMalloc failed");
  h.tw_fwd = c->tw_fwd_;
  h.tw_inv = c->tw_inv_;
  if (hipMemcpy(c->tw_fwd_, twf.data(), tw_bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->tw_inv_, twi.data(), tw_bytes, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->dev_, &h, sizeof(DevCtx), hipMemcpyHostToDevice) != hipSuccess)
    return fail("hipMemcpy failed");
  if (!bmap.empty()) {
    if (hipMalloc((void**)&c->batch_map_, bmap.size() * sizeof(u32)) != hipSuccess ||
        hipMemcpy(c->batch_map_, bmap.data(), bmap.size() * sizeof(u32), hipMemcpyHostToDevice) != hipSuccess)
      return fail("hipMalloc failed");
  }
  return c.release();
}

std::shared_ptr<Context> Context::next_level(std::string* err) {
  std::lock_guard<std::mutex> g(next_mu_);
  if (next_) return next_;
  if (host_.K < 2 || !chain_enabled_) {
    if (err) *err = "end of modulus switching chain reached";
    return nullptr;
  }
  std::vector<u64> primes(key_primes_.begin(), key_primes_.begin() + (host_.K - 1));
  if (host_.KK > host_.K) primes.push_back(key_primes_.back());  // the special prime stays
  Context* c = Context::create(host_.n, primes, host_.t, device_, err);
  if (!c) return nullptr;
  c->level_ = level_ + 1;
  c->chain_enabled_ = chain_enabled_;
  next_.reset(c);
  return next_;
}

}  // namespace hipbfv