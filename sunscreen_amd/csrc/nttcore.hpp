// sunscreen_amd/csrc/nttcore.hpp -- device-side building blocks shared by the NTT kernels
// (kernels.hip: whole-polynomial transforms; kernels_split.hip: head / middle / tail split transforms).
#pragma once
#include <hip/hip_runtime.h>

#include "devarith.hpp"
#include "nttshape.hpp"

namespace hipbfv {

// LDS placement of coefficient e: a GF(2)-linear map of the low five index bits (a bijection on each 32-word block), found by
// tools/lds_swizzle_search.py under the LDS model of MI355X_MICROARCH.md:
//   ds_read_b64                          : two groups of 32 lanes, 32 eight-byte banks -> the five lowest lane bits must reach five
//                                          independent bank vectors;
//   ds_write_b64, ds_read2 / ds_write2   : FOUR groups of 16 contiguous lanes, 16 eight-byte banks -> the four lowest lane bits must
//                                          be independent modulo bank bit 4.
// r01-r04 searched under the first rule only (bits 0..4 ^= e5*00001 ^ e6*01010 ^ e7*10100 ^ e8*11000 here, another map in the
// middle kernels): conflict free for the reads, 25-33 % extra LDS cycles on every store whose 16 lanes reach index bit 4 -- measured
// in r05 as SQ_LDS_BANK_CONFLICT = 27-29 % of SQ_LDS_IDX_ACTIVE in mul_mid<13> / ks_mid<13> (profiles/r05_s5_*_pmc_lds.txt).  No map
// that leaves bit 4 alone satisfies both rules.  This one is conflict free under BOTH for every pass window of the whole-polynomial
// transforms (N = 1024 ... 16384, 8 and 16 elements per thread, the linear read-out included) AND of the middle kernels
// (kernels_split.hip blk_pos: L = 12 ... 15, 8 and 16 elements per thread):
//   low5(e) = e[0:5] ^ e4*00101 ^ e5*01110 ^ e6*01001 ^ e7*11000 ^ e8*10000
__device__ __forceinline__ u32 lds_pos(u32 e) {
  const u32 m = (((e >> 4) & 1u) * 0x05u) ^ (((e >> 5) & 1u) * 0x0Eu) ^ (((e >> 6) & 1u) * 0x09u) ^ (((e >> 7) & 1u) * 0x18u) ^ (((e >> 8) & 1u) * 0x10u);
  return e ^ m;
}

template <int LOGN, int EPT = kElemsPerThread>
struct NttShape {
  static constexpr int N = 1 << LOGN;
  static constexpr int E = EPT;
  static constexpr int T = N / EPT;
  static constexpr int NPASS = ntt_num_passes(LOGN, EPT);
  static constexpr int LDS_WORDS = N;
  // radix (number of stages) of pass p, and the number of stages before it
  static constexpr int radix(int p) { return ntt_pass_radix(LOGN, p, EPT); }
  static constexpr int before(int p) { return ntt_stages_before(LOGN, p, EPT); }
};

// element index handled by virtual thread vt in a pass that covers bit positions [LOW, LOW+R)
template <int LOW, int R>
__device__ __forceinline__ u32 elem_index(u32 vt, u32 k) {
  const u32 lo = vt & ((1u << LOW) - 1u);
  const u32 hi = vt >> LOW;
  return (hi << (LOW + R)) | (k << LOW) | lo;
}

// LDS position of element (virtual thread tid + g*T, k) of a pass over [LOW, LOW+R), written as a per-thread part and
// compile-time parts.  With tid < T and T a power of two the element is e0 | C, e0 = elem_index(tid, 0) and
// C = elem_index(g*T, k) on disjoint bits; the swizzle is linear over GF(2), so lds_pos(e0 | C) = lds_pos(e0) ^ lds_pos(C).
// P = lds_pos(e0) is zero wherever C has bits at positions >= 5 (the swizzle only changes bits 0..4), so those bits of
// lds_pos(C) are an ADDITION -- which the DS instructions take as their immediate offset -- and only its low five bits
// need an XOR: one v_xor per DISTINCT low part per pass instead of or + xor + shift-add per access (measured in r02
// against the per-access form: NTT +1-2 %, ks_mid -4.5 %).
template <int LOW, int R, int T>
__device__ __forceinline__ u32 pass_pos(u32 P, u32 tid, int g, int k) {
  static_assert((T & (T - 1)) == 0, "tid and g*T must occupy disjoint bits");
  const u32 X = lds_pos(elem_index<LOW, R>((u32)(g * T), (u32)k));  // a constant once the g, k loops are unrolled
  return (P ^ (X & 31u)) + (X & ~31u);
}
template <int LOW, int R>
__device__ __forceinline__ u32 pass_pos_base(u32 tid) { return lds_pos(elem_index<LOW, R>(tid, 0)); }

// A pointer the compiler can prove wave-uniform (scalar loads stay possible) but cannot hoist loads through:
// used to keep twiddle loads inside the transform that consumes them.
// The opacity comes from an OFFSET (an SGPR holding zero that went through an asm), not from the pointer itself: a pointer
// rebuilt from an integer loses its address space, and every load through it becomes a flat_load -- which counts against
// lgkmcnt as well as vmcnt, so the LDS exchanges' `s_waitcnt lgkmcnt(0)` wait for the twiddle fetches they were supposed to
// overlap with (found in r03: the key-switch middle kernels fetched all their vector twiddles that way).  `p + zero` keeps the
// provenance of p (a kernel argument: global address space), so the loads stay global_load / s_load.
template <class T>
__device__ __forceinline__ const T* opaque_uniform(const T* p) {
  u32 zero = 0;
  asm volatile("" : "+s"(zero));
  return p + zero;
}

// ---- arithmetic policies -------------------------------------------------------------
// ArithI: 64-bit integers, Harvey lazy butterflies with Shoup twiddles (any prime < 2^62).
struct ArithI {
  using V = u64;
  using Tw = MulOp;  // a twiddle factor
  using Sc = MulOp;  // a per-modulus constant (n^-1, the fused BEHZ scalings)
  u64 q, q2;
  const DevMod* dm;
  __device__ __forceinline__ explicit ArithI(const DevMod& m) : q(m.q), q2(m.q << 1), dm(&m) {}
  // products of lazy operands (< 4q each): canonical results
  __device__ __forceinline__ V mul_var(V a, V b) const { return reduce128_fast((u128)a * b, *dm); }
  __device__ __forceinline__ V mul_add(V a, V b, V c) const { return reduce128_fast((u128)a * b + c, *dm); }
  static __device__ __forceinline__ V from_u64(u64 x) { return x; }
  __device__ __forceinline__ V reduce(V v) const { return v; }  // lazy invariants hold without it
  // forward: X,Y in [0,4q) -> [0,4q)
  __device__ __forceinline__ void fwd(V& X, V& Y, const Tw& w) const {
    const u64 x = X >= q2 ? X - q2 : X;
    const u64 t = mul_shoup_lazy(Y, w.w, w.wq, q);
    X = x + t;
    Y = x + q2 - t;
  }
  // inverse: X,Y in [0,2q) -> [0,2q)
  __device__ __forceinline__ void inv(V& X, V& Y, const Tw& w) const {
    const u64 u = X, y = Y, s = u + y;
    X = s >= q2 ? s - q2 : s;
    Y = mul_shoup_lazy(u + q2 - y, w.w, w.wq, q);
  }
  __device__ __forceinline__ u64 canonical(V v) const {  // v in [0,4q)
    v = v >= q2 ? v - q2 : v;
    return v >= q ? v - q : v;
  }
  __device__ __forceinline__ u64 scale_canonical(V v, const Sc& sc) const { return mul_shoup(v, sc.w, sc.wq, q); }
};

// ArithD: residues as exact integers in doubles (primes < 2^50).  T = Y*W - rint(Y*(W/q))*q is exact:
// the product is split error-free with an fma, the quotient estimate is off by at most
// 0.5 + |Y|*2^-52, and every intermediate is an integer below 2^53 (range plan: context.cpp).
// A TWIDDLE is the 8-byte W alone (r03): the quotient is estimated from the rounded product, rint(fl(Y*W) * fl(1/q)) -- three
// roundings instead of two, off by at most 0.5 + |Y|*1.5*2^-52, which the range plan prices (kTwEps) -- so a transform
// fetches half the twiddle bytes it did with (W, W/q) pairs and a staged twiddle takes two registers instead of four.  Every
// workload runs at the package power cap (profiles/r03_power_samples.txt): bytes moved are time.  Per-modulus CONSTANTS
// (Sc: n^-1, the BEHZ scalings) keep the pair: they come through the scalar cache once per kernel.
struct ArithD {
  using V = double;
  using Tw = double;
  using Sc = MulOpD;
  double q, qinv;
  __device__ __forceinline__ explicit ArithD(const DevMod& m) : q(m.qd), qinv(m.qinv) {}
  static __device__ __forceinline__ V from_u64(u64 x) {
    // exact for x < 2^52: plant the integer in the mantissa of 2^52 and subtract 2^52
    return __longlong_as_double((long long)(x | 0x4330000000000000ull)) - 4503599627370496.0;
  }
  __device__ __forceinline__ V mul_const(V y, const Sc& w) const {
    const double qf = rint(y * w.wq);
    const double xh = y * w.w;
    const double xl = fma(y, w.w, -xh);
    return fma(-qf, q, xh) + xl;
  }
  // y * W mod q for a twiddle W in [0, q): |result| <= q*(0.5 + |y|*1.5*2^-52)
  __device__ __forceinline__ V mul_tw(V y, Tw w) const {
    const double xh = y * w;
    const double xl = fma(y, w, -xh);
    const double qf = rint(xh * qinv);
    return fma(-qf, q, xh) + xl;
  }
  __device__ __forceinline__ V reduce(V v) const { return fma(-rint(v * qinv), q, v); }
  // a*b mod q for two variable operands (|a*b| < 2^105): |result| <= q*(0.5 + |a*b/q|*2^-52)
  __device__ __forceinline__ V mul_var(V a, V b) const {
    const double xh = a * b;
    const double xl = fma(a, b, -xh);
    const double qf = rint(xh * qinv);
    return fma(-qf, q, xh) + xl;
  }
  __device__ __forceinline__ V mul_add(V a, V b, V c) const { return reduce(mul_var(a, b) + c); }
  __device__ __forceinline__ void fwd(V& X, V& Y, const Tw& w) const {
    const double t = mul_tw(Y, w), x = X;
    X = x + t;
    Y = x - t;
  }
  __device__ __forceinline__ void inv(V& X, V& Y, const Tw& w) const {
    const double u = X, y = Y;
    X = u + y;
    Y = mul_tw(u - y, w);
  }
  static __device__ __forceinline__ u64 to_bits(V v) {  // v an integer in [0, 2^52)
    return (u64)__double_as_longlong(v + 4503599627370496.0) & 0x000FFFFFFFFFFFFFull;
  }
  __device__ __forceinline__ u64 to_u64(V v) const {  // v an integer in (-q, q)
    v = v < 0.0 ? v + q : v;
    return to_bits(v);
  }
  __device__ __forceinline__ u64 canonical(V v) const { return to_u64(reduce(v)); }
  __device__ __forceinline__ u64 scale_canonical(V v, const Sc& sc) const { return to_u64(reduce(mul_const(v, sc))); }
};

}  // namespace hipbfv
