// sunscreen_amd/csrc/devarith.hpp -- 64-bit modular arithmetic for gfx950 device code.
//
// gfx950 has no 64x64->128 multiply: a 64-bit product is 4 v_mad_u64_u32 (half rate, measured
// 50 lane-ops/clk/CU; tools/microbench.hip), so every routine here is written to minimise the
// number of 32x32 multiplies:
//   * constant multiplicands carry a Shoup quotient (MulOp): 10 multiplies, no division;
//   * variable x variable products use a one-word Barrett with a per-modulus shift: 11 multiplies;
//   * sums of products are accumulated in 128 bits and reduced once.
// All results are canonical residues unless the name says "lazy".
#pragma once
#include <hip/hip_runtime.h>

#include "devctx.hpp"

namespace hipbfv {

typedef unsigned __int128 u128;

// A pointer that was LOADED from memory (the descriptor / pointer tables of the combined handle-level calls and of the graph
// executor) has no provenance the compiler could infer an address space from: accesses through it would be flat_load /
// flat_store, which also count against lgkmcnt and pay the aperture check.  Everything those tables point to is device or
// pinned host memory, i.e. the global address space: say so.
template <class T>
using global_ptr = __attribute__((address_space(1))) T*;
template <class T>
__device__ __forceinline__ global_ptr<T> as_global(T* p) {
  return (global_ptr<T>)p;
}

__device__ __forceinline__ u64 mulhi64(u64 a, u64 b) { return __umul64hi(a, b); }

// x * w mod q in [0, 2q) for any 64-bit x (w < q, wq = floor(w*2^64/q))
__device__ __forceinline__ u64 mul_shoup_lazy(u64 x, u64 w, u64 wq, u64 q) {
  return x * w - mulhi64(x, wq) * q;
}
__device__ __forceinline__ u64 mul_shoup(u64 x, u64 w, u64 wq, u64 q) {
  u64 r = mul_shoup_lazy(x, w, wq, q);
  return r >= q ? r - q : r;
}
__device__ __forceinline__ u64 mul_shoup(u64 x, const MulOp& m, u64 q) { return mul_shoup(x, m.w, m.wq, q); }

__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q) {
  u64 s = a + b;
  return s >= q ? s - q : s;
}
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { return a >= b ? a - b : a + q - b; }
__device__ __forceinline__ u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// x mod q for any 64-bit x, using the high word of floor(2^128/q)
__device__ __forceinline__ u64 reduce64(u64 x, const DevMod& m) {
  u64 r = x - mulhi64(x, m.bar_hi) * m.q;
  return r >= m.q ? r - m.q : r;
}

// x mod q for any 128-bit x (two-word Barrett)
__device__ __forceinline__ u64 reduce128(u128 x, const DevMod& m) {
  const u64 x0 = (u64)x, x1 = (u64)(x >> 64);
  u64 carry = mulhi64(x0, m.bar_lo);
  u128 t2 = (u128)x0 * m.bar_hi;
  u64 t1 = (u64)t2 + carry;
  u64 t3 = (u64)(t2 >> 64) + (t1 < (u64)t2);
  u128 t4 = (u128)x1 * m.bar_lo;
  u64 t5 = t1 + (u64)t4;
  carry = (u64)(t4 >> 64) + (t5 < t1);
  u64 qhat = x1 * m.bar_hi + t3 + carry;
  u64 r = x0 - qhat * m.q;
  return r >= m.q ? r - m.q : r;
}

// x mod (2^61 - c) for x < 2^127 and c < 2^28: 2^64 = 8c and 2^61 = c (mod q), so fold twice
__device__ __forceinline__ u64 reduce128_pm61(u128 x, u64 q, u32 c) {
  const u64 h = (u64)(x >> 64), l = (u64)x;
  const u128 y = (u128)h * (u64)(c << 3) + l;                 // < 2^94 + 2^64
  const u64 yh = (u64)(y >> 61), yl = (u64)y & ((1ull << 61) - 1);  // yh < 2^34
  u64 z = yl + yh * (u64)c;                                   // < 2^61 + 2^62
  z = z >= 2 * q ? z - 2 * q : z;
  return z >= q ? z - q : z;
}

// 128-bit reduction that takes the pseudo-Mersenne shortcut when the modulus allows it (wave-uniform branch)
__device__ __forceinline__ u64 reduce128_fast(u128 x, const DevMod& m) {
  if (m.pm_c) return reduce128_pm61(x, m.q, m.pm_c);
  return reduce128(x, m);
}

__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, const DevMod& m) { return reduce128((u128)a * b, m); }

}  // namespace hipbfv
