// sunscreen_amd/csrc/kernels.hip -- hand-written gfx950 kernels for the BFV hot path.
//
// What each kernel replaces in the reference (all inside SEAL behind seal_fhe::Evaluator):
//   ntt_fwd / ntt_inv        NTTTables + ntt_negacyclic_harvey (context.rs:63-80 builds them)
//   behz_extend              RNSTool::fastbconv_m_tilde + sm_mrq        (Evaluator_Multiply, evaluator_base.rs:198-212)
//   tensor                   behz_ciphertext_product (dyadic)            (same)
//   behz_floor_sk            RNSTool::fast_floor + fastbconv_sk          (same)
//   ks_decompose/mac/moddown Evaluator::switch_key_inplace               (Evaluator_Relinearize / RotateRows / RotateColumns,
//                                                                          bfv_evaluator.rs:148-244)
//   galois                   GaloisTool::apply_galois                    (bfv_evaluator.rs:177-247)
//   eltwise / plain kernels  add/sub/negate/add_plain/sub_plain/multiply_plain (evaluator_base.rs:89-404)
//
// Data layout everywhere: u64[...][residue][N] with the coefficient index fastest, so that a
// wavefront's 64 lanes touch 512 contiguous bytes (coefficient-parallel kernels) or one
// workgroup owns one residue polynomial (NTT kernels, staged through LDS).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

#include "devarith.hpp"
#include "kernels.hpp"

#include "behzcore.hpp"
#include "nttcore.hpp"
#include "nttshape.hpp"

namespace hipbfv {

// Non-temporal hints on the polynomial data of the stand-alone transforms (bit 1 = loads, bit 2 = stores).  Measured on the
// batched forward + inverse workload (interleaved A/B on one box): forward with both -9 % (0.506 -> 0.46 ms per 12288
// polynomials), inverse with non-temporal stores -1.5 %; non-temporal LOADS make the inverse 13 % slower there and in the
// encrypt / decrypt pipeline (its input was just written by the previous kernel and is partly cache-resident), so it keeps
// temporal loads.
#define NTT_NT_FWD 3
#define NTT_NT_INV 2
template <bool NT, class T>
__device__ __forceinline__ T ntt_ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void ntt_st(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// =====================================================================================
// NTT: one workgroup per residue polynomial, N/16 threads, 16 coefficients per thread.
// The log2(N) radix-2 stages are grouped into ceil(logn/4) register passes; between passes
// the polynomial is exchanged through LDS (XOR-swizzled so that the strided accesses of every
// pass are bank-conflict free).
// =====================================================================================

// ---- which LDS exchanges need a workgroup barrier ----
// Virtual thread vt = tid + g*T handles, in the pass over the index window [LOW, LOW+R), the elements
// (vt >> LOW) << (LOW+R) | k << LOW | (vt & (2^LOW - 1)).  The lane bits of vt (0..5), k and g range over everything inside
// one wavefront; only the WAVE-ID bits of vt (6 .. log2 T - 1) tie a wavefront to a part of the polynomial: bit b lands at
// element bit b + R when b >= LOW, else at b.  When two consecutive passes send every wave-id bit to the same element
// bit, a wavefront reads back in the second pass exactly the elements it wrote in the first: the exchange between them is
// private to the wavefront, whose LDS accesses execute in order -- no s_barrier.  At N = 8192 (windows 9..12, 6..8, 3..5,
// 0..2) only the first exchange of a forward transform (the last of an inverse one) couples wavefronts: 1 workgroup
// barrier per transform instead of 4, and after it the eight wavefronts of a workgroup run their 2 x 512-coefficient
// blocks independently, so their load / compute / store phases stagger instead of meeting at every pass.
#define NTT_WAVE_PRIVATE 1
constexpr int wave_bit_target(int b, int low, int r) { return b >= low ? b + r : b; }
// pa, pb: forward pass numbers of the two passes an exchange connects
constexpr bool exchange_is_wave_private(int logn, int ept, int pa, int pb) {
  if (!NTT_WAVE_PRIVATE) return false;
  const int logt = logn - ilog2(ept);
  const int ra = ntt_pass_radix(logn, pa, ept), rb = ntt_pass_radix(logn, pb, ept);
  const int lowa = logn - ntt_stages_before(logn, pa, ept) - ra, lowb = logn - ntt_stages_before(logn, pb, ept) - rb;
  for (int b = 6; b < logt; b++)
    if (wave_bit_target(b, lowa, ra) != wave_bit_target(b, lowb, rb)) return false;
  return true;
}
template <bool PRIVATE>
__device__ __forceinline__ void exchange_sync() {
  if constexpr (PRIVATE) {
    // same wavefront, LDS operations are issued and executed in order: only the compiler must not move them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

// ---- forward (Cooley-Tukey, gap shrinking) ----
template <class A, int LOGN, int EPT, int S0, int R>
__device__ __forceinline__ void fwd_pass_compute(const A& ar, typename A::V (&v)[EPT], u32 tid,
                                                 const typename A::Tw* __restrict__ tw) {
  constexpr int T = NttShape<LOGN, EPT>::T;
  constexpr int G = EPT >> R;
  constexpr int LOW = LOGN - S0 - R;
#pragma unroll
  for (int g = 0; g < G; g++) {
    const u32 vt = tid + g * T;
    u32 hi = vt >> LOW;
    // LOW >= 6: the 64 lanes of a wavefront share `hi`, so the twiddle index is wave-uniform and the
    // loads go through the scalar cache (s_load) instead of 64 identical vector loads
    if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int half = 1 << (R - 1 - j);
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        u32 widx = (1u << (S0 + j)) + ((hi << j) | (u32)(k >> (R - j)));
        const typename A::Tw w = tw[widx];
        ar.fwd(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w);
      }
    }
  }
}

template <class A, int LOGN, int EPT, int PASS, bool KEEP_REGS = false>
struct FwdPasses {
  // run passes PASS..NPASS-1 with data resident in LDS on entry to every pass but the first.
  // KEEP_REGS: the last pass leaves its results in v (element ((tid + g*T) << R) | k) instead of LDS.
  static __device__ __forceinline__ void run(const A& ar, typename A::V (&v)[EPT], typename A::V* smem, u32 tid,
                                             const typename A::Tw* tw, u32 reduce_mask) {
    using Sh = NttShape<LOGN, EPT>;
    constexpr int R = Sh::radix(PASS);
    constexpr int S0 = Sh::before(PASS);
    constexpr int LOW = LOGN - S0 - R;
    constexpr int G = EPT >> R;
    const u32 P = pass_pos_base<LOW, R>(tid);
    if constexpr (PASS > 0) {
      exchange_sync<exchange_is_wave_private(LOGN, EPT, PASS - 1, PASS)>();
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) v[g * (1 << R) + k] = smem[pass_pos<LOW, R, Sh::T>(P, tid, g, k)];
    }
    if ((reduce_mask >> PASS) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[e] = ar.reduce(v[e]);
    }
    if ((reduce_mask >> (PASS + 16)) & 1u) {  // very wide primes: a second reduction (context.cpp range plan)
#pragma unroll
      for (int e = 0; e < EPT; e++) v[e] = ar.reduce(v[e]);
    }
    fwd_pass_compute<A, LOGN, EPT, S0, R>(ar, v, tid, tw);
    if constexpr (!(KEEP_REGS && PASS + 1 == Sh::NPASS)) {
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) smem[pass_pos<LOW, R, Sh::T>(P, tid, g, k)] = v[g * (1 << R) + k];
    }
    if constexpr (PASS + 1 < Sh::NPASS) FwdPasses<A, LOGN, EPT, PASS + 1, KEEP_REGS>::run(ar, v, smem, tid, tw, reduce_mask);
  }
};

// Forward NTT of one polynomial: src (global, canonical u64) -> LDS (policy representation, lazy).
template <class A, int LOGN>
__device__ __forceinline__ void ntt_fwd_to_lds(const A& ar, const u64* __restrict__ src, typename A::V* smem, u32 tid,
                                               const typename A::Tw* tw, u32 reduce_mask) {
  using Sh = NttShape<LOGN>;
  constexpr int EPT = kElemsPerThread;
  constexpr int R0 = Sh::radix(0);
  constexpr int LOW0 = LOGN - R0;
  constexpr int G0 = kElemsPerThread >> R0;
  typename A::V v[kElemsPerThread];
#pragma unroll
  for (int g = 0; g < G0; g++)
#pragma unroll
    for (int k = 0; k < (1 << R0); k++)
      v[g * (1 << R0) + k] = ar.from_u64(ntt_ld<(NTT_NT_FWD & 1) != 0>(src + elem_index<LOW0, R0>(tid + g * Sh::T, k)));
  FwdPasses<A, LOGN, EPT, 0>::run(ar, v, smem, tid, tw, reduce_mask);
}

// After a forward transform whose last pass left its results in LDS: element j (0 .. EPT-1) of the set THIS wavefront wrote
// there, enumerated so that the 64 lanes are 64 consecutive coefficients (512 contiguous bytes per wave instruction).  The
// last pass has LOW = 0: wave-id bit b sits at element bit b + R, i.e. the wave id occupies [6 + R, log2 T + R).
template <int LOGN, int EPT>
__device__ __forceinline__ u32 own_element_after_fwd(u32 tid, u32 j) {
  using Sh = NttShape<LOGN, EPT>;
  constexpr int R = Sh::radix(Sh::NPASS - 1);
  constexpr int LOGT = LOGN - ilog2(EPT);
  const u32 lane = tid & 63u, wave = tid >> 6;
  return lane | ((j & ((1u << R) - 1u)) << 6) | (wave << (6 + R)) | ((j >> R) << (LOGT + R));
}

// ---- inverse (Gentleman-Sande, gap growing) ----
template <class A, int LOGN, int EPT, int LOW, int R>
__device__ __forceinline__ void inv_pass_compute(const A& ar, typename A::V (&v)[EPT], u32 tid,
                                                 const typename A::Tw* __restrict__ tw) {
  constexpr int T = NttShape<LOGN, EPT>::T;
  constexpr int G = EPT >> R;
#pragma unroll
  for (int g = 0; g < G; g++) {
    const u32 vt = tid + g * T;
    u32 hi = vt >> LOW;
    if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);  // wave-uniform twiddles -> scalar loads
#pragma unroll
    for (int j = 0; j < R; j++) {
      const int half = 1 << j;
#pragma unroll
      for (int k = 0; k < (1 << R); k++) {
        if (k & half) continue;
        // global gap 2^(LOW+j): m = N >> (LOW+j+1) blocks, block index = element >> (LOW+j+1)
        u32 widx = (1u << (LOGN - 1 - LOW - j)) + ((hi << (R - 1 - j)) | (u32)(k >> (j + 1)));
        const typename A::Tw w = tw[widx];
        ar.inv(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w);
      }
    }
  }
}

template <class A, int LOGN, int EPT, int PASS, bool FROM_REGS = false, int STOP = 99>
struct InvPasses {
  // STOP: run passes [PASS, STOP) only (the exchange write that follows pass STOP - 1 included); a second call continues
  // inverse pass PASS covers the same bit window as forward pass NPASS-1-PASS.
  // FROM_REGS: pass 0 takes its input from v (the layout a KEEP_REGS forward transform leaves) instead of LDS.
  static __device__ __forceinline__ void run(const A& ar, typename A::V (&v)[EPT], typename A::V* smem, u32 tid,
                                             const typename A::Tw* tw, u32 reduce_mask) {
    using Sh = NttShape<LOGN, EPT>;
    constexpr int FP = Sh::NPASS - 1 - PASS;
    constexpr int R = Sh::radix(FP);
    constexpr int LOW = LOGN - Sh::before(FP) - R;
    constexpr int G = EPT >> R;
    const u32 P = pass_pos_base<LOW, R>(tid);
    if constexpr (PASS == 0) {
      if constexpr (!FROM_REGS) __syncthreads();  // the caller filled LDS cooperatively
    } else {
      exchange_sync<exchange_is_wave_private(LOGN, EPT, FP + 1, FP)>();
    }
    if constexpr (!(FROM_REGS && PASS == 0)) {
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) v[g * (1 << R) + k] = smem[pass_pos<LOW, R, Sh::T>(P, tid, g, k)];
    }
    if ((reduce_mask >> PASS) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[e] = ar.reduce(v[e]);
    }
    if ((reduce_mask >> (PASS + 16)) & 1u) {  // very wide primes: a second reduction (context.cpp range plan)
#pragma unroll
      for (int e = 0; e < EPT; e++) v[e] = ar.reduce(v[e]);
    }
    inv_pass_compute<A, LOGN, EPT, LOW, R>(ar, v, tid, tw);
    if constexpr (PASS + 1 < Sh::NPASS) {
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) smem[pass_pos<LOW, R, Sh::T>(P, tid, g, k)] = v[g * (1 << R) + k];
      if constexpr (PASS + 1 < STOP) InvPasses<A, LOGN, EPT, PASS + 1, FROM_REGS, STOP>::run(ar, v, smem, tid, tw, reduce_mask);
    }
  }
};

// Inverse NTT of the polynomial resident in LDS; the last pass leaves each thread holding the
// elements elem_index<LOGN-R, R>(tid + g*T, k), which the caller scales and stores (coalesced).
template <class A, int LOGN>
__device__ __forceinline__ void ntt_inv_from_lds(const A& ar, typename A::V (&v)[kElemsPerThread], typename A::V* smem, u32 tid,
                                                 const typename A::Tw* tw, u32 reduce_mask) {
  InvPasses<A, LOGN, kElemsPerThread, 0>::run(ar, v, smem, tid, tw, reduce_mask);
}

// Which modulus polynomial `poly` belongs to -- through the SCALAR unit.  `plan.mod[i]` with a run-time i is a byte load from the
// kernel-argument segment, which gfx950 can only do as a VECTOR load: every workgroup of a transform kernel began with
// global_load_ubyte + s_waitcnt vmcnt(0), a dependent memory round trip before its first polynomial load could be issued
// (r04, found in the load / wait listing of the ISA).  The aligned 32-bit word that holds the byte is a wave-uniform s_load.
__device__ __forceinline__ u32 plan_mod(const NttPlan& plan, u32 poly) {
  const u32 i = __builtin_amdgcn_readfirstlane((poly / plan.div) % plan.period);
  const u32* words = reinterpret_cast<const u32*>(plan.mod);
  return (words[i >> 2] >> ((i & 3u) * 8u)) & 0xffu;
}

// the store half of a forward transform whose last pass left its results in LDS
template <class A, int LOGN>
__device__ __forceinline__ void ntt_fwd_store(const A& ar, u64* x, typename A::V* smem, u32 tid) {
  using Sh = NttShape<LOGN>;
  if constexpr (NTT_WAVE_PRIVATE && Sh::T >= 64) {
    // every wavefront stores the coefficients its own last pass produced: no barrier before the store either
    exchange_sync<true>();
    {
      // per-thread and compile-time parts of the position, as in pass_pos
      const u32 e0 = own_element_after_fwd<LOGN, kElemsPerThread>(tid, 0), P = lds_pos(e0);
      u64* const x0 = x + e0;
#pragma unroll
      for (u32 j = 0; j < (u32)kElemsPerThread; j++) {
        const u32 C = own_element_after_fwd<LOGN, kElemsPerThread>(0, j), X = lds_pos(C);
        const u32 pos = (P ^ (X & 31u)) + (X & ~31u);
        ntt_st<(NTT_NT_FWD & 2) != 0>(x0 + C, ar.canonical(smem[pos]));
      }
    }
  } else {
    __syncthreads();
    for (u32 e = tid; e < (u32)Sh::N; e += Sh::T) ntt_st<(NTT_NT_FWD & 2) != 0>(x + e, ar.canonical(smem[lds_pos(e)]));
  }
}

template <class A, int LOGN>
__device__ __forceinline__ void ntt_fwd_body(const DevMod& dm, const typename A::Tw* tw, u64* x, typename A::V* smem, u32 tid) {
  const A ar(dm);
  // Both arithmetic policies begin with the same polynomial loads; the compiler hoists the first COMMON one above the policy
  // branch of the kernel, where it stands alone before an s_waitcnt vmcnt(0) -- one extra memory round trip per workgroup ahead of
  // the other fifteen requests (r04, ISA listing).  An opaque zero offset per instantiation makes the two branches' addresses
  // distinct values (the pointer keeps its provenance and address space: nttcore.hpp opaque_uniform).
  x = const_cast<u64*>(opaque_uniform(const_cast<const u64*>(x)));
  // (storing the last pass's 2^R-element runs straight from registers was measured 15 % slower than this staged,
  // fully coalesced store; the mirror-image direct LOAD in ntt_inv_body is 25 % faster than staging)
  ntt_fwd_to_lds<A, LOGN>(ar, x, smem, tid, tw, dm.fwd_reduce_mask);
  ntt_fwd_store<A, LOGN>(ar, x, smem, tid);
}

template <int LOGN>
__global__ __launch_bounds__(NttShape<LOGN>::T) void ntt_fwd_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twbase, u64* data, NttPlan plan) {
  using Sh = NttShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const u32 tid = threadIdx.x;
  const u32 poly = blockIdx.x;
  const u32 m = plan_mod(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * Sh::N;
  const MulOp* tw = twbase + (size_t)m * Sh::N;  // kernel argument: known global address space, scalar loads possible
  if (dm.use_f64)
    ntt_fwd_body<ArithD, LOGN>(dm, reinterpret_cast<const double*>(tw), x, reinterpret_cast<double*>(smem_raw), tid);
  else
    ntt_fwd_body<ArithI, LOGN>(dm, tw, x, reinterpret_cast<u64*>(smem_raw), tid);
}

// mul_a / mul_b (both or neither): the transform's input is the pointwise product mul_a (.) mul_b mod q (canonical
// operands) instead of x -- the dyadic multiply that precedes an inverse transform costs no pass of its own
template <class A, int LOGN>
__device__ __forceinline__ void ntt_inv_body(const DevMod& dm, const typename A::Tw* tw, const typename A::Sc& sc, u64* x,
                                             typename A::V* smem, u32 tid, const u64* __restrict__ mul_a = nullptr,
                                             const u64* __restrict__ mul_b = nullptr) {
  using Sh = NttShape<LOGN>;
  const A ar(dm);
  x = const_cast<u64*>(opaque_uniform(const_cast<const u64*>(x)));  // keeps the first load out of the policy branch's common prefix (ntt_fwd_body)
  typename A::V v[kElemsPerThread];
  {  // the first inverse pass consumes runs of 2^R consecutive elements per thread: load them straight into registers
    constexpr int RF = Sh::radix(Sh::NPASS - 1), GF = kElemsPerThread >> RF;
#pragma unroll
    for (int g = 0; g < GF; g++) {
      const size_t at = (size_t)(tid + g * Sh::T) << RF;
      const ulonglong2* src = reinterpret_cast<const ulonglong2*>((mul_b ? mul_a : x) + at);
#pragma unroll
      for (int k = 0; k < (1 << RF); k += 2) {
        typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
        const u64x2_t wv = ntt_ld<(NTT_NT_INV & 1) != 0>(reinterpret_cast<const u64x2_t*>(src + (k >> 1)));
        const ulonglong2 w = make_ulonglong2(wv.x, wv.y);
        v[g * (1 << RF) + k] = ar.from_u64(w.x);
        v[g * (1 << RF) + k + 1] = ar.from_u64(w.y);
      }
      if (mul_b) {
        const ulonglong2* other = reinterpret_cast<const ulonglong2*>(mul_b + at);
#pragma unroll
        for (int k = 0; k < (1 << RF); k += 2) {
          const ulonglong2 w = other[k >> 1];
          v[g * (1 << RF) + k] = ar.mul_var(v[g * (1 << RF) + k], ar.from_u64(w.x));
          v[g * (1 << RF) + k + 1] = ar.mul_var(v[g * (1 << RF) + k + 1], ar.from_u64(w.y));
        }
      }
    }
    InvPasses<A, LOGN, kElemsPerThread, 0, true>::run(ar, v, smem, tid, tw, dm.inv_reduce_mask);
  }
  constexpr int R = Sh::radix(0);
  constexpr int LOW = LOGN - R;
  constexpr int G = kElemsPerThread >> R;
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) ntt_st<(NTT_NT_INV & 2) != 0>(x + elem_index<LOW, R>(tid + g * Sh::T, k), ar.scale_canonical(v[g * (1 << R) + k], sc));
}

// scale_mode: 0 = n^{-1}; 1 = BEHZ epilogue (n^{-1} * t [* (q/q_i)^{-1}]), see DevCtx::intt_scale_*
template <int LOGN>
__global__ __launch_bounds__(NttShape<LOGN>::T) void ntt_inv_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twbase, u64* data, NttPlan plan, int scale_mode) {
  using Sh = NttShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const u32 tid = threadIdx.x;
  const u32 poly = blockIdx.x;
  const u32 m = plan_mod(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * Sh::N;
  const MulOp* tw = twbase + (size_t)m * Sh::N;
  if (dm.use_f64) {
    MulOpD sc = dm.ninv_d;
    if (scale_mode == 1) sc = m < ctx->KK ? ctx->intt_scale_q_d[m] : ctx->intt_scale_bsk_d[m - ctx->KK];
    ntt_inv_body<ArithD, LOGN>(dm, reinterpret_cast<const double*>(tw), sc, x, reinterpret_cast<double*>(smem_raw), tid);
  } else {
    MulOp sc = dm.ninv;
    if (scale_mode == 1) sc = m < ctx->KK ? ctx->intt_scale_q[m] : ctx->intt_scale_bsk[m - ctx->KK];
    ntt_inv_body<ArithI, LOGN>(dm, tw, sc, x, reinterpret_cast<u64*>(smem_raw), tid);
  }
}

// c[op][j][i] = INTT(a[op][i] (.) b[j][i]) for j < nb over the first `nmod` moduli, one workgroup per output polynomial:
// the key-level product pk * u of a public-key encryption (a = NTT(u), b = the public key, nb = 2) and the c1 * s of a
// decryption (a = NTT(c1), b = the secret key, nb = 1)
template <int LOGN>
__global__ __launch_bounds__(NttShape<LOGN>::T) void ntt_inv_dyadic_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twbase,
                                                                           const u64* __restrict__ a, const u64* __restrict__ b, u64* __restrict__ c,
                                                                           u32 nmod, u32 nb, u32 bstride) {
  using Sh = NttShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const u32 tid = threadIdx.x;
  const u32 poly = blockIdx.x;
  const u32 m = poly % nmod, j = (poly / nmod) % nb, op = poly / (nb * nmod);
  const DevMod& dm = ctx->mod[m];
  u64* x = c + (size_t)poly * Sh::N;
  const u64* pa = a + ((size_t)op * nmod + m) * Sh::N;
  const u64* pb = b + ((size_t)j * bstride + m) * Sh::N;  // b: u64[nb][bstride][N] (bstride >= nmod rows per polynomial)
  const MulOp* tw = twbase + (size_t)m * Sh::N;
  if (dm.use_f64)
    ntt_inv_body<ArithD, LOGN>(dm, reinterpret_cast<const double*>(tw), dm.ninv_d, x, reinterpret_cast<double*>(smem_raw), tid, pa, pb);
  else
    ntt_inv_body<ArithI, LOGN>(dm, tw, dm.ninv, x, reinterpret_cast<u64*>(smem_raw), tid, pa, pb);
}

// =====================================================================================
// Paired transforms: one workgroup advances TWO polynomials of the same modulus pass by pass (as the middle kernels of
// kernels_split.hip do): one barrier and one set of twiddle fetches per pass for both, two independent butterfly
// streams per thread, both polynomials' loads in flight together.  2 * N words of LDS; used for N <= 4096, where two such workgroups fit a CU.
// =====================================================================================
template <class A, int LOGN, int PASS>
__device__ __forceinline__ void fwd_passes2(const A& ar, typename A::V (&v)[2][kElemsPerThread], typename A::V* smem, u32 tid,
                                            const typename A::Tw* tw, u32 reduce_mask) {
  using Sh = NttShape<LOGN>;
  constexpr int EPT = kElemsPerThread;
  constexpr int R = Sh::radix(PASS), S0 = Sh::before(PASS), LOW = LOGN - S0 - R, G = EPT >> R;
  if constexpr (PASS > 0) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) v[i][g * (1 << R) + k] = smem[i * Sh::N + lds_pos(elem_index<LOW, R>(tid + g * Sh::T, k))];
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if ((reduce_mask >> PASS) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[i][e] = ar.reduce(v[i][e]);
    }
    if ((reduce_mask >> (PASS + 16)) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[i][e] = ar.reduce(v[i][e]);
    }
  }
  fwd_pass_compute<A, LOGN, EPT, S0, R>(ar, v[0], tid, tw);
  fwd_pass_compute<A, LOGN, EPT, S0, R>(ar, v[1], tid, tw);  // same twiddle addresses: the loads are shared
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) smem[i * Sh::N + lds_pos(elem_index<LOW, R>(tid + g * Sh::T, k))] = v[i][g * (1 << R) + k];
  if constexpr (PASS + 1 < Sh::NPASS) fwd_passes2<A, LOGN, PASS + 1>(ar, v, smem, tid, tw, reduce_mask);
}

template <class A, int LOGN, int PASS>
__device__ __forceinline__ void inv_passes2(const A& ar, typename A::V (&v)[2][kElemsPerThread], typename A::V* smem, u32 tid,
                                            const typename A::Tw* tw, u32 reduce_mask) {
  using Sh = NttShape<LOGN>;
  constexpr int EPT = kElemsPerThread;
  constexpr int FP = Sh::NPASS - 1 - PASS;
  constexpr int R = Sh::radix(FP), LOW = LOGN - Sh::before(FP) - R, G = EPT >> R;
  if constexpr (PASS > 0) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) v[i][g * (1 << R) + k] = smem[i * Sh::N + lds_pos(elem_index<LOW, R>(tid + g * Sh::T, k))];
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    if ((reduce_mask >> PASS) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[i][e] = ar.reduce(v[i][e]);
    }
    if ((reduce_mask >> (PASS + 16)) & 1u) {
#pragma unroll
      for (int e = 0; e < EPT; e++) v[i][e] = ar.reduce(v[i][e]);
    }
  }
  inv_pass_compute<A, LOGN, EPT, LOW, R>(ar, v[0], tid, tw);
  inv_pass_compute<A, LOGN, EPT, LOW, R>(ar, v[1], tid, tw);
  if constexpr (PASS + 1 < Sh::NPASS) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < (1 << R); k++) smem[i * Sh::N + lds_pos(elem_index<LOW, R>(tid + g * Sh::T, k))] = v[i][g * (1 << R) + k];
    inv_passes2<A, LOGN, PASS + 1>(ar, v, smem, tid, tw, reduce_mask);
  }
}

template <class A, int LOGN>
__device__ __forceinline__ void ntt_fwd2_body(const DevMod& dm, const typename A::Tw* tw, u64* x0, u64* x1, typename A::V* smem, u32 tid) {
  using Sh = NttShape<LOGN>;
  const A ar(dm);
  constexpr int R0 = Sh::radix(0), LOW0 = LOGN - R0, G0 = kElemsPerThread >> R0;
  typename A::V v[2][kElemsPerThread];
#pragma unroll
  for (int g = 0; g < G0; g++)
#pragma unroll
    for (int k = 0; k < (1 << R0); k++) {
      const u32 e = elem_index<LOW0, R0>(tid + g * Sh::T, k);
      v[0][g * (1 << R0) + k] = ar.from_u64(x0[e]);
      v[1][g * (1 << R0) + k] = ar.from_u64(x1[e]);
    }
  fwd_passes2<A, LOGN, 0>(ar, v, smem, tid, tw, dm.fwd_reduce_mask);
  __syncthreads();
  for (u32 e = tid; e < (u32)Sh::N; e += Sh::T) {
    x0[e] = ar.canonical(smem[lds_pos(e)]);
    x1[e] = ar.canonical(smem[Sh::N + lds_pos(e)]);
  }
}

template <class A, int LOGN>
__device__ __forceinline__ void ntt_inv2_body(const DevMod& dm, const typename A::Tw* tw, const typename A::Sc& sc, u64* x0, u64* x1,
                                              typename A::V* smem, u32 tid) {
  using Sh = NttShape<LOGN>;
  const A ar(dm);
  typename A::V v[2][kElemsPerThread];
  constexpr int RF = Sh::radix(Sh::NPASS - 1), GF = kElemsPerThread >> RF;
#pragma unroll
  for (int g = 0; g < GF; g++) {
    const ulonglong2* s0 = reinterpret_cast<const ulonglong2*>(x0 + ((size_t)(tid + g * Sh::T) << RF));
    const ulonglong2* s1 = reinterpret_cast<const ulonglong2*>(x1 + ((size_t)(tid + g * Sh::T) << RF));
#pragma unroll
    for (int k = 0; k < (1 << RF); k += 2) {
      const ulonglong2 a = s0[k >> 1], b = s1[k >> 1];
      v[0][g * (1 << RF) + k] = ar.from_u64(a.x);
      v[0][g * (1 << RF) + k + 1] = ar.from_u64(a.y);
      v[1][g * (1 << RF) + k] = ar.from_u64(b.x);
      v[1][g * (1 << RF) + k + 1] = ar.from_u64(b.y);
    }
  }
  inv_passes2<A, LOGN, 0>(ar, v, smem, tid, tw, dm.inv_reduce_mask);
  constexpr int R = Sh::radix(0), LOW = LOGN - R, G = kElemsPerThread >> R;
#pragma unroll
  for (int g = 0; g < G; g++)
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
      const u32 e = elem_index<LOW, R>(tid + g * Sh::T, k);
      x0[e] = ar.scale_canonical(v[0][g * (1 << R) + k], sc);
      x1[e] = ar.scale_canonical(v[1][g * (1 << R) + k], sc);
    }
}

// grid: polys / 2 workgroups; plan.div == 1 and polys % (2 * plan.period) == 0: workgroup w takes the polynomials
// p0 = (w / period) * 2 * period + w % period and p0 + period, which share their modulus
template <int LOGN, bool INVERSE>
__global__ __launch_bounds__(NttShape<LOGN>::T, 2) void ntt_pair_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twbase, u64* data,
                                                                        NttPlan plan, int scale_mode) {
  using Sh = NttShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const u32 tid = threadIdx.x;
  const u32 w = blockIdx.x;
  const u32 p0 = (w / plan.period) * 2 * plan.period + w % plan.period, p1 = p0 + plan.period;
  const u32 m = plan_mod(plan, (w % plan.period) * plan.div);  // (scalar lookup; div == 1 for paired launches)
  const DevMod& dm = ctx->mod[m];
  u64* x0 = data + (size_t)p0 * Sh::N;
  u64* x1 = data + (size_t)p1 * Sh::N;
  const MulOp* tw = twbase + (size_t)m * Sh::N;
  if (dm.use_f64) {
    const double* twd = reinterpret_cast<const double*>(tw);
    double* sm = reinterpret_cast<double*>(smem_raw);
    if constexpr (INVERSE) {
      MulOpD sc = dm.ninv_d;
      if (scale_mode == 1) sc = m < ctx->KK ? ctx->intt_scale_q_d[m] : ctx->intt_scale_bsk_d[m - ctx->KK];
      ntt_inv2_body<ArithD, LOGN>(dm, twd, sc, x0, x1, sm, tid);
    } else {
      ntt_fwd2_body<ArithD, LOGN>(dm, twd, x0, x1, sm, tid);
    }
  } else {
    u64* sm = reinterpret_cast<u64*>(smem_raw);
    if constexpr (INVERSE) {
      MulOp sc = dm.ninv;
      if (scale_mode == 1) sc = m < ctx->KK ? ctx->intt_scale_q[m] : ctx->intt_scale_bsk[m - ctx->KK];
      ntt_inv2_body<ArithI, LOGN>(dm, tw, sc, x0, x1, sm, tid);
    } else {
      ntt_fwd2_body<ArithI, LOGN>(dm, tw, x0, x1, sm, tid);
    }
  }
}

// Kernels that take more than 64 KB of dynamic LDS need the attribute raised once per (device, kernel): the attribute
// belongs to the device's copy of the code object, and a process may move to another device between contexts
// (hipbfv_set_device), so a per-process flag is not enough.
static void allow_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  if (done.insert(std::make_pair(dev, kernel)).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int LOGN>
static hipError_t launch_ntt_t(const DevCtx* ctx, const MulOp* tw, u64* data, size_t polys, const NttPlan& plan, bool inverse, int scale_mode, hipStream_t s) {
  using Sh = NttShape<LOGN>;
  const size_t lds = (size_t)Sh::LDS_WORDS * sizeof(u64);
  if constexpr (LOGN <= 12) {  // measured: +5 % at N = 4096; at N = 8192 (128 KB of LDS, one workgroup per CU) 5 % slower
    // as many polynomials as possible go two per workgroup; the remainder (< 2 * period) one per workgroup below
    const size_t paired = plan.div == 1 ? polys / (2 * (size_t)plan.period) * (2 * (size_t)plan.period) : 0;
    if (paired) {
      allow_dynamic_lds((const void*)ntt_pair_kernel<LOGN, true>, 2 * lds);
      allow_dynamic_lds((const void*)ntt_pair_kernel<LOGN, false>, 2 * lds);
      if (inverse)
        ntt_pair_kernel<LOGN, true><<<dim3((unsigned)(paired / 2)), dim3(Sh::T), 2 * lds, s>>>(ctx, tw, data, plan, scale_mode);
      else
        ntt_pair_kernel<LOGN, false><<<dim3((unsigned)(paired / 2)), dim3(Sh::T), 2 * lds, s>>>(ctx, tw, data, plan, scale_mode);
      if (paired == polys) return hipGetLastError();
      data += paired * Sh::N;
      polys -= paired;  // paired is a multiple of the period: the plan's modulus cycle continues unchanged
    }
  }
  if (inverse) {
    allow_dynamic_lds((const void*)ntt_inv_kernel<LOGN>, lds);
    ntt_inv_kernel<LOGN><<<dim3((unsigned)polys), dim3(Sh::T), lds, s>>>(ctx, tw, data, plan, scale_mode);
  } else {
    allow_dynamic_lds((const void*)ntt_fwd_kernel<LOGN>, lds);
    ntt_fwd_kernel<LOGN><<<dim3((unsigned)polys), dim3(Sh::T), lds, s>>>(ctx, tw, data, plan);
  }
  return hipGetLastError();
}

template <int LOGN>
static hipError_t launch_ntt_inv_dyadic_t(const DevCtx* ctx, const MulOp* tw, const u64* a, const u64* b, u64* c, u32 nmod, u32 nb, u32 bstride,
                                          size_t ops, hipStream_t s) {
  using Sh = NttShape<LOGN>;
  const size_t lds = (size_t)Sh::LDS_WORDS * sizeof(u64);
  allow_dynamic_lds((const void*)ntt_inv_dyadic_kernel<LOGN>, lds);
  ntt_inv_dyadic_kernel<LOGN><<<dim3((unsigned)(ops * nb * nmod)), dim3(Sh::T), lds, s>>>(ctx, tw, a, b, c, nmod, nb, bstride);
  return hipGetLastError();
}
// =====================================================================================
// Ciphertext x plaintext (Evaluator_MultiplyPlain, seal_fhe/src/bfv_evaluator.rs multiply_plain; SEAL
// multiply_plain_normal): the plaintext is lifted and transformed once (plain_lift_kernel + ntt_fwd_kernel, K residue
// polynomials); then ONE kernel per chunk takes every residue polynomial of the ciphertext through transform -> product
// with the plaintext's transform (read from global memory at the point of use, in the register layout the last forward
// pass leaves) -> inverse transform -> n^-1 scale -> store.  The ciphertext crosses HBM twice (in, out) instead of six
// times (transform in place, dyadic product in place, inverse in place), and no copy precedes an out-of-place call.
// (Holding the plaintext's transform in registers across the polynomials of one workgroup was tried first: with two
// transforms' worth of live values the kernel needs > 128 registers at N = 16384 -- 536 bytes of scratch -- and the
// loop-invariant twiddle / address hoisting has to be fought; one polynomial per workgroup needs neither.)
// =====================================================================================
template <class A, int LOGN>
__device__ __forceinline__ void ct_plain_body(const DevMod& dm, const typename A::Tw* twf, const typename A::Tw* twi, const typename A::Sc& ninv,
                                              const u64* __restrict__ pn, const u64* x, u64* y, typename A::V* smem, u32 tid) {
  using Sh = NttShape<LOGN>;
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  constexpr int EPT = kElemsPerThread;
  constexpr int R0 = Sh::radix(0), LOW0 = LOGN - R0, G0 = EPT >> R0;
  const A ar(dm);
  typename A::V v[EPT];
#pragma unroll
  for (int g = 0; g < G0; g++)
#pragma unroll
    for (int k = 0; k < (1 << R0); k++) v[g * (1 << R0) + k] = ar.from_u64(x[elem_index<LOW0, R0>(tid + g * Sh::T, k)]);
  FwdPasses<A, LOGN, EPT, 0, true>::run(ar, v, smem, tid, twf, dm.fwd_reduce_mask);
  // the last forward pass leaves element ((tid + g*T) << RF) | k in v[g * 2^RF + k]: runs of 2^RF consecutive coefficients
  constexpr int RF = Sh::radix(Sh::NPASS - 1), GF = EPT >> RF;
  static_assert(RF >= 1, "pairs of consecutive coefficients");
  // fence: the inverse transform's twiddle loads and the plaintext loads must not be scheduled up into the forward transform
  // (the two transforms' live values together do not fit 128 registers: 140 at N = 8192, scratch at N = 16384)
  const typename A::Tw* twi_c = opaque_uniform(twi);
  const u64* pn_c = opaque_uniform(pn);
#pragma unroll
  for (int g = 0; g < GF; g++) {
    const u64x2_t* src = reinterpret_cast<const u64x2_t*>(pn_c + ((size_t)(tid + g * Sh::T) << RF));
#pragma unroll
    for (int k = 0; k < (1 << RF); k += 2) {
      const u64x2_t w = src[k >> 1];
      v[g * (1 << RF) + k] = ar.mul_var(v[g * (1 << RF) + k], ar.from_u64(w.x));
      v[g * (1 << RF) + k + 1] = ar.mul_var(v[g * (1 << RF) + k + 1], ar.from_u64(w.y));
    }
  }
  __syncthreads();  // slower wavefronts may still be reading LDS in the forward transform's last pass
  InvPasses<A, LOGN, EPT, 0, true>::run(ar, v, smem, tid, twi_c, dm.inv_reduce_mask);
#pragma unroll
  for (int g = 0; g < G0; g++)
#pragma unroll
    for (int k = 0; k < (1 << R0); k++) y[elem_index<LOW0, R0>(tid + g * Sh::T, k)] = ar.scale_canonical(v[g * (1 << R0) + k], ninv);
}

// grid: ops * size * K workgroups; pn = the plaintexts' transforms u64[ops or 1][K][N] (pnstride = 0: one for every op)
// POLICY_D: one instantiation per arithmetic policy (the integer body needs 152 registers, the FP64 one 126 = two
// workgroups per CU at N = 8192); a workgroup whose residue has the other policy leaves at once
template <int LOGN, bool POLICY_D>
__global__ __launch_bounds__(NttShape<LOGN>::T) void ct_plain_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                     const MulOp* __restrict__ twi_base, const u64* __restrict__ pn, size_t pnstride,
                                                                     const u64* in, u64* out, u32 size) {
  using Sh = NttShape<LOGN>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const u32 tid = threadIdx.x;
  const u32 K = ctx->K;
  const u32 row = blockIdx.x;  // (op * size + c) * K + i
  const u32 i = row % K, op = row / (K * size);
  const DevMod& dm = ctx->mod[i];
  if ((dm.use_f64 != 0) != POLICY_D) return;
  const u64* x = in + (size_t)row * Sh::N;
  u64* y = out + (size_t)row * Sh::N;
  const u64* p = pn + (size_t)op * pnstride + (size_t)i * Sh::N;
  const MulOp* twf = twf_base + (size_t)i * Sh::N;
  const MulOp* twi = twi_base + (size_t)i * Sh::N;
  if constexpr (POLICY_D)
    ct_plain_body<ArithD, LOGN>(dm, reinterpret_cast<const double*>(twf), reinterpret_cast<const double*>(twi), dm.ninv_d, p, x, y,
                                reinterpret_cast<double*>(smem_raw), tid);
  else
    ct_plain_body<ArithI, LOGN>(dm, twf, twi, dm.ninv, p, x, y, reinterpret_cast<u64*>(smem_raw), tid);
}

template <int LOGN>
static hipError_t launch_ct_plain_t(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 K, bool any_d, bool any_i, const u64* pn, size_t pnstride,
                                    const u64* in, u64* out, u32 size, size_t ops, hipStream_t s) {
  using Sh = NttShape<LOGN>;
  const size_t lds = (size_t)Sh::LDS_WORDS * sizeof(u64);
  if (any_d) {
    allow_dynamic_lds((const void*)ct_plain_kernel<LOGN, true>, lds);
    ct_plain_kernel<LOGN, true><<<dim3((unsigned)(ops * size * K)), dim3(Sh::T), lds, s>>>(ctx, twf, twi, pn, pnstride, in, out, size);
  }
  if (any_i) {
    allow_dynamic_lds((const void*)ct_plain_kernel<LOGN, false>, lds);
    ct_plain_kernel<LOGN, false><<<dim3((unsigned)(ops * size * K)), dim3(Sh::T), lds, s>>>(ctx, twf, twi, pn, pnstride, in, out, size);
  }
  return hipGetLastError();
}
// out u64[ops][size][K][N] = INTT(NTT(in) (.) pn); pn u64[ops or 1][K][N] = the lifted plaintexts' transforms; out may be in.
// any_d / any_i: whether any of the K data primes takes the FP64 / the integer policy (DevMod::use_f64 on the host's copy).
// hipErrorNotSupported: no instantiation for this degree (the caller falls back to the separate kernels).
hipError_t launch_ct_plain(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, u32 K, bool any_d, bool any_i, const u64* pn, size_t pnstride,
                           const u64* in, u64* out, u32 size, size_t ops, hipStream_t s) {
  if (ops == 0) return hipSuccess;
  switch (logn) {
    case 10: return launch_ct_plain_t<10>(ctx, twf, twi, K, any_d, any_i, pn, pnstride, in, out, size, ops, s);
    case 11: return launch_ct_plain_t<11>(ctx, twf, twi, K, any_d, any_i, pn, pnstride, in, out, size, ops, s);
    case 12: return launch_ct_plain_t<12>(ctx, twf, twi, K, any_d, any_i, pn, pnstride, in, out, size, ops, s);
    case 13: return launch_ct_plain_t<13>(ctx, twf, twi, K, any_d, any_i, pn, pnstride, in, out, size, ops, s);
    case 14: return launch_ct_plain_t<14>(ctx, twf, twi, K, any_d, any_i, pn, pnstride, in, out, size, ops, s);
    default: break;
  }
  return hipErrorNotSupported;
}

// c u64[ops][nb][nmod][N] = INTT(a[op][i] (.) b[j][i]); a u64[ops][nmod][N], b u64[nb][bstride][N] (N <= 16384)
hipError_t launch_ntt_inv_dyadic(const DevCtx* ctx, const MulOp* tw_inv, u32 logn, const u64* a, const u64* b, u64* c, u32 nmod, u32 nb, u32 bstride,
                                 size_t ops, hipStream_t s) {
  if (ops == 0) return hipSuccess;
  switch (logn) {
    case 10: return launch_ntt_inv_dyadic_t<10>(ctx, tw_inv, a, b, c, nmod, nb, bstride, ops, s);
    case 11: return launch_ntt_inv_dyadic_t<11>(ctx, tw_inv, a, b, c, nmod, nb, bstride, ops, s);
    case 12: return launch_ntt_inv_dyadic_t<12>(ctx, tw_inv, a, b, c, nmod, nb, bstride, ops, s);
    case 13: return launch_ntt_inv_dyadic_t<13>(ctx, tw_inv, a, b, c, nmod, nb, bstride, ops, s);
    case 14: return launch_ntt_inv_dyadic_t<14>(ctx, tw_inv, a, b, c, nmod, nb, bstride, ops, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ntt(const DevCtx* ctx, const MulOp* tw, u32 logn, u64* data, size_t polys, const NttPlan& plan, bool inverse, int scale_mode, hipStream_t s) {
  if (polys == 0) return hipSuccess;
  switch (logn) {
    case 10: return launch_ntt_t<10>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
    case 11: return launch_ntt_t<11>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
    case 12: return launch_ntt_t<12>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
    case 13: return launch_ntt_t<13>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
    case 14: return launch_ntt_t<14>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
    case 15: return launch_ntt_split(ctx, tw, logn, data, polys, plan, inverse, scale_mode, s);  // 256 KB polynomials: two kernels
    default: return hipErrorInvalidValue;
  }
}

// =====================================================================================
// Coefficient-parallel kernels.  Thread = one coefficient index k of one (op, poly); the
// residues of that coefficient are N words apart, so every load/store is a 512-byte
// wavefront transaction.
// =====================================================================================

constexpr int kCoefThreads = 256;

// ---- BEHZ step 1+2: base q -> q u Bsk with Montgomery reduction of the q-overflow ----
// in0: u64[ops][sa][K][N], in1: u64[ops][sb][K][N]  (ciphertext polynomials, coefficient form)
// out: u64[ops][sa+sb][K+S][N]                      (q residues copied, then Bsk residues)
template <int KMAX>
__global__ __launch_bounds__(kCoefThreads) void behz_extend_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in0, u32 sa,
                                                                   const u64* __restrict__ in1, u32 sb, u64* __restrict__ out) {
  const u32 n = ctx->n, K = ctx->K, S = ctx->S;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 poly = blockIdx.y;  // global poly index: op * (sa+sb) + p
  const u32 op = poly / (sa + sb), p = poly % (sa + sb);
  const u64* src = p < sa ? in0 + ((size_t)op * sa + p) * K * n : in1 + ((size_t)op * sb + (p - sa)) * K * n;
  u64* dst = out + (size_t)poly * (K + S) * n;
  if (k >= n) return;
  u64 x[KMAX], e[KMAX + 2];
  // branch-free requests, all in flight together: rows i >= K re-read row K - 1 and are discarded (behind `i < K` branches
  // every load got its own basic block and an s_waitcnt vmcnt(0): KMAX dependent round trips -- r04, kernels_split.hip mul_head)
#pragma unroll
  for (int i = 0; i < KMAX; i++) x[i] = src[(size_t)((u32)i < K ? (u32)i : K - 1) * n + k];
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    if ((u32)i < K) dst[(size_t)i * n + k] = x[i];
  }
  if (ctx->aux_f64) {  // the library's own FP64 auxiliary base (context.cpp): exact double arithmetic, canonical words out
    double xd[KMAX], ed[KMAX + 2];
#pragma unroll
    for (int i = 0; i < KMAX; i++) xd[i] = (u32)i < K ? ArithD::from_u64(x[i]) : 0.0;
    behz_extend_coeff_d<KMAX>(ctx, xd, ed);
#pragma unroll
    for (int j = 0; j < KMAX + 2; j++)
      if ((u32)j < S) dst[(size_t)(K + j) * n + k] = ArithD(ctx->mod[ctx->KK + j]).to_u64(ed[j]);
    return;
  }
  behz_extend_coeff<KMAX>(ctx, x, e);
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++)
    if ((u32)j < S) dst[(size_t)(K + j) * n + k] = e[j];
}

// ---- BEHZ step 4: dyadic tensor product per residue: d_p = sum_{i+j=p} a_i * b_j ----
// ext: u64[ops][sa+sb][R][N] (NTT form, a polys then b polys), D: u64[ops][sa+sb-1][R][N]
__global__ __launch_bounds__(kCoefThreads) void tensor_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ext, u32 sa, u32 sb,
                                                              u64* __restrict__ D) {
  const u32 n = ctx->n, R = ctx->K + ctx->S;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 r = blockIdx.y, op = blockIdx.z;
  if (k >= n) return;
  const u32 m = r < ctx->K ? r : ctx->KK + (r - ctx->K);
  const DevMod& pm = ctx->mod[m];
  const u64* A = ext + ((size_t)op * (sa + sb)) * R * n + (size_t)r * n + k;
  u64* d = D + ((size_t)op * (sa + sb - 1)) * R * n + (size_t)r * n + k;
  if (sa == 2 && sb == 2) {
    const u64 a0 = A[0], a1 = A[(size_t)R * n], b0 = A[(size_t)2 * R * n], b1 = A[(size_t)3 * R * n];
    d[0] = reduce128_fast((u128)a0 * b0, pm);
    d[(size_t)R * n] = reduce128_fast((u128)a0 * b1 + (u128)a1 * b0, pm);
    d[(size_t)2 * R * n] = reduce128_fast((u128)a1 * b1, pm);
    return;
  }
  for (u32 p = 0; p + 1 < sa + sb; p++) {
    u128 acc = 0;
    for (u32 i = 0; i < sa; i++) {
      if (p < i || p - i >= sb) continue;
      acc += (u128)A[(size_t)i * R * n] * A[(size_t)(sa + p - i) * R * n];
    }
    d[(size_t)p * R * n] = reduce128(acc, pm);
  }
}

// ---- BEHZ steps 7+8: fast_floor (q u Bsk -> Bsk) then Shenoy-Kumaresan (Bsk -> q) ----
// D: u64[npoly][K+S][N] after the scaled inverse NTT; out: u64[npoly][K][N]
template <int KMAX>
__global__ __launch_bounds__(kCoefThreads) void behz_floor_sk_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ D, u64* __restrict__ out) {
  const u32 n = ctx->n, K = ctx->K, S = ctx->S;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 poly = blockIdx.y;
  if (k >= n) return;
  const u64* d = D + (size_t)poly * (K + S) * n;
  u64 y[KMAX], xb[KMAX + 2], r[KMAX];
  // branch-free requests, all in flight together (rows beyond K / S re-read the last valid row and are ignored below)
#pragma unroll
  for (int i = 0; i < KMAX; i++) y[i] = d[(size_t)((u32)i < K ? (u32)i : K - 1) * n + k];  // already x * t * (q/q_i)^{-1} mod q_i
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) xb[j] = d[(size_t)(K + ((u32)j < S ? (u32)j : S - 1)) * n + k];
  if (ctx->aux_f64) {
    double yd[KMAX], xd[KMAX + 2];
#pragma unroll
    for (int i = 0; i < KMAX; i++) yd[i] = (u32)i < K ? ArithD::from_u64(y[i]) : 0.0;
#pragma unroll
    for (int j = 0; j < KMAX + 2; j++) xd[j] = (u32)j < S ? ArithD::from_u64(xb[j]) : 0.0;
    behz_floor_sk_coeff_d<KMAX>(ctx, yd, xd, r);
  } else {
    behz_floor_sk_coeff<KMAX>(ctx, y, xb, r);
  }
  u64* o = out + (size_t)poly * K * n;
#pragma unroll
  for (int i = 0; i < KMAX; i++)
    if ((u32)i < K) o[(size_t)i * n + k] = r[i];
}

// ---- key switching ----
// target: u64[ops][K][N] with op stride `tstride` words; T: u64[ops][KK][K][N]
__global__ __launch_bounds__(kCoefThreads) void ks_decompose_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ target,
                                                                    size_t tstride, u64* __restrict__ T) {
  const u32 n = ctx->n, K = ctx->K, KK = ctx->KK;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 J = blockIdx.y, op = blockIdx.z;
  if (k >= n) return;
  const u64 x = target[(size_t)op * tstride + (size_t)J * n + k];
  const u64 qJ = ctx->mod[J].q;
  for (u32 I = 0; I < KK; I++) {
    const DevMod& mI = ctx->mod[I];
    T[(((size_t)op * KK + I) * K + J) * n + k] = qJ <= mI.q ? x : reduce64(x, mI);
  }
}

// ACC[op][c][I][k] = sum_J T[op][I][J][k] * key[J][c][I][k]   (128-bit lazy sum, one reduction)
__global__ __launch_bounds__(kCoefThreads) void ks_mac_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ T,
                                                              const u64* __restrict__ key, u64* __restrict__ ACC, KeyMap km) {
  const u32 n = ctx->n, K = ctx->K, KK = ctx->KK;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  u32 I = blockIdx.y, op = blockIdx.z;
  if (km.keys) {  // per-item keys: walk position -> {item, key index} (kernels.hpp KeyMap)
    const uint2 m = km.order[op];
    op = m.x;
    key = km.keys[m.y];
  }
  if (k >= n) return;
  const DevMod& mI = ctx->mod[I];
  u128 a0 = 0, a1 = 0;
  // four digits per trip, their twelve words requested together and branch-free (digits beyond K re-read digit K - 1 and count
  // as zero): one digit per trip was one dependent memory round trip per digit -- 15 in a row at n = 32768, and the whole
  // latency of this kernel when one ciphertext is relinearised (r04)
  for (u32 J0 = 0; J0 < K; J0 += 4) {
    u64 tv[4], ka[4], kb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u32 J = J0 + (u32)u < K ? J0 + (u32)u : K - 1;
      tv[u] = T[(((size_t)op * KK + I) * K + J) * n + k];
      ka[u] = as_global(key)[(((size_t)J * 2 + 0) * KK + I) * n + k];  // (key may be a loaded pointer: name the address space)
      kb[u] = as_global(key)[(((size_t)J * 2 + 1) * KK + I) * n + k];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u64 tu = J0 + (u32)u < K ? tv[u] : 0;
      a0 += (u128)tu * ka[u];
      a1 += (u128)tu * kb[u];
    }
  }
  ACC[(((size_t)op * 2 + 0) * KK + I) * n + k] = reduce128(a0, mI);
  ACC[(((size_t)op * 2 + 1) * KK + I) * n + k] = reduce128(a1, mI);
}

// out[op][c][J] = base[op][c][J] + (ACC[op][c][J] - round-fix(ACC[op][c][sp])) * q_sp^{-1}
// base: op stride bstride words, poly stride K*N; base_mask bit c = 0 means "treat base poly c as zero"
__global__ __launch_bounds__(kCoefThreads) void ks_moddown_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ ACC,
                                                                  const u64* __restrict__ base, size_t bstride, u32 base_mask,
                                                                  const u64* __restrict__ extra, u64* __restrict__ out) {
  const u32 n = ctx->n, K = ctx->K, KK = ctx->KK;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 c = blockIdx.y, op = blockIdx.z;
  if (k >= n) return;
  const DevMod& sp = ctx->mod[KK - 1];
  const u64* acc = ACC + ((size_t)op * 2 + c) * KK * n;
  const u64 tl = add_mod(acc[(size_t)(KK - 1) * n + k], ctx->qsp_half, sp.q);
  const bool has_base = ((base_mask >> c) & 1u) != 0;
  // four data primes per trip: their accumulator, base and addend words are requested together and unconditionally (an absent
  // operand reads the first accumulator word -- one cached line per wavefront -- and is ignored; primes beyond K re-read K - 1)
  for (u32 J0 = 0; J0 < K; J0 += 4) {
    u64 av[4], bw[4], ex[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u32 J = J0 + (u32)u < K ? J0 + (u32)u : K - 1;
      av[u] = acc[(size_t)J * n + k];
      bw[u] = *(has_base ? base + (size_t)op * bstride + ((size_t)c * K + J) * n + k : acc);
      ex[u] = *(extra ? extra + (((size_t)op * 2 + c) * K + J) * n + k : acc);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u32 J = J0 + (u32)u;
      if (J < K) {
        const DevMod& mj = ctx->mod[J];
        u64 tk = sp.q > mj.q ? reduce64(tl, mj) : tl;
        tk = sub_mod(tk, ctx->qsp_half_mod_q[J], mj.q);
        u64 d = sub_mod(av[u], tk, mj.q);
        d = mul_shoup(d, ctx->inv_qsp_mod_q[J], mj.q);
        u64 b = has_base ? bw[u] : 0;
        b = extra ? add_mod(b, ex[u], mj.q) : b;
        out[(((size_t)op * 2 + c) * K + J) * n + k] = add_mod(b, d, mj.q);
      }
    }
  }
}

// ---- Evaluator_ModSwitchToNext (BFV): SEAL RNSTool::divide_and_round_q_last_inplace on every polynomial ----
// in: u64[npoly][K][N] at this level; out: u64[npoly][K-1][N] at the next level
__global__ __launch_bounds__(kCoefThreads) void mod_switch_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 poly = blockIdx.y;
  if (k >= n) return;
  const DevMod& ml = ctx->mod[K - 1];
  const u64* x = in + (size_t)poly * K * n;
  const u64 tl = add_mod(x[(size_t)(K - 1) * n + k], ctx->ms_half, ml.q);
  for (u32 i = 0; i + 1 < K; i++) {
    const DevMod& mi = ctx->mod[i];
    u64 tk = ml.q > mi.q ? reduce64(tl, mi) : tl;
    tk = sub_mod(tk, ctx->ms_half_mod_q[i], mi.q);
    const u64 d = sub_mod(x[(size_t)i * n + k], tk, mi.q);
    out[((size_t)poly * (K - 1) + i) * n + k] = mul_shoup(d, ctx->ms_inv_last_mod_q[i], mi.q);
  }
}

// ---- Galois automorphism x -> x^g in coefficient form (gather form) ----
// in/out: u64[npoly][K][N]; ginv = g^{-1} mod 2N
__global__ __launch_bounds__(kCoefThreads) void galois_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out, u32 ginv) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 o = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 poly = blockIdx.y;
  if (o >= n) return;
  const u32 kk = (u32)(((u64)o * ginv) & (2 * n - 1));
  const bool negate = kk >= n;
  const u32 src = negate ? kk - n : kk;
  for (u32 i = 0; i < K; i++) {
    const u64 v = in[((size_t)poly * K + i) * n + src];
    out[((size_t)poly * K + i) * n + o] = negate ? neg_mod(v, ctx->mod[i].q) : v;
  }
}

// The same permutation with the whole residue polynomial staged in LDS (N <= 16384): global loads and stores are 16 bytes
// per lane and fully coalesced, the gather happens in LDS.  Even- and odd-indexed coefficients live in separate halves
// (position (s >> 1) + (s & 1) * N/2): a lane's two outputs 2j, 2j+1 come from sources 2j*ginv (even) and 2j*ginv + ginv
// (odd), so across the lanes of a wavefront each of the two LDS reads walks one half with the odd stride ginv --
// conflict-free -- and the staging writes are contiguous in each half.
constexpr int kGaloisThreads = 1024;
template <int LOGN>
__global__ __launch_bounds__(kGaloisThreads) void galois_lds_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out, u32 ginv) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  constexpr u32 N = 1u << LOGN, H = N / 2;
  __shared__ u64 smem[N];
  const u32 tid = threadIdx.x;
  const size_t row = blockIdx.x;  // residue polynomial: u64[npoly * K][N]
  const u64 q = ctx->mod[row % ctx->K].q;
  const u64x2_t* src = reinterpret_cast<const u64x2_t*>(in + row * N);
  u64x2_t* dst = reinterpret_cast<u64x2_t*>(out + row * N);
#pragma unroll
  for (u32 j = tid; j < H; j += kGaloisThreads) {
    const u64x2_t w = __builtin_nontemporal_load(src + j);
    smem[j] = w.x;
    smem[H + j] = w.y;
  }
  __syncthreads();
#pragma unroll
  for (u32 j = tid; j < H; j += kGaloisThreads) {
    const u32 s0 = (2u * j * ginv) & (2u * N - 1u);  // even
    const u32 s1 = (s0 + ginv) & (2u * N - 1u);       // odd
    const u64 v0 = smem[(s0 & (N - 1u)) >> 1], v1 = smem[H + ((s1 & (N - 1u)) >> 1)];
    u64x2_t r;
    r.x = s0 >= N ? neg_mod(v0, q) : v0;
    r.y = s1 >= N ? neg_mod(v1, q) : v1;
    dst[j] = r;
  }
}

// ---- element-wise ciphertext ops ----
// mode 0: a+b, 1: a-b, 2: -a.  polys laid out u64[npoly][K][N]; two adjacent coefficients per thread (16-byte accesses)
__global__ __launch_bounds__(kCoefThreads) void eltwise_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ a, const u64* __restrict__ b,
                                                               u64* __restrict__ out, int mode) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = 2u * (blockIdx.x * kCoefThreads + threadIdx.x);
  const u32 res = blockIdx.y;  // global residue-poly index
  if (k >= n) return;
  const u64 q = ctx->mod[res % K].q;
  const size_t off = (size_t)res * n + k;
  const u64x2_t x = *reinterpret_cast<const u64x2_t*>(a + off);
  u64x2_t r;
  if (mode == 2) {
    r.x = neg_mod(x.x, q);
    r.y = neg_mod(x.y, q);
  } else {
    const u64x2_t y = *reinterpret_cast<const u64x2_t*>(b + off);
    r.x = mode == 0 ? add_mod(x.x, y.x, q) : sub_mod(x.x, y.x, q);
    r.y = mode == 0 ? add_mod(x.y, y.y, q) : sub_mod(x.y, y.y, q);
  }
  *reinterpret_cast<u64x2_t*>(out + off) = r;
}

// c0 +/-= round(q/t * m)  (SEAL multiply_add_plain_with_scaling_variant); ct u64[ops][size][K][N]
// plain u64[ops or 1][N] (values < t, zero padded); pstride = 0 broadcasts one plaintext
__global__ __launch_bounds__(kCoefThreads) void plain_addsub_kernel(const DevCtx* __restrict__ ctx, u64* __restrict__ ct, size_t ctstride,
                                                                    const u64* __restrict__ plain, size_t pstride, int sub) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (k >= n) return;
  const u64 m = plain[(size_t)op * pstride + k];
  const u64 t = ctx->t;
  const u128 num = (u128)m * ctx->q_mod_t + ctx->t_half_up;
  const u64 fix = (u64)(num / t);
  for (u32 i = 0; i < K; i++) {
    const DevMod& qm = ctx->mod[i];
    const u64 v = reduce128((u128)m * ctx->q_div_t_mod_q[i] + fix, qm);
    u64* d = ct + (size_t)op * ctstride + (size_t)i * n + k;
    *d = sub ? sub_mod(*d, v, qm.q) : add_mod(*d, v, qm.q);
  }
}

// lift plaintext coefficients (mod t, centred) to every q_i: out u64[ops][K][N]
// nonzero[op] = number of non-zero coefficients of plaintext op (must be zeroed by the caller)
__global__ __launch_bounds__(kCoefThreads) void plain_count_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ plain, size_t pstride,
                                                                   u32* __restrict__ nonzero) {
  const u32 n = ctx->n;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  const bool nz = k < n && plain[(size_t)op * pstride + k] != 0;
  const unsigned long long ballot = __ballot(nz);
  if ((threadIdx.x & 63u) == 0 && ballot) atomicAdd(&nonzero[op], (u32)__popcll(ballot));
}

// nonzero: optional per-op non-zero counts.  SEAL's multiply_plain has a monomial shortcut which, when every q_i
// exceeds t (fast plain lift), multiplies by the coefficient AS IS -- without the centred lift of upper-half values
// (SEAL evaluator.cpp multiply_plain_normal: "no need to adjust the monomial").  Lifting such a plaintext unadjusted
// makes the general transform-domain product bit-identical to that shortcut.
__global__ __launch_bounds__(kCoefThreads) void plain_lift_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ plain, size_t pstride,
                                                                  u64* __restrict__ out, const u32* __restrict__ nonzero) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 op = blockIdx.y;
  if (k >= n) return;
  const u64 m = plain[(size_t)op * pstride + k];
  const bool mono_as_is = nonzero && ctx->fast_plain_lift && nonzero[op] == 1;
  const bool upper = m >= ctx->t_half_up && !mono_as_is;
  for (u32 i = 0; i < K; i++) {
    const DevMod& qm = ctx->mod[i];
    u64 v;
    if (!upper)
      v = reduce64(m, qm);
    else
      v = neg_mod(reduce64(ctx->t - m, qm), qm.q);
    out[((size_t)op * K + i) * n + k] = v;
  }
}

// x[op][p][i][k] = x * pl[op or 0][i][k] mod q_i (both NTT form)
__global__ __launch_bounds__(kCoefThreads) void dyadic_plain_kernel(const DevCtx* __restrict__ ctx, u64* __restrict__ x, u32 size,
                                                                    const u64* __restrict__ pl, size_t plstride) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 i = blockIdx.y, op = blockIdx.z;
  if (k >= n) return;
  const DevMod& qm = ctx->mod[i];
  const u64 pv = pl[(size_t)op * plstride + (size_t)i * n + k];
  for (u32 p = 0; p < size; p++) {
    u64* d = x + (((size_t)op * size + p) * K + i) * n + k;
    *d = mul_mod(*d, pv, qm);
  }
}

// negacyclic multiply by the monomial coeff * x^e (SEAL negacyclic_multiply_poly_mono_coeffmod)
// coeff_rns: u64[K] per-prime scalar (device)
__global__ __launch_bounds__(kCoefThreads) void mono_mul_kernel(const DevCtx* __restrict__ ctx, const u64* __restrict__ in, u64* __restrict__ out,
                                                                const u64* __restrict__ coeff_rns, u32 e) {
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = blockIdx.x * kCoefThreads + threadIdx.x;
  const u32 res = blockIdx.y;
  if (k >= n) return;
  const u32 i = res % K;
  const DevMod& qm = ctx->mod[i];
  u64 v = mul_mod(in[(size_t)res * n + k], coeff_rns[i], qm);
  u32 idx = k + e;
  if (idx >= n) {
    idx -= n;
    v = neg_mod(v, qm.q);
  }
  out[(size_t)res * n + idx] = v;
}

// flags[op] |= 1 if any word of polys 1..size-1 of ciphertext op is non-zero (transparent check)
__global__ __launch_bounds__(kCoefThreads) void nonzero_tail_kernel(const u64* __restrict__ ct, size_t words_per_ct, size_t skip_words,
                                                                    u32* __restrict__ flags) {
  const u32 op = blockIdx.y;
  const u64* p = ct + (size_t)op * words_per_ct + skip_words;
  const size_t len = words_per_ct - skip_words;
  bool nz = false;
  for (size_t i = (size_t)blockIdx.x * kCoefThreads + threadIdx.x; i < len; i += (size_t)gridDim.x * kCoefThreads) nz |= p[i] != 0;
  if (__any(nz) && (threadIdx.x & 63) == 0) atomicOr(&flags[op], 1u);
}

// Transparent-result watch of the batched path (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT, seal_fhe/build.rs:46-66): one
// workgroup per ciphertext; *status = min(*status, first_item + op) for every op whose polynomials 1.. are all zero.
// A genuine ciphertext has a non-zero word among the first kCoefThreads of c1, so the common case reads 2 KB per item
// and leaves after one round; only an (almost) all-zero tail is scanned in full.
__global__ __launch_bounds__(kCoefThreads) void transparent_watch_kernel(const u64* __restrict__ ct, size_t words_per_ct, size_t skip_words,
                                                                         u32 first_item, u32* __restrict__ status) {
  const u32 op = blockIdx.x;
  const u64* p = ct + (size_t)op * words_per_ct + skip_words;
  const size_t len = words_per_ct - skip_words;
  for (size_t base = 0; base < len; base += kCoefThreads) {
    const size_t i = base + threadIdx.x;
    const bool nz = i < len && p[i] != 0;
    if (__syncthreads_or(nz)) return;
  }
  if (threadIdx.x == 0) atomicMin(status, first_item + op);
}

// The graph executor's forms.  (1) the outputs of one n-ary sum launch, through its own descriptor table: workgroup (output, item).
__global__ __launch_bounds__(kCoefThreads) void transparent_watch_nary_kernel(const DevCtx* __restrict__ ctx, const NaryOut* __restrict__ outs, u32 batch,
                                                                              u32* __restrict__ status) {
  const NaryOut o = outs[blockIdx.x / batch];
  const u32 b = blockIdx.x % batch;
  const size_t poly = (size_t)ctx->K * ctx->n;
  const auto p = as_global((const u64*)(o.out + (size_t)b * o.size * poly + poly));
  const size_t len = (size_t)(o.size - 1) * poly;
  for (size_t base = 0; base < len; base += kCoefThreads) {
    const size_t i = base + threadIdx.x;
    const bool nz = i < len && p[i] != 0;
    if (__syncthreads_or(nz)) return;
  }
  if (threadIdx.x == 0) atomicMin(status, b);
}
// (2) a plaintext that is identically zero makes SEAL's multiply_plain fail (transparent product); a sum of products kept in
// the transform domain never materialises the single products, so the zero plaintext itself raises the status: nonzero[op] is
// plain_count_kernel's count of non-zero coefficients, item = first_item + op * item_step.
__global__ void zero_plain_watch_kernel(const u32* __restrict__ nonzero, u32 first_item, u32 item_step, u32 count, u32* __restrict__ status) {
  const u32 op = blockIdx.x * blockDim.x + threadIdx.x;
  if (op < count && nonzero[op] == 0) atomicMin(status, first_item + op * item_step);
}

// The handle-level form of the same check: ONE ciphertext, the verdict (1 = not transparent) written straight into a word of
// pinned host memory the device can address -- the caller only has to synchronise its stream, no memset, no copy back.
__global__ __launch_bounds__(kCoefThreads) void transparent_flag_kernel(const u64* __restrict__ ct, size_t words, size_t skip_words,
                                                                       volatile u32* __restrict__ host_flag) {
  for (size_t base = skip_words; base < words; base += kCoefThreads) {
    const size_t i = base + threadIdx.x;
    const bool nz = i < words && ct[i] != 0;
    if (__syncthreads_or(nz)) {
      if (threadIdx.x == 0) *host_flag = 1u;
      return;
    }
  }
  if (threadIdx.x == 0) *host_flag = 0u;
}

// ---- handle-level calls combined into one batched launch (capi.cpp Combiner) ----
// Each host thread's operands and results are separate device buffers; `table` (pinned host memory the device can address)
// lists them.  gather: stage[item][0..words) = *table[item]; scatter: *table[item] = stage[item][..]; 16-byte accesses.
__global__ __launch_bounds__(kCoefThreads) void gather_items_kernel(const u64* const* __restrict__ table, u64* __restrict__ stage, size_t words) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const auto src = as_global(reinterpret_cast<const u64x2_t*>(table[blockIdx.y]));
  u64x2_t* dst = reinterpret_cast<u64x2_t*>(stage + (size_t)blockIdx.y * words);
  for (size_t i = (size_t)blockIdx.x * kCoefThreads + threadIdx.x; i < words / 2; i += (size_t)gridDim.x * kCoefThreads) dst[i] = src[i];
}
__global__ __launch_bounds__(kCoefThreads) void scatter_items_kernel(const u64* __restrict__ stage, u64* const* __restrict__ table, size_t words) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const u64x2_t* src = reinterpret_cast<const u64x2_t*>(stage + (size_t)blockIdx.y * words);
  const auto dst = as_global(reinterpret_cast<u64x2_t*>(table[blockIdx.y]));
  for (size_t i = (size_t)blockIdx.x * kCoefThreads + threadIdx.x; i < words / 2; i += (size_t)gridDim.x * kCoefThreads) dst[i] = src[i];
}
// words from pinned, device-addressable host memory into device memory, ordered on the stream like any kernel (the graph
// executor's descriptor tables: a hipMemcpyAsync on the caller's stream drains it first when that stream is the null stream)
__global__ __launch_bounds__(kCoefThreads) void copy_words_kernel(const u64* __restrict__ src, u64* __restrict__ dst, size_t words) {
  for (size_t i = (size_t)blockIdx.x * kCoefThreads + threadIdx.x; i < words; i += (size_t)gridDim.x * kCoefThreads) dst[i] = src[i];
}
// add / sub of size-2 ciphertexts straight through the pointer tables (no staging: the operation is one pass anyway):
// *out[item] = *a[item] +/- *b[item]; grid (N/2/256, 2K residue rows, items); mode 0 add, 1 sub
__global__ __launch_bounds__(kCoefThreads) void eltwise_items_kernel(const DevCtx* __restrict__ ctx, const u64* const* __restrict__ ta,
                                                                     const u64* const* __restrict__ tb, u64* const* __restrict__ tout, int mode) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = 2u * (blockIdx.x * kCoefThreads + threadIdx.x);
  const u32 row = blockIdx.y, item = blockIdx.z;
  if (k >= n) return;
  const u64 q = ctx->mod[row % K].q;
  const size_t off = (size_t)row * n + k;
  const u64x2_t x = *as_global(reinterpret_cast<const u64x2_t*>(ta[item] + off)), y = *as_global(reinterpret_cast<const u64x2_t*>(tb[item] + off));
  u64x2_t r;
  r.x = mode == 0 ? add_mod(x.x, y.x, q) : sub_mod(x.x, y.x, q);
  r.y = mode == 0 ? add_mod(x.y, y.y, q) : sub_mod(x.y, y.y, q);
  *as_global(reinterpret_cast<u64x2_t*>(tout[item] + off)) = r;
}
// ---- graph executor (program.cpp): signed n-ary sums through descriptor tables ----
// Every maximal Add / Sub / Negate tree of a program is one output here: out = sum_t sign_t * term_t (mod q_i), exact canonical
// arithmetic, so the bits are those of the reference's node-by-node evaluation whatever the association order
// (sunscreen_runtime/src/run.rs:217-236,283-311).  Terms may be shorter than the output (run.rs accepts add / sub of different
// ciphertext sizes): missing polynomials count as zero.  grid (N/2/256, max_size * K rows, outputs * batch).
__global__ __launch_bounds__(kCoefThreads) void nary_sum_kernel(const DevCtx* __restrict__ ctx, const NaryOut* __restrict__ outs,
                                                                const NaryTerm* __restrict__ terms, u32 batch) {
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const u32 n = ctx->n, K = ctx->K;
  const u32 k = 2u * (blockIdx.x * kCoefThreads + threadIdx.x);
  const u32 row = blockIdx.y;
  const NaryOut o = outs[blockIdx.z / batch];
  const u32 b = blockIdx.z % batch;
  if (k >= n || row >= o.size * K) return;
  const u32 poly = row / K;
  const u64 q = ctx->mod[row % K].q;
  u64x2_t acc;
  acc.x = 0, acc.y = 0;
  for (u32 t = 0; t < o.count; t++) {
    const NaryTerm tm = terms[o.first + t];
    if (poly >= tm.size) continue;
    const u64x2_t x = *as_global(reinterpret_cast<const u64x2_t*>(tm.ptr + ((size_t)b * tm.size * K + row) * n + k));
    acc.x = tm.sign > 0 ? add_mod(acc.x, x.x, q) : sub_mod(acc.x, x.x, q);
    acc.y = tm.sign > 0 ? add_mod(acc.y, x.y, q) : sub_mod(acc.y, x.y, q);
  }
  *as_global(reinterpret_cast<u64x2_t*>(o.out + ((size_t)b * o.size * K + row) * n + k)) = acc;
}
// ... and the transparent verdict of results that live in the callers' own buffers
__global__ __launch_bounds__(kCoefThreads) void transparent_flags_items_kernel(const u64* const* __restrict__ table, size_t words_per_ct, size_t skip_words,
                                                                              volatile u32* __restrict__ host_flags) {
  const auto p = as_global(table[blockIdx.x]);
  for (size_t base = skip_words; base < words_per_ct; base += kCoefThreads) {
    const size_t i = base + threadIdx.x;
    const bool nz = i < words_per_ct && p[i] != 0;
    if (__syncthreads_or(nz)) {
      if (threadIdx.x == 0) host_flags[blockIdx.x] = 1u;
      return;
    }
  }
  if (threadIdx.x == 0) host_flags[blockIdx.x] = 0u;
}
// transparent_flag_kernel for a batch: host_flags[item] = 1 if item's polynomials 1.. hold a non-zero word, else 0
__global__ __launch_bounds__(kCoefThreads) void transparent_flags_kernel(const u64* __restrict__ ct, size_t words_per_ct, size_t skip_words,
                                                                        volatile u32* __restrict__ host_flags) {
  const u64* p = ct + (size_t)blockIdx.x * words_per_ct;
  for (size_t base = skip_words; base < words_per_ct; base += kCoefThreads) {
    const size_t i = base + threadIdx.x;
    const bool nz = i < words_per_ct && p[i] != 0;
    if (__syncthreads_or(nz)) {
      if (threadIdx.x == 0) host_flags[blockIdx.x] = 1u;
      return;
    }
  }
  if (threadIdx.x == 0) host_flags[blockIdx.x] = 0u;
}

// =====================================================================================
// host launchers
// =====================================================================================

static inline dim3 coef_grid(u32 n, u32 y, u32 z = 1) { return dim3((n + kCoefThreads - 1) / kCoefThreads, y, z); }

template <int KMAX>
static void launch_extend_t(const DevCtx* ctx, u32 n, const u64* in0, u32 sa, const u64* in1, u32 sb, size_t ops, u64* out, hipStream_t s) {
  behz_extend_kernel<KMAX><<<coef_grid(n, (u32)(ops * (sa + sb))), kCoefThreads, 0, s>>>(ctx, in0, sa, in1, sb, out);
}

// K here: max(data primes, auxiliary primes - 2), selects the instantiation
hipError_t launch_behz_extend(const DevCtx* ctx, u32 n, u32 K, const u64* in0, u32 sa, const u64* in1, u32 sb, size_t ops, u64* out, hipStream_t s) {
  if (K <= 4)
    launch_extend_t<4>(ctx, n, in0, sa, in1, sb, ops, out, s);
  else if (K <= 8)
    launch_extend_t<8>(ctx, n, in0, sa, in1, sb, ops, out, s);
  else
    launch_extend_t<16>(ctx, n, in0, sa, in1, sb, ops, out, s);
  return hipGetLastError();
}

hipError_t launch_tensor(const DevCtx* ctx, u32 n, u32 R, const u64* ext, u32 sa, u32 sb, u64* D, size_t ops, hipStream_t s) {
  tensor_kernel<<<coef_grid(n, R, (u32)ops), kCoefThreads, 0, s>>>(ctx, ext, sa, sb, D);
  return hipGetLastError();
}

hipError_t launch_behz_floor_sk(const DevCtx* ctx, u32 n, u32 K, const u64* D, u64* out, size_t polys, hipStream_t s) {
  if (K <= 4)
    behz_floor_sk_kernel<4><<<coef_grid(n, (u32)polys), kCoefThreads, 0, s>>>(ctx, D, out);
  else if (K <= 8)
    behz_floor_sk_kernel<8><<<coef_grid(n, (u32)polys), kCoefThreads, 0, s>>>(ctx, D, out);
  else
    behz_floor_sk_kernel<16><<<coef_grid(n, (u32)polys), kCoefThreads, 0, s>>>(ctx, D, out);
  return hipGetLastError();
}

hipError_t launch_ks_decompose(const DevCtx* ctx, u32 n, u32 K, const u64* target, size_t tstride, u64* T, size_t ops, hipStream_t s) {
  ks_decompose_kernel<<<coef_grid(n, K, (u32)ops), kCoefThreads, 0, s>>>(ctx, target, tstride, T);
  return hipGetLastError();
}

hipError_t launch_ks_mac(const DevCtx* ctx, u32 n, u32 KK, const u64* T, const u64* key, u64* ACC, size_t ops, hipStream_t s, KeyMap km) {
  ks_mac_kernel<<<coef_grid(n, KK, (u32)ops), kCoefThreads, 0, s>>>(ctx, T, key, ACC, km);
  return hipGetLastError();
}

hipError_t launch_ks_moddown(const DevCtx* ctx, u32 n, const u64* ACC, const u64* base, size_t bstride, u32 base_mask, const u64* extra, u64* out,
                             size_t ops, hipStream_t s) {
  ks_moddown_kernel<<<coef_grid(n, 2, (u32)ops), kCoefThreads, 0, s>>>(ctx, ACC, base, bstride, base_mask, extra, out);
  return hipGetLastError();
}

hipError_t launch_mod_switch(const DevCtx* ctx, u32 n, const u64* in, u64* out, size_t polys, hipStream_t s) {
  mod_switch_kernel<<<coef_grid(n, (u32)polys), kCoefThreads, 0, s>>>(ctx, in, out);
  return hipGetLastError();
}

hipError_t launch_galois(const DevCtx* ctx, u32 n, u32 K, const u64* in, u64* out, size_t polys, u32 ginv, hipStream_t s) {
  // polys * K residue polynomials, one workgroup each (the host reads K from its own copy of the context: see the caller)
  switch (n) {
    case 4096: galois_lds_kernel<12><<<dim3((unsigned)(polys * K)), kGaloisThreads, 0, s>>>(ctx, in, out, ginv); return hipGetLastError();
    case 8192: galois_lds_kernel<13><<<dim3((unsigned)(polys * K)), kGaloisThreads, 0, s>>>(ctx, in, out, ginv); return hipGetLastError();
    case 16384: galois_lds_kernel<14><<<dim3((unsigned)(polys * K)), kGaloisThreads, 0, s>>>(ctx, in, out, ginv); return hipGetLastError();
    default: break;
  }
  galois_kernel<<<coef_grid(n, (u32)polys), kCoefThreads, 0, s>>>(ctx, in, out, ginv);
  return hipGetLastError();
}

hipError_t launch_eltwise(const DevCtx* ctx, u32 n, const u64* a, const u64* b, u64* out, size_t residue_polys, int mode, hipStream_t s) {
  // caller guarantees residue_polys <= 65535 and that `a` starts at a residue index that is a multiple of K
  eltwise_kernel<<<coef_grid(n / 2, (u32)residue_polys), kCoefThreads, 0, s>>>(ctx, a, b, out, mode);
  return hipGetLastError();
}

hipError_t launch_plain_addsub(const DevCtx* ctx, u32 n, u64* ct, size_t ctstride, const u64* plain, size_t pstride, size_t ops, int sub, hipStream_t s) {
  plain_addsub_kernel<<<coef_grid(n, (u32)ops), kCoefThreads, 0, s>>>(ctx, ct, ctstride, plain, pstride, sub);
  return hipGetLastError();
}

// nonzero: device u32[ops] scratch (zeroed here, filled with the per-plaintext non-zero counts) or nullptr = always centred lift
hipError_t launch_plain_lift(const DevCtx* ctx, u32 n, const u64* plain, size_t pstride, u64* out, size_t ops, u32* nonzero, hipStream_t s) {
  if (nonzero) {
    hipError_t e = hipMemsetAsync(nonzero, 0, ops * sizeof(u32), s);
    if (e != hipSuccess) return e;
    plain_count_kernel<<<coef_grid(n, (u32)ops), kCoefThreads, 0, s>>>(ctx, plain, pstride, nonzero);
  }
  plain_lift_kernel<<<coef_grid(n, (u32)ops), kCoefThreads, 0, s>>>(ctx, plain, pstride, out, nonzero);
  return hipGetLastError();
}

hipError_t launch_dyadic_plain(const DevCtx* ctx, u32 n, u32 K, u64* x, u32 size, const u64* pl, size_t plstride, size_t ops, hipStream_t s) {
  dyadic_plain_kernel<<<coef_grid(n, K, (u32)ops), kCoefThreads, 0, s>>>(ctx, x, size, pl, plstride);
  return hipGetLastError();
}

hipError_t launch_mono_mul(const DevCtx* ctx, u32 n, const u64* in, u64* out, size_t residue_polys, const u64* coeff_rns, u32 e, hipStream_t s) {
  mono_mul_kernel<<<coef_grid(n, (u32)residue_polys), kCoefThreads, 0, s>>>(ctx, in, out, coeff_rns, e);
  return hipGetLastError();
}

hipError_t launch_nonzero_tail(const u64* ct, size_t words_per_ct, size_t skip_words, u32* flags, size_t ops, hipStream_t s) {
  nonzero_tail_kernel<<<dim3(64, (u32)ops), kCoefThreads, 0, s>>>(ct, words_per_ct, skip_words, flags);
  return hipGetLastError();
}

hipError_t launch_transparent_flag(const u64* ct, size_t words, size_t skip_words, u32* host_flag, hipStream_t s) {
  transparent_flag_kernel<<<dim3(1), kCoefThreads, 0, s>>>(ct, words, skip_words, host_flag);
  return hipGetLastError();
}

hipError_t launch_gather_items(const u64* const* table, u64* stage, size_t words, size_t items, hipStream_t s) {
  gather_items_kernel<<<dim3(32, (u32)items), kCoefThreads, 0, s>>>(table, stage, words);
  return hipGetLastError();
}
hipError_t launch_copy_words(const u64* src, u64* dst, size_t words, hipStream_t s) {
  const u32 blocks = (u32)std::min<size_t>(256, (words + kCoefThreads - 1) / kCoefThreads);
  copy_words_kernel<<<dim3(blocks ? blocks : 1), kCoefThreads, 0, s>>>(src, dst, words);
  return hipGetLastError();
}
hipError_t launch_scatter_items(const u64* stage, u64* const* table, size_t words, size_t items, hipStream_t s) {
  scatter_items_kernel<<<dim3(32, (u32)items), kCoefThreads, 0, s>>>(stage, table, words);
  return hipGetLastError();
}
hipError_t launch_eltwise_items(const DevCtx* ctx, u32 n, u32 K, const u64* const* ta, const u64* const* tb, u64* const* tout, int mode, size_t items,
                                hipStream_t s) {
  eltwise_items_kernel<<<coef_grid(n / 2, 2 * K, (u32)items), kCoefThreads, 0, s>>>(ctx, ta, tb, tout, mode);
  return hipGetLastError();
}
hipError_t launch_nary_sum(const DevCtx* ctx, u32 n, u32 K, const NaryOut* outs, const NaryTerm* terms, u32 nouts, u32 max_size, u32 batch, hipStream_t s) {
  // grid z <= 65535: the outputs go in slices (the tables are indexed from the slice's first output)
  if (batch == 0 || batch > 65535u) return hipErrorInvalidValue;  // one output's items must fit grid z (callers chunk the batch)
  const u32 per = std::max(1u, 65535u / batch);
  for (u32 off = 0; off < nouts; off += per) {
    const u32 c = std::min(per, nouts - off);
    nary_sum_kernel<<<coef_grid(n / 2, max_size * K, c * batch), kCoefThreads, 0, s>>>(ctx, outs + off, terms, batch);
  }
  return hipGetLastError();
}
hipError_t launch_transparent_flags_items(const u64* const* table, size_t words_per_ct, size_t skip_words, u32* host_flags, size_t items, hipStream_t s) {
  transparent_flags_items_kernel<<<dim3((u32)items), kCoefThreads, 0, s>>>(table, words_per_ct, skip_words, host_flags);
  return hipGetLastError();
}
hipError_t launch_transparent_flags(const u64* ct, size_t words_per_ct, size_t skip_words, u32* host_flags, size_t items, hipStream_t s) {
  transparent_flags_kernel<<<dim3((u32)items), kCoefThreads, 0, s>>>(ct, words_per_ct, skip_words, host_flags);
  return hipGetLastError();
}

hipError_t launch_transparent_watch_nary(const DevCtx* ctx, const NaryOut* outs, u32 nouts, u32 batch, u32* status, hipStream_t s) {
  const u32 per = std::max(1u, 0x7FFFFFFFu / batch);
  for (u32 off = 0; off < nouts; off += per) {
    const u32 c = std::min(per, nouts - off);
    transparent_watch_nary_kernel<<<dim3(c * batch), kCoefThreads, 0, s>>>(ctx, outs + off, batch, status);
  }
  return hipGetLastError();
}
hipError_t launch_zero_plain_watch(const u32* nonzero, u32 first_item, u32 item_step, u32 count, u32* status, hipStream_t s) {
  zero_plain_watch_kernel<<<dim3((count + 255) / 256), 256, 0, s>>>(nonzero, first_item, item_step, count, status);
  return hipGetLastError();
}

hipError_t launch_transparent_watch(const u64* ct, size_t words_per_ct, size_t skip_words, u32 first_item, u32* status, size_t ops, hipStream_t s) {
  transparent_watch_kernel<<<dim3((u32)ops), kCoefThreads, 0, s>>>(ct, words_per_ct, skip_words, first_item, status);
  return hipGetLastError();
}

}  // namespace hipbfv
