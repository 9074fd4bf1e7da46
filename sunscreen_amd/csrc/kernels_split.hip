// sunscreen_amd/csrc/kernels_split.hip -- "head / middle / tail" split-transform kernels.
//
// A negacyclic NTT of N = 2^L points is L radix-2 stages.  The first head_log(L) stages (gaps >= N/8) and the
// last tail_log(L) stages of the inverse (gaps >= N/4; N/8 at L = 14, nttshape.hpp) are the only ones that couple distant coefficients;
// all stages in between act inside contiguous blocks of N/4 coefficients.  So instead of one LDS-resident
// whole-polynomial transform per residue (kernels.hip), the pipeline is cut at those two places:
//
//   head   : coefficient-parallel producer (RNS decomposition / base extension) that ALSO performs the first
//            three forward stages on the 8 coefficients {t + k*N/8} it owns (wave-uniform twiddles);
//   middle : one workgroup per (op, prime, block of N/4 coefficients): remaining forward stages, the pointwise
//            work (key multiply-accumulate / tensor product), first L-2 inverse stages -- 16 KB of LDS per
//            polynomial block instead of 64 KB, no whole-polynomial round trips through HBM;
//   tail   : coefficient-parallel consumer that performs the last two inverse stages on the 4 coefficients
//            {t + k*N/4} it owns, scales, and runs the per-coefficient epilogue (mod-down / BEHZ floor).
//
// Intermediates between the three kernels are stored in the arithmetic policy's native lazy representation
// (IEEE doubles holding exact integers for the FP64 path) -- they never leave this file.
//
// The middle kernels advance all polynomials of their (prime, block) together, pass by pass (mid_forward_multi):
// one barrier and one set of twiddle fetches per pass, independent butterfly streams, all loads in flight at once.
//
// This file: key switching (SEAL Evaluator::switch_key_inplace; Evaluator_Relinearize / RotateRows /
// RotateColumns, seal_fhe/src/bfv_evaluator.rs:148-244) for contexts whose key-level primes all take the
// FP64 path; the BEHZ multiply (Evaluator_Multiply) for up to 8 data primes; the two-kernel transforms of N = 32768.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "behzcore.hpp"
#include "moddown_d.hpp"
#include "kernels.hpp"
#include "nttcore.hpp"

namespace hipbfv {

// Inside a block guarded by a WAVE-UNIFORM condition that is false for the common parameter sets: keeps the block a branch.  Left
// alone, the compiler if-converts small blocks -- computes the reduction for every value and selects (3 FP64 + 2 v_cndmask per value,
// seen in the ISA of the mod-down), which is exactly the work the condition was there to skip.
#define HIPBFV_KEEP_BRANCH() asm volatile("")

// LDS placement inside a block: the map of the whole-polynomial kernels (nttcore.hpp lds_pos; one search covers both families)
__device__ __forceinline__ u32 blk_pos(u32 e) { return lds_pos(e); }

// EPT: elements per middle-kernel thread (8).  At N = 16384 a block is 4096 coefficients, and 512-thread workgroups with four
// 32 KB exchange regions leave room for ONE workgroup per CU -- nothing overlaps its load and store phases (an N = 8192 middle
// kernel padded down to one workgroup per CU ran 1.7x slower in an r02 experiment).  The machinery is parametrised on EPT so that 16
// elements per thread (256-thread workgroups, two of them resident) could be tried there: see MID_EPT_14.
template <int L, int EPT = kBlkEPT>
struct SplitShape {
  static constexpr int N = 1 << L;
  static constexpr int LB = L - tail_log(L);
  static constexpr int BLOCK = 1 << LB;           // coefficients per middle workgroup
  static constexpr int TPB = BLOCK / EPT;         // threads per middle workgroup
  static constexpr int NBLK = 1 << tail_log(L);   // blocks per polynomial
  static constexpr int NPF = split_fwd_passes(L);
  static constexpr int NPI = split_inv_passes(L);
  // the pointwise work (tensor product / key multiply-accumulate) happens on the register layout the last forward pass
  // leaves, and the first inverse pass starts from it: both must use the window [0, R) with the same R
  static_assert(split_fwd_radix(L, NPF - 1) == split_inv_radix(L, 0), "last forward and first inverse pass must share their window");
  static_assert(split_fwd_low(L, NPF - 1) == 0 && split_inv_low(L, 0) == 0, "pointwise layout is the window at bit 0");
};

// One pass of a middle kernel over the index window [LOW, LOW+R).  `blk` is the block index inside the
// polynomial; virtual threads are numbered globally so that twiddle indices are those of the full transform.
template <class A, int L, int LOW, int R, int EPT = kBlkEPT>
struct BlkPass {
  using Sh = SplitShape<L, EPT>;
  static constexpr int G = EPT >> R;
  static __device__ __forceinline__ u32 vt(u32 tid, u32 blk, int g) { return blk * (u32)(Sh::BLOCK >> R) + tid + (u32)g * Sh::TPB; }
  static __device__ __forceinline__ u32 elem(u32 tid, u32 blk, int g, int k) { return elem_index<LOW, R>(vt(tid, blk, g), (u32)k); }
  // position inside the block's LDS image: per-thread part ^ / + compile-time parts (nttcore.hpp pass_pos: the DS
  // instructions take the constant part as their immediate offset)
  static __device__ __forceinline__ u32 pos(u32 tid, u32 blk, int g, int k) {
    static_assert((Sh::TPB & (Sh::TPB - 1)) == 0, "tid and g*TPB must occupy disjoint bits");
    const u32 P = blk_pos(elem_index<LOW, R>(tid, 0));
    const u32 X = blk_pos(elem_index<LOW, R>((u32)g * Sh::TPB, (u32)k));
    return (P ^ (X & 31u)) + (X & ~31u);
  }
  static __device__ __forceinline__ void load_lds(typename A::V (&v)[EPT], const typename A::V* smem, u32 tid, u32 blk) {
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) v[g * (1 << R) + k] = smem[pos(tid, blk, g, k)];
  }
  static __device__ __forceinline__ void store_lds(const typename A::V (&v)[EPT], typename A::V* smem, u32 tid, u32 blk) {
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int k = 0; k < (1 << R); k++) smem[pos(tid, blk, g, k)] = v[g * (1 << R) + k];
  }
  // forward stages S0..S0+R-1, S0 = L - LOW - R
  static __device__ __forceinline__ void fwd(const A& ar, typename A::V (&v)[EPT], u32 tid, u32 blk, const typename A::Tw* __restrict__ tw) {
    constexpr int S0 = L - LOW - R;
#pragma unroll
    for (int g = 0; g < G; g++) {
      u32 hi = vt(tid, blk, g) >> LOW;
      if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int half = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
          if (k & half) continue;
          const typename A::Tw w = tw[(1u << (S0 + j)) + ((hi << j) | (u32)(k >> (R - j)))];
          ar.fwd(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w);
        }
      }
    }
  }
  // The same stages with the twiddles fetched separately (load_tw_*) so that the fetch can be issued a pass ahead
  // of its use: w[g*NW + slot], forward slot of (stage j, c) = 2^j - 1 + c, inverse slot = 2^R - 2^(R-j) + c.
  static constexpr int NW = (1 << R) - 1;
  static __device__ __forceinline__ void load_tw_fwd(typename A::Tw (&w)[EPT - 1], u32 tid, u32 blk, const typename A::Tw* __restrict__ tw) {
    constexpr int S0 = L - LOW - R;
#pragma unroll
    for (int g = 0; g < G; g++) {
      u32 hi = vt(tid, blk, g) >> LOW;
      if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);
#pragma unroll
      for (int j = 0; j < R; j++)
#pragma unroll
        for (int c = 0; c < (1 << j); c++) w[g * NW + (1 << j) - 1 + c] = tw[(1u << (S0 + j)) + ((hi << j) | (u32)c)];
    }
  }
  static __device__ __forceinline__ void fwd_tw(const A& ar, typename A::V (&v)[EPT], const typename A::Tw (&w)[EPT - 1]) {
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int half = 1 << (R - 1 - j);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
          if (k & half) continue;
          ar.fwd(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w[g * NW + (1 << j) - 1 + (k >> (R - j))]);
        }
      }
  }
  static __device__ __forceinline__ void load_tw_inv(typename A::Tw (&w)[EPT - 1], u32 tid, u32 blk, const typename A::Tw* __restrict__ tw) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      u32 hi = vt(tid, blk, g) >> LOW;
      if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);
#pragma unroll
      for (int j = 0; j < R; j++)
#pragma unroll
        for (int c = 0; c < (1 << (R - 1 - j)); c++)
          w[g * NW + (1 << R) - (1 << (R - j)) + c] = tw[(1u << (L - 1 - LOW - j)) + ((hi << (R - 1 - j)) | (u32)c)];
    }
  }
  static __device__ __forceinline__ void inv_tw(const A& ar, typename A::V (&v)[EPT], const typename A::Tw (&w)[EPT - 1]) {
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int half = 1 << j;
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
          if (k & half) continue;
          ar.inv(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w[g * NW + (1 << R) - (1 << (R - j)) + (k >> (j + 1))]);
        }
      }
  }
  // inverse stages with global gaps 2^LOW .. 2^(LOW+R-1)
  static __device__ __forceinline__ void inv(const A& ar, typename A::V (&v)[EPT], u32 tid, u32 blk, const typename A::Tw* __restrict__ tw) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      u32 hi = vt(tid, blk, g) >> LOW;
      if constexpr (LOW >= 6) hi = __builtin_amdgcn_readfirstlane(hi);
#pragma unroll
      for (int j = 0; j < R; j++) {
        const int half = 1 << j;
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
          if (k & half) continue;
          const typename A::Tw w = tw[(1u << (L - 1 - LOW - j)) + ((hi << (R - 1 - j)) | (u32)(k >> (j + 1)))];
          ar.inv(v[g * (1 << R) + k], v[g * (1 << R) + k + half], w);
        }
      }
    }
  }
};

// Which exchanges of a middle kernel need a workgroup barrier: the same argument as in kernels.hip (exchange_is_wave_private).
// Virtual thread vt = blk*(BLOCK >> R) + tid + g*TPB; the wave-id bits of tid (6 .. log2 TPB - 1) land at element bit b + R when
// b >= LOW, else at b.  Two consecutive passes that agree on this for every wave-id bit exchange data inside a wavefront only.
// At N = 8192 that holds for the last forward exchange (windows 2..3 -> 0..1) and for the inverse exchange 2..4 -> 5..7.
#define MID_WAVE_PRIVATE 1
// Experiment hook (tools/build_variant.sh): wave priority around the LDS exchanges of the pass-batched middle kernels.
// 1 = raised from the exchange's stores until its loads are issued (a wave that reaches the exchange gets through it first),
// 2 = raised while the butterflies run, normal inside the exchange.  0 = no s_setprio at all (the default build).
#ifndef MID_SETPRIO
#define MID_SETPRIO 0
#endif
__device__ __forceinline__ void mid_prio_exchange_begin() {
  if constexpr (MID_SETPRIO == 1) __builtin_amdgcn_s_setprio(3);
  if constexpr (MID_SETPRIO == 2) __builtin_amdgcn_s_setprio(0);
}
__device__ __forceinline__ void mid_prio_exchange_end() {
  if constexpr (MID_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
  if constexpr (MID_SETPRIO == 2) __builtin_amdgcn_s_setprio(3);
}
template <int L, int EPT = kBlkEPT>
constexpr bool blk_exchange_private(int lowa, int ra, int lowb, int rb) {
  if (!MID_WAVE_PRIVATE) return false;
  constexpr int logt = SplitShape<L, EPT>::LB - ilog2(EPT);
  for (int b = 6; b < logt; b++)
    if ((b >= lowa ? b + ra : b) != (b >= lowb ? b + rb : b)) return false;
  return true;
}
template <bool PRIVATE>
__device__ __forceinline__ void blk_exchange_sync() {
  if constexpr (PRIVATE) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
}

template <class A, int EPT>
__device__ __forceinline__ void reduce_all(const A& ar, typename A::V (&v)[EPT]) {
#pragma unroll
  for (int e = 0; e < EPT; e++) v[e] = ar.reduce(v[e]);
}

// TW_PIPE_D / TW_PIPE_I (FP64 / integer policy): 0 = every pass fetches its twiddles when it needs them; 1 = the next pass's twiddles are fetched
// before the LDS exchange that precedes it; 2 = before the current pass's butterflies (needs both sets live).
#define TW_PIPE_D 1
#define TW_PIPE_I 1
template <class A>
struct TwPipe {
  static constexpr int value = TW_PIPE_I;
};
template <>
struct TwPipe<ArithD> {
  static constexpr int value = TW_PIPE_D;
};

template <class A, int L, int P>
__device__ __forceinline__ void mid_forward_p(const A& ar, typename A::V (&v)[kBlkEPT], typename A::V* smem, u32 tid, u32 blk,
                                              const typename A::Tw* tw, u32 mask, const typename A::Tw (&w)[kBlkEPT - 1]) {
  constexpr int R = split_fwd_radix(L, P);
  constexpr int LOW = split_fwd_low(L, P);
  using Pass = BlkPass<A, L, LOW, R>;
  constexpr bool more = P + 1 < SplitShape<L>::NPF;
  constexpr int PN = more ? P + 1 : P;
  using Next = BlkPass<A, L, split_fwd_low(L, PN), split_fwd_radix(L, PN)>;
  typename A::Tw wn[kBlkEPT - 1];
  if constexpr (P > 0) {
    blk_exchange_sync<blk_exchange_private<L>(split_fwd_low(L, P > 0 ? P - 1 : 0), split_fwd_radix(L, P > 0 ? P - 1 : 0), LOW, R)>();
    Pass::load_lds(v, smem, tid, blk);
  }
  if constexpr (more && TwPipe<A>::value == 2) Next::load_tw_fwd(wn, tid, blk, tw);
  if ((mask >> P) & 1u) reduce_all(ar, v);
  if ((mask >> (P + 16)) & 1u) reduce_all(ar, v);  // very wide primes: a second reduction (context.cpp range plan)
  Pass::fwd_tw(ar, v, w);
  if constexpr (more) {
    if constexpr (TwPipe<A>::value != 2) Next::load_tw_fwd(wn, tid, blk, tw);
    Pass::store_lds(v, smem, tid, blk);
    mid_forward_p<A, L, PN>(ar, v, smem, tid, blk, tw, mask, wn);
  }
}

template <class A, int L, int P>
__device__ __forceinline__ void mid_inverse_p(const A& ar, typename A::V (&v)[kBlkEPT], typename A::V* smem, u32 tid, u32 blk,
                                              const typename A::Tw* tw, u32 mask, const typename A::Tw (&w)[kBlkEPT - 1]) {
  constexpr int R = split_inv_radix(L, P);
  constexpr int LOW = split_inv_low(L, P);
  using Pass = BlkPass<A, L, LOW, R>;
  constexpr bool more = P + 1 < SplitShape<L>::NPI;
  constexpr int PN = more ? P + 1 : P;
  using Next = BlkPass<A, L, split_inv_low(L, PN), split_inv_radix(L, PN)>;
  typename A::Tw wn[kBlkEPT - 1];
  if constexpr (P > 0) {
    blk_exchange_sync<blk_exchange_private<L>(split_inv_low(L, P > 0 ? P - 1 : 0), split_inv_radix(L, P > 0 ? P - 1 : 0), LOW, R)>();
    Pass::load_lds(v, smem, tid, blk);
  }
  if constexpr (more && TwPipe<A>::value == 2) Next::load_tw_inv(wn, tid, blk, tw);
  if ((mask >> P) & 1u) reduce_all(ar, v);
  if ((mask >> (P + 16)) & 1u) reduce_all(ar, v);
  Pass::inv_tw(ar, v, w);
  if constexpr (more) {
    if constexpr (TwPipe<A>::value != 2) Next::load_tw_inv(wn, tid, blk, tw);
    Pass::store_lds(v, smem, tid, blk);
    mid_inverse_p<A, L, PN>(ar, v, smem, tid, blk, tw, mask, wn);
  }
}

// Remaining forward stages of one block.  On entry v holds the block's values in the layout of forward
// middle pass 0 (window split_fwd_low(L,0)); on exit v holds the last pass's results (window LOW = 0).
template <class A, int L, int P>
__device__ __forceinline__ void mid_forward(const A& ar, typename A::V (&v)[kBlkEPT], typename A::V* smem, u32 tid, u32 blk,
                                            const typename A::Tw* tw, u32 mask) {
  constexpr int R = split_fwd_radix(L, P);
  constexpr int LOW = split_fwd_low(L, P);
  using Pass = BlkPass<A, L, LOW, R>;
  if constexpr (TwPipe<A>::value != 0) {
    static_assert(P == 0, "pipelined transforms start at pass 0");
    typename A::Tw w[kBlkEPT - 1];
    Pass::load_tw_fwd(w, tid, blk, tw);
    mid_forward_p<A, L, 0>(ar, v, smem, tid, blk, tw, mask, w);
  } else {
    if constexpr (P > 0) {
      __syncthreads();
      Pass::load_lds(v, smem, tid, blk);
    }
    if ((mask >> P) & 1u) reduce_all(ar, v);
    if ((mask >> (P + 16)) & 1u) reduce_all(ar, v);
    Pass::fwd(ar, v, tid, blk, tw);
    if constexpr (P + 1 < SplitShape<L>::NPF) {
      Pass::store_lds(v, smem, tid, blk);
      constexpr int PN = TwPipe<A>::value != 0 ? 0 : P + 1;
      mid_forward<A, L, PN>(ar, v, smem, tid, blk, tw, mask);
    }
  }
}

// First L-2 inverse stages of one block.  On entry v holds values in the layout mid_forward leaves
// (window LOW = 0); on exit v holds the results of the last middle pass (window split_inv_low(L, NPI-1)).
template <class A, int L, int P>
__device__ __forceinline__ void mid_inverse(const A& ar, typename A::V (&v)[kBlkEPT], typename A::V* smem, u32 tid, u32 blk,
                                            const typename A::Tw* tw, u32 mask) {
  constexpr int R = split_inv_radix(L, P);
  constexpr int LOW = split_inv_low(L, P);
  using Pass = BlkPass<A, L, LOW, R>;
  if constexpr (TwPipe<A>::value != 0) {
    static_assert(P == 0, "pipelined transforms start at pass 0");
    typename A::Tw w[kBlkEPT - 1];
    Pass::load_tw_inv(w, tid, blk, tw);
    mid_inverse_p<A, L, 0>(ar, v, smem, tid, blk, tw, mask, w);
  } else {
    if constexpr (P > 0) {
      __syncthreads();
      Pass::load_lds(v, smem, tid, blk);
    }
    if ((mask >> P) & 1u) reduce_all(ar, v);
    if ((mask >> (P + 16)) & 1u) reduce_all(ar, v);
    Pass::inv(ar, v, tid, blk, tw);
    if constexpr (P + 1 < SplitShape<L>::NPI) {
      Pass::store_lds(v, smem, tid, blk);
      constexpr int PN = TwPipe<A>::value != 0 ? 0 : P + 1;
      mid_inverse<A, L, PN>(ar, v, smem, tid, blk, tw, mask);
    }
  }
}

// The same transforms for NP polynomials of one (residue, block) at once, pass by pass: one barrier and one set
// of twiddle fetches per pass serves all NP polynomials, their butterflies are independent instruction streams
// the scheduler can interleave, and the NP global loads / stores are in flight together.  smem: NP regions of
// BLOCK elements.
// PIPE: the next pass's twiddles are fetched before the LDS exchange that precedes it (two sets live: 2 * (EPT - 1) twiddles);
// false: every pass fetches its own after the exchange (one set live) -- the register diet of the two-workgroups-per-CU
// instantiations at N = 16384.
template <class A, int L, int P, int NP, int EPT = kBlkEPT, bool PIPE = true>
__device__ __forceinline__ void mid_forward_multi_p(const A& ar, typename A::V (&v)[NP][EPT], typename A::V* smem, u32 tid, u32 blk,
                                                    const typename A::Tw* tw, u32 mask, const typename A::Tw (&w)[EPT - 1]) {
  using Sh = SplitShape<L, EPT>;
  constexpr int R = split_fwd_radix(L, P);
  constexpr int LOW = split_fwd_low(L, P);
  using Pass = BlkPass<A, L, LOW, R, EPT>;
  constexpr bool more = P + 1 < Sh::NPF;
  constexpr int PN = more ? P + 1 : P;
  using Next = BlkPass<A, L, split_fwd_low(L, PN), split_fwd_radix(L, PN), EPT>;
  typename A::Tw wn[EPT - 1];
  if constexpr (P > 0) {
    blk_exchange_sync<blk_exchange_private<L, EPT>(split_fwd_low(L, P > 0 ? P - 1 : 0), split_fwd_radix(L, P > 0 ? P - 1 : 0), LOW, R)>();
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::load_lds(v[i], smem + i * Sh::BLOCK, tid, blk);
    mid_prio_exchange_end();
  }
#pragma unroll
  for (int i = 0; i < NP; i++) {
    if ((mask >> P) & 1u) reduce_all(ar, v[i]);
    if ((mask >> (P + 16)) & 1u) reduce_all(ar, v[i]);
  }
  if constexpr (PIPE || P == 0) {
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::fwd_tw(ar, v[i], w);
  } else {
    typename A::Tw wo[EPT - 1];
    Pass::load_tw_fwd(wo, tid, blk, tw);
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::fwd_tw(ar, v[i], wo);
  }
  if constexpr (more) {
    if constexpr (PIPE) Next::load_tw_fwd(wn, tid, blk, tw);
    mid_prio_exchange_begin();
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::store_lds(v[i], smem + i * Sh::BLOCK, tid, blk);
    mid_forward_multi_p<A, L, PN, NP, EPT, PIPE>(ar, v, smem, tid, blk, tw, mask, PIPE ? wn : w);
  }
}
template <class A, int L, int NP, int EPT = kBlkEPT, bool PIPE = true>
__device__ __forceinline__ void mid_forward_multi(const A& ar, typename A::V (&v)[NP][EPT], typename A::V* smem, u32 tid, u32 blk,
                                                  const typename A::Tw* tw, u32 mask) {
  using Pass = BlkPass<A, L, split_fwd_low(L, 0), split_fwd_radix(L, 0), EPT>;
  typename A::Tw w[EPT - 1];
  Pass::load_tw_fwd(w, tid, blk, tw);
  mid_forward_multi_p<A, L, 0, NP, EPT, PIPE>(ar, v, smem, tid, blk, tw, mask, w);
}

template <class A, int L, int P, int NP, int EPT = kBlkEPT, bool PIPE = true>
__device__ __forceinline__ void mid_inverse_multi_p(const A& ar, typename A::V (&v)[NP][EPT], typename A::V* smem, u32 tid, u32 blk,
                                                    const typename A::Tw* tw, u32 mask, const typename A::Tw (&w)[EPT - 1]) {
  using Sh = SplitShape<L, EPT>;
  constexpr int R = split_inv_radix(L, P);
  constexpr int LOW = split_inv_low(L, P);
  using Pass = BlkPass<A, L, LOW, R, EPT>;
  constexpr bool more = P + 1 < Sh::NPI;
  constexpr int PN = more ? P + 1 : P;
  using Next = BlkPass<A, L, split_inv_low(L, PN), split_inv_radix(L, PN), EPT>;
  typename A::Tw wn[EPT - 1];
  if constexpr (P > 0) {
    blk_exchange_sync<blk_exchange_private<L, EPT>(split_inv_low(L, P > 0 ? P - 1 : 0), split_inv_radix(L, P > 0 ? P - 1 : 0), LOW, R)>();
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::load_lds(v[i], smem + i * Sh::BLOCK, tid, blk);
    mid_prio_exchange_end();
  }
#pragma unroll
  for (int i = 0; i < NP; i++) {
    if ((mask >> P) & 1u) reduce_all(ar, v[i]);
    if ((mask >> (P + 16)) & 1u) reduce_all(ar, v[i]);
  }
  if constexpr (PIPE || P == 0) {
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::inv_tw(ar, v[i], w);
  } else {
    typename A::Tw wo[EPT - 1];
    Pass::load_tw_inv(wo, tid, blk, tw);
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::inv_tw(ar, v[i], wo);
  }
  if constexpr (more) {
    if constexpr (PIPE) Next::load_tw_inv(wn, tid, blk, tw);
    mid_prio_exchange_begin();
#pragma unroll
    for (int i = 0; i < NP; i++) Pass::store_lds(v[i], smem + i * Sh::BLOCK, tid, blk);
    mid_inverse_multi_p<A, L, PN, NP, EPT, PIPE>(ar, v, smem, tid, blk, tw, mask, PIPE ? wn : w);
  }
}
template <class A, int L, int NP, int EPT = kBlkEPT, bool PIPE = true>
__device__ __forceinline__ void mid_inverse_multi(const A& ar, typename A::V (&v)[NP][EPT], typename A::V* smem, u32 tid, u32 blk,
                                                  const typename A::Tw* tw, u32 mask) {
  using Pass = BlkPass<A, L, split_inv_low(L, 0), split_inv_radix(L, 0), EPT>;
  typename A::Tw w[EPT - 1];
  Pass::load_tw_inv(w, tid, blk, tw);
  mid_inverse_multi_p<A, L, 0, NP, EPT, PIPE>(ar, v, smem, tid, blk, tw, mask, w);
}

// Entry `idx` of a residue list (DevCtx::mid_res_* / ks_res_*: byte arrays at 4-byte-aligned offsets of the context) through the
// SCALAR unit: gfx950 has no scalar byte load, so `list[idx]` is a vector global_load_ubyte followed by s_waitcnt vmcnt(0) -- a
// dependent HBM / L2 round trip at the very start of every middle-kernel workgroup, before its first polynomial load can be
// issued (found in r04 in the load / wait listing of the ISA).  The aligned word that holds the byte is a wave-uniform s_load.
__device__ __forceinline__ u32 residue_of(const unsigned char* __restrict__ list, u32 idx) {
  idx = __builtin_amdgcn_readfirstlane(idx);
  const u32 word = reinterpret_cast<const u32*>(list)[idx >> 2];
  return (word >> ((idx & 3u) * 8u)) & 0xffu;
}

constexpr int kHeadThreads = 256;
// the mixed-base head (integer data primes, FP64 auxiliary primes: the <L, 4, AUXD = false, PACK = true> instantiations): waves per
// SIMD its registers are limited for.  Measured on the 3 x 54-bit workload: 3 (168 registers, 76 bytes of scratch) 2.66-2.81 ms,
// the compiler's choice (185 registers, 2 waves) 2.90-3.01, 4 (128 registers, 236 bytes) 3.03-3.07.
// coefficients per trip of the mixed-base tail's epilogue (1, 2 or 4 of the thread's four)
#define TAIL_MIXED_NC 4  // measured on the 3 x 54-bit workload: 1 -> 3.56, 2 -> 3.66-3.71, 4 (fully unrolled, 154 registers) -> 3.22 ms
#define HEAD_MIXED_WAVES 3
#define EDGE_BOUNDS(KMAX) __launch_bounds__(kHeadThreads)
// mul_mid, FP64 instantiation: at N = 8192 the four forward transforms go through the exchange buffer as two pairs
// (48 KB of LDS per workgroup instead of 64 KB: 3 workgroups = 12 waves per CU instead of 2 = 8; -5 % mul_mid).  At N = 4096
// the regions are small anyway and the extra barriers cost 37 %; at N = 16384 one workgroup fills the CU either way.
#define MID_FWD_PAIRS(L) ((L) == 13)
// MODE 3: the two single-operand forward transforms run while two transformed operands and one product row are live (96
// registers): whether they also prefetch the next pass's twiddles (a second set of twiddle registers)
#define MID3_PIPE_SINGLE true  // measured: 6.10-6.15 with, 6.19-6.34 ms without; the scratch is the same either way
// MODE 3, last forward round: the first MID3_PARK values of the waiting partial product a0 b1 sit in thread-private LDS slots
// behind the two exchange regions while a1 is transformed.  r03 left them to the register allocator, which spilt 22 dwords per
// lane to scratch -- written once, read once, and visible in the PMC passes as 1.6 MB of writes and 1.7 MB of reads per op over
// the kernel's own model (72 workgroups x 256 lanes x 88 bytes).  8 values: 240 registers (254 with 48-bit packed rows), no
// scratch, 80 KB of LDS per workgroup -- two are resident per CU (measured: 6 values / 76 KB and 8 / 80 KB run the same 5.76-5.79
// ms per 1024 ops against 6.10-6.13 with the scratch; 6 leave the packed-row instantiation 24 bytes of scratch, 4 leave both some).
#define MID3_PARK 8
// mul_mid (pass-batched bodies): next pass's twiddles fetched before the exchange (a second set of twiddle registers)
#define MID_TW_PIPE(L) true
#define MID_WAVES_D(L) ((L) == 13 ? 3 : 2)
#define MID_WAVES_I 3
// ks_mid: digits transformed together (4 or 2; LDS = that many exchange regions) and waves per SIMD the register
// allocation aims at.  Pairs + 3 waves: -5.5 % at N = 8192, -11 % at N = 4096; at N = 16384 (32 KB regions, K = 8) groups
// of four at 2 waves stay faster (+4 % the other way).
// (experiment hooks for N = 16384, tools/build_variant.sh: KS_GROUP_14 digits per group, KS_WAVES_14 waves per SIMD, KS_PIPE_14)
#ifndef KS_GROUP_14
#define KS_GROUP_14 (HIPBFV_GEOM14 == 4 ? 2 : 4)
#endif
#ifndef KS_WAVES_14
#define KS_WAVES_14 2
#endif
#ifndef KS_PIPE_14
#define KS_PIPE_14 true
#endif
#define KS_GROUP_MAX(L) ((L) <= 13 ? 2 : KS_GROUP_14)
#define KS_MID_WAVES(L) ((L) <= 13 ? 3 : KS_WAVES_14)
// r03, N = 16384: 16 elements per thread (256-thread workgroups) with the digits in PAIRS: 64 KB of LDS and 256 registers per
// workgroup, two workgroups per CU whose load / compute / store phases overlap -- ks_mid 5.53 -> 5.29 ms per 1024 ops
// (interleaved A/B; the first configuration with two resident workgroups that does not spill: two 512-thread workgroups would
// have to live in 128 registers each, and every such variant spilt and lost 4-11 %).  One digit at a time (32 KB) is slower (5.93).
// ks_mid (FP64 policy): elements per thread
#define KS_EPT(L) ((L) == 14 && HIPBFV_GEOM14 == 4 ? 16 : kBlkEPT)
// ks_mid: twiddles of the next pass fetched before the exchange (costs a second set of twiddle registers)
#define KS_TW_PIPE(L) ((L) == 14 ? KS_PIPE_14 : true)
// ks_mid: key words loaded per accumulation step (elements; 8 = a whole window at once)
#define KS_MAC_CHUNK(L) 8

// -------------------------------------------------------------------------------------------------
// 48-bit packed storage of FP64-policy intermediates (PACK): every kernel of the split pipelines is HBM-bound or
// close to it and its intermediates are integers below 2^47 in magnitude once reduced, so they travel as 6 bytes instead
// of 8.  A residue polynomial keeps its 8N-byte region; the first 4N bytes hold the low words (u32 plane), the next
// 2N bytes bits 32..47 (i16 plane) of the two's-complement value: both planes are read and written fully coalesced.
// Packing/unpacking rides on the 2^52-magic conversion: r + (2^52 + 2^51 + 2^47) has mantissa 2^51 + (r + 2^47) with r + 2^47 in
// [0, 2^48), so the high dword of the sum is 0x43380000 + (16 bits) and the i16 plane holds those 16 bits BIASED (r05: the plane used
// to hold the signed high part, and every unpack sign-extended it first): unpack = one OR onto a zero-extended load.
// -------------------------------------------------------------------------------------------------
constexpr double kPackMagic = 6755399441055744.0 + 140737488355328.0;  // 2^52 + 2^51 + 2^47

// Non-temporal hints per site, per degree (NAT_NT(L) bits): 1 = loads of mul_mid, 2 = loads of the tail kernels,
// 4 = stores of mul_head, 8 = stores of mul_mid, 16 = stores of ks_head, 32 = loads of ks_mid, 64 = stores of ks_mid.
// The intermediates are written once and read once, by the next kernel, in chunks far larger than the L2 / Infinity Cache:
// the middle kernels mark their loads and stores non-temporal (measured, interleaved A/B on one box: mul_mid -2.3...3.4 %,
// ks_mid -3.4...5.4 %; mul+relin +1.7...2.2 % at n = 8192, +2.9 % at n = 4096, +1.5 % at n = 16384).  The hints interact
// through the cache state the next kernel finds: non-temporal loads in the tail kernels speed ks_tail up and slow ks_head
// down by as much, non-temporal stores in ks_head cost it 10-20 %, and at N = 16384 (8-byte intermediates) they slow
// mul_mid's stores down by 5 %: those sites stay temporal.
#define NAT_NT(L) ((L) <= 13 ? (1 | 8 | 32 | 64) : (32 | 64))
template <int L>
struct NtSites {
  static constexpr int bits = NAT_NT(L);
  static constexpr bool mul_mid_ld = (bits & 1) != 0, tail_ld = (bits & 2) != 0, head_st = (bits & 4) != 0, mul_mid_st = (bits & 8) != 0,
                        ks_head_st = (bits & 16) != 0, ks_mid_ld = (bits & 32) != 0, ks_mid_st = (bits & 64) != 0;
};
template <bool NT, class T>
__device__ __forceinline__ T nt_ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, class T>
__device__ __forceinline__ void nt_st(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <bool PACK, bool NT = false>
__device__ __forceinline__ double nat_load(const double* __restrict__ region, u32 n, size_t idx) {
  if constexpr (!PACK) {
    return nt_ld<NT>(region + idx);
  } else {
    const u32 lo = nt_ld<NT>(reinterpret_cast<const u32*>(region) + idx);
    const u32 hi = nt_ld<NT>(reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(region) + 4 * (size_t)n) + idx);
    return __hiloint2double((int)(0x43380000u | hi), (int)lo) - kPackMagic;
  }
}
// the same in two steps, so that a caller can keep the loads of the NEXT item in flight while it works on this one
template <bool PACK>
struct NatRaw {
  double d;
};
template <>
struct NatRaw<true> {
  u32 lo;
  u32 hi;  // the biased 16 bits, zero-extended
};
template <bool PACK, bool NT = false>
__device__ __forceinline__ NatRaw<PACK> nat_fetch(const double* __restrict__ region, u32 n, size_t idx) {
  NatRaw<PACK> r;
  if constexpr (!PACK) {
    r.d = nt_ld<NT>(region + idx);
  } else {
    r.lo = nt_ld<NT>(reinterpret_cast<const u32*>(region) + idx);
    r.hi = nt_ld<NT>(reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(region) + 4 * (size_t)n) + idx);
  }
  return r;
}
template <bool PACK>
__device__ __forceinline__ double nat_unpack(const NatRaw<PACK>& r) {
  if constexpr (!PACK) {
    return r.d;
  } else {
    return __hiloint2double((int)(0x43380000u | r.hi), (int)r.lo) - kPackMagic;
  }
}
// PACK: v must be an integer with |v| < 2^47 (callers reduce first)
template <bool PACK, bool NT = false>
__device__ __forceinline__ void nat_store(double* __restrict__ region, u32 n, size_t idx, double v) {
  if constexpr (!PACK) {
    nt_st<NT>(region + idx, v);
  } else {
    const double m = v + kPackMagic;
    nt_st<NT>(reinterpret_cast<u32*>(region) + idx, (u32)__double2loint(m));
    nt_st<NT>(reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(region) + 4 * (size_t)n) + idx, (unsigned short)__double2hiint(m));
  }
}

// -------------------------------------------------------------------------------------------------
// Who owns which coefficients in the head / tail kernels (EdgeGeom).
// Plain degrees: a head thread t owns the NC = 2^head_log coefficients {t + k*N/NC}, a tail thread the 4 coefficients
// {t + k*N/4}.  Lane-split degrees (N = 16384): the 8 coefficients {p + k*N/8} belong to the lane PAIR (l, l + 32) of one
// wavefront -- so that each half-wave still touches 32 consecutive coefficients per access -- and each lane holds 4 of them,
// in one of two arrangements (s = 0 for lanes 0..31, 1 for lanes 32..63):
//   natural : lane s holds k = 4s + j                    -- the side facing the middle kernels (head output, tail input)
//   split   : lane s holds k = (j & 1) | (j >> 1) << 2 | s << 1, i.e. {0,1,4,5} / {2,3,6,7}  -- the coefficient-wise side
// The stage with gap N/2 (pairs k, k+4) is local in the split arrangement, the stages with gaps N/4 and N/8 in the natural
// one; between them the two lanes trade two values each with v_permlane32_swap, the wavefront-level half exchange of gfx950:
// one instruction per dword, no LDS, no selects (head_fwd_owned / tail_inv_owned).
// -------------------------------------------------------------------------------------------------
template <int L>
struct EdgeGeom {
  static constexpr bool SPLIT = lane_split(L);
  static constexpr u32 N = 1u << L;
  static constexpr int HEAD_NC = SPLIT ? 4 : (1 << head_log(L));  // coefficients a head thread owns
  static constexpr u32 HEAD_THREADS = SPLIT ? N / 4 : (N >> head_log(L));  // head threads per polynomial
  static constexpr u32 QH = SPLIT ? N / 8 : N / (u32)HEAD_NC;              // distance between the set's coefficients
  static constexpr u32 QT = SPLIT ? N / 8 : N / 4;
  static __device__ __forceinline__ u32 half(u32 t) { return (t >> 5) & 1u; }
  static __device__ __forceinline__ u32 base(u32 t) { return SPLIT ? (((t >> 6) << 5) | (t & 31u)) : t; }
  static __device__ __forceinline__ u32 k_natural(u32 t, int j) { return SPLIT ? (u32)j + 4u * half(t) : (u32)j; }
  static __device__ __forceinline__ u32 k_split(u32 t, int j) { return SPLIT ? (((u32)j & 1u) | (((u32)j >> 1) << 2) | (half(t) << 1)) : (u32)j; }
  // coefficient index of owned slot j
  static __device__ __forceinline__ u32 head_in(u32 t, int j) { return base(t) + k_split(t, j) * QH; }
  static __device__ __forceinline__ u32 head_out(u32 t, int j) { return base(t) + k_natural(t, j) * QH; }
  static __device__ __forceinline__ u32 tail_in(u32 t, int j) { return base(t) + k_natural(t, j) * QT; }
  static __device__ __forceinline__ u32 tail_out(u32 t, int j) { return base(t) + k_split(t, j) * QT; }
};

// lower half-wave's `lo_side` <-> upper half-wave's `hi_side` (v_permlane32_swap: lanes 32..63 of vdst swap with lanes 0..31 of src)
__device__ __forceinline__ void half_wave_swap(u32& hi_side, u32& lo_side) {
  const auto r = __builtin_amdgcn_permlane32_swap(hi_side, lo_side, false, false);
  hi_side = r[0];
  lo_side = r[1];
}
__device__ __forceinline__ void half_wave_swap(double& hi_side, double& lo_side) {
  u32 hl = (u32)__double2loint(hi_side), hh = (u32)__double2hiint(hi_side), ll = (u32)__double2loint(lo_side), lh = (u32)__double2hiint(lo_side);
  half_wave_swap(hl, ll);
  half_wave_swap(hh, lh);
  hi_side = __hiloint2double((int)hh, (int)hl);
  lo_side = __hiloint2double((int)lh, (int)ll);
}
__device__ __forceinline__ void half_wave_swap(u64& hi_side, u64& lo_side) {
  u32 hl = (u32)hi_side, hh = (u32)(hi_side >> 32), ll = (u32)lo_side, lh = (u32)(lo_side >> 32);
  half_wave_swap(hl, ll);
  half_wave_swap(hh, lh);
  hi_side = ((u64)hh << 32) | hl;
  lo_side = ((u64)lh << 32) | ll;
}
// natural <-> split: the lower lane gives its slots 2,3 and takes the upper lane's slots 0,1 (the map is its own inverse)
template <class V>
__device__ __forceinline__ void lane_pair_exchange(V (&v)[4]) {
  half_wave_swap(v[0], v[2]);
  half_wave_swap(v[1], v[3]);
}

template <class A, int NC>
__device__ __forceinline__ void head_fwd(const A& ar, typename A::V (&v)[NC], const typename A::Tw* __restrict__ tw);

// The twiddle of a lane-split stage depends on the half-wave only.  Loading tw[i + s] would be a VECTOR load (two distinct
// addresses per wavefront) in the middle of the dependency chain; both candidates are wave-uniform, so they come through the
// scalar cache and the lane picks one (4 v_cndmask per twiddle).
// (pinning the candidates in SGPRs keeps the compiler from turning select(load a, load b) into one divergent load)
__device__ __forceinline__ u64 pin_scalar(u64 v) {
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ double pick_tw(double a, double b, bool upper) {
  const u64 aw = pin_scalar((u64)__double_as_longlong(a)), bw = pin_scalar((u64)__double_as_longlong(b));
  return __longlong_as_double((long long)(upper ? bw : aw));
}
__device__ __forceinline__ MulOp pick_tw(const MulOp& a, const MulOp& b, bool upper) {
  const u64 aw = pin_scalar(a.w), aq = pin_scalar(a.wq), bw = pin_scalar(b.w), bq = pin_scalar(b.wq);
  MulOp r;
  r.w = upper ? bw : aw;
  r.wq = upper ? bq : aq;
  return r;
}

// the head's forward stages on the coefficients thread t owns: in = coefficient-wise side, out = middle-kernel side
template <class A, int L>
__device__ __forceinline__ void head_fwd_owned(const A& ar, typename A::V (&v)[EdgeGeom<L>::HEAD_NC], const typename A::Tw* __restrict__ tw, u32 t) {
  if constexpr (!EdgeGeom<L>::SPLIT) {
    head_fwd(ar, v, tw);
  } else {
    const bool upper = EdgeGeom<L>::half(t) != 0;
    // gap N/2 (pairs k, k+4) in the split arrangement: slots (0,2), (1,3); twiddle index 1
    ar.fwd(v[0], v[2], tw[1]);
    ar.fwd(v[1], v[3], tw[1]);
    lane_pair_exchange(v);
    // gap N/4 (pairs k, k+2) in the natural arrangement: slots (0,2), (1,3); twiddle 2 + (k >> 2) = 2 + s
    const typename A::Tw w1 = pick_tw(tw[2], tw[3], upper);
    ar.fwd(v[0], v[2], w1);
    ar.fwd(v[1], v[3], w1);
    // gap N/8 (pairs k, k+1): slots (0,1), (2,3); twiddle 4 + (k >> 1) = 4 + 2s + slot/2
    ar.fwd(v[0], v[1], pick_tw(tw[4], tw[6], upper));
    ar.fwd(v[2], v[3], pick_tw(tw[5], tw[7], upper));
  }
}

template <int L, class A>
__device__ __forceinline__ void tail_inverse4(const A& ar, typename A::V (&v)[4], const typename A::Tw* __restrict__ tw, u32 mask);

// the tail's inverse stages on the 4 values thread t holds: in = middle-kernel side, out = coefficient-wise side
template <class A, int L>
__device__ __forceinline__ void tail_inv_owned(const A& ar, typename A::V (&v)[4], const typename A::Tw* __restrict__ tw, u32 mask, u32 t) {
  if constexpr (!EdgeGeom<L>::SPLIT) {
    tail_inverse4<L, A>(ar, v, tw, mask);
  } else {
    const bool upper = EdgeGeom<L>::half(t) != 0;
    if ((mask >> 8) & 1u) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ar.reduce(v[k]);
    }
    if ((mask >> 24) & 1u) {
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = ar.reduce(v[k]);
    }
    // window [L-3, L): gap N/8 (pairs k, k+1), twiddle 4 + (k >> 1); gap N/4 (k, k+2), twiddle 2 + (k >> 2); gap N/2, twiddle 1
    ar.inv(v[0], v[1], pick_tw(tw[4], tw[6], upper));
    ar.inv(v[2], v[3], pick_tw(tw[5], tw[7], upper));
    const typename A::Tw w1 = pick_tw(tw[2], tw[3], upper);
    ar.inv(v[0], v[2], w1);
    ar.inv(v[1], v[3], w1);
    lane_pair_exchange(v);
    ar.inv(v[0], v[2], tw[1]);
    ar.inv(v[1], v[3], tw[1]);
  }
}

// the NC head outputs of thread t to their places facing the middle kernels (EdgeGeom::head_out)
template <int L, bool PACK, bool NT>
__device__ __forceinline__ void nat_store_head(double* __restrict__ region, u32 t, const double (&v)[EdgeGeom<L>::HEAD_NC]);

template <bool PACK, bool NT, int NC>
__device__ __forceinline__ void nat_store_owned(double* __restrict__ region, u32 n, u32 t, size_t Q, const double (&v)[NC]) {
#pragma unroll
  for (int k = 0; k < NC; k++) nat_store<PACK, NT>(region, n, t + (size_t)k * Q, v[k]);
}
template <int L, bool PACK, bool NT>
__device__ __forceinline__ void nat_store_head(double* __restrict__ region, u32 t, const double (&v)[EdgeGeom<L>::HEAD_NC]) {
  using G = EdgeGeom<L>;
  if constexpr (!G::SPLIT) {
    nat_store_owned<PACK, NT, G::HEAD_NC>(region, G::N, t, G::QH, v);
  } else {
#pragma unroll
    for (int k = 0; k < G::HEAD_NC; k++) nat_store<PACK, NT>(region, G::N, G::head_out(t, k), v[k]);
  }
}

// -------------------------------------------------------------------------------------------------
// Buffer addressing for the coefficient-parallel (head / tail) kernels [r05].  A head / tail thread touches the SAME per-lane
// position t in many rows and at several wave-uniform distances (k * N/8, k * N/4): with flat global pointers every access
// cost two or three VALU instructions of 64-bit address arithmetic (v_lshl_add_u64 / v_add_co + v_addc_co, plus the s_nops of
// their vcc hazard) -- 150 of mul_head<13>'s 1500 VALU instructions, 210 of mulrelin_head's, found in the ISA listing.  A
// buffer instruction takes the per-lane byte offset in ONE VGPR (shared by every access of the thread) and the whole
// wave-uniform part -- row, polynomial, k * Q -- in an SGPR offset: the address arithmetic moves to the scalar unit.
// BufRow = descriptor over a batch item's rows + the wave-uniform byte offset of one residue row inside it.  The middle
// kernels keep global pointers: their accesses already compile to the saddr + voffset form.
// -------------------------------------------------------------------------------------------------
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc buf_rsrc(const void* base) {
  // raw buffer (stride 0), no bounds: every offset is in range by construction, exactly as with the pointers this replaces
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
// an OPTIONAL operand: absent = a descriptor of zero records, whose loads return 0 without touching memory (the hardware bounds
// check) -- so the request can be issued unconditionally, beside the others, and costs nothing when there is nothing to read.
// `base` is used as it is (for an absent operand it may be null + an offset: never dereferenced).  The record count is formed by a
// SCALAR instruction inside an asm: written as `present ? ~0 : 0` it is lowered to v_cndmask (and a readfirstlane of it is dropped as
// redundant -- the value IS uniform), the descriptor ends up half in VGPRs and every load from it is wrapped in a waterfall loop
// (4 v_readfirstlane, 2 v_cmp_eq_u64, s_and_saveexec ...: 12 such loops in mulrelin_tail<13,4>, 29 in <14,8>, 8 in every ks_tail,
// found in the ISA listing).  Removing them is -6 ... -20 % static instructions in those kernels and no measurable time
// (profiles/r05_s8_no_waterfall_*.txt: a waterfall over a uniform value runs once, mostly on the scalar unit).
__device__ __forceinline__ BufRsrc buf_rsrc_opt(const void* base, bool present) {
  const u32 flag = __builtin_amdgcn_readfirstlane(present ? 1u : 0u);
  u32 records;
  asm volatile("s_sub_u32 %0, 0, %1" : "=s"(records) : "s"(flag) : "scc");  // 0 or 0xFFFFFFFF, in an SGPR
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)records, 0x00020000);
}
struct BufRow {
  BufRsrc r;
  u32 soff;  // wave-uniform byte offset of the row from the descriptor's base
};
__device__ __forceinline__ BufRow buf_row(BufRsrc r, size_t words) { return BufRow{r, (u32)(words * 8u)}; }
template <bool NT>
__device__ __forceinline__ u64 buf_ld64(BufRsrc r, u32 voff, u32 soff) {
  const auto w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, NT ? 2 : 0);
  return (u64)(u32)w[0] | ((u64)(u32)w[1] << 32);
}
template <bool NT>
__device__ __forceinline__ void buf_st64(BufRsrc r, u32 voff, u32 soff, u64 v) {
  typedef unsigned int v2u __attribute__((ext_vector_type(2)));
  v2u w;
  w[0] = (u32)v, w[1] = (u32)(v >> 32);
  __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)voff, (int)soff, NT ? 2 : 0);
}
// element `lane + uni` (per-lane part + wave-uniform part, in elements) of a row
template <bool PACK, bool NT>
__device__ __forceinline__ NatRaw<PACK> nat_fetch(const BufRow& row, u32 n, u32 lane, u32 uni) {
  NatRaw<PACK> x;
  if constexpr (!PACK) {
    x.d = __longlong_as_double((long long)buf_ld64<NT>(row.r, lane * 8u, row.soff + uni * 8u));
  } else {
    x.lo = (u32)__builtin_amdgcn_raw_buffer_load_b32(row.r, (int)(lane * 4u), (int)(row.soff + uni * 4u), NT ? 2 : 0);
    x.hi = (u32)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(row.r, (int)(lane * 2u), (int)(row.soff + 4u * n + uni * 2u), NT ? 2 : 0);
  }
  return x;
}
template <bool PACK, bool NT>
__device__ __forceinline__ void nat_store(const BufRow& row, u32 n, u32 lane, u32 uni, double v) {
  if constexpr (!PACK) {
    buf_st64<NT>(row.r, lane * 8u, row.soff + uni * 8u, (u64)__double_as_longlong(v));
  } else {
    const double m = v + kPackMagic;
    __builtin_amdgcn_raw_buffer_store_b32((u32)__double2loint(m), row.r, (int)(lane * 4u), (int)(row.soff + uni * 4u), NT ? 2 : 0);
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)__double2hiint(m), row.r, (int)(lane * 2u), (int)(row.soff + 4u * n + uni * 2u), NT ? 2 : 0);
  }
}
// EdgeGeom's four index maps as (per-lane, wave-uniform) pairs: plain degrees t + k * Q; lane-split degrees keep everything per lane
template <int L>
struct EdgeSplitIdx {
  using G = EdgeGeom<L>;
  static __device__ __forceinline__ u32 head_in_lane(u32 t, int j) { return G::SPLIT ? G::head_in(t, j) : t; }
  static __device__ __forceinline__ u32 head_out_lane(u32 t, int j) { return G::SPLIT ? G::head_out(t, j) : t; }
  static __device__ __forceinline__ u32 tail_in_lane(u32 t, int j) { return G::SPLIT ? G::tail_in(t, j) : t; }
  static __device__ __forceinline__ u32 tail_out_lane(u32 t, int j) { return G::SPLIT ? G::tail_out(t, j) : t; }
  static __device__ __forceinline__ u32 head_uni(int j) { return G::SPLIT ? 0u : (u32)j * G::QH; }
  static __device__ __forceinline__ u32 tail_uni(int j) { return G::SPLIT ? 0u : (u32)j * G::QT; }
};
template <int L, bool PACK, bool NT>
__device__ __forceinline__ NatRaw<PACK> nat_fetch_tail(const BufRow& row, u32 t, int k) {  // coefficient EdgeGeom::tail_in(t, k)
  using X = EdgeSplitIdx<L>;
  return nat_fetch<PACK, NT>(row, 1u << L, X::tail_in_lane(t, k), X::tail_uni(k));
}
template <int L, bool PACK, bool NT>
__device__ __forceinline__ void nat_store_head(const BufRow& row, u32 t, const double (&v)[EdgeGeom<L>::HEAD_NC]) {
  using X = EdgeSplitIdx<L>;
#pragma unroll
  for (int k = 0; k < EdgeGeom<L>::HEAD_NC; k++) nat_store<PACK, NT>(row, 1u << L, X::head_out_lane(t, k), X::head_uni(k), v[k]);
}
// canonical words: inputs at EdgeGeom::head_in(t, k) / outputs at EdgeGeom::tail_out(t, k)
template <int L>
__device__ __forceinline__ u64 ld_head_in(const BufRow& row, u32 t, int k) {
  using X = EdgeSplitIdx<L>;
  return buf_ld64<false>(row.r, X::head_in_lane(t, k) * 8u, row.soff + X::head_uni(k) * 8u);
}
template <int L>
__device__ __forceinline__ void st_head_out(const BufRow& row, u32 t, int k, u64 v) {  // 8-byte rows of either policy
  using X = EdgeSplitIdx<L>;
  buf_st64<false>(row.r, X::head_out_lane(t, k) * 8u, row.soff + X::head_uni(k) * 8u, v);
}
template <int L>
__device__ __forceinline__ u64 ld_tail_in(const BufRow& row, u32 t, int k) {
  using X = EdgeSplitIdx<L>;
  return buf_ld64<false>(row.r, X::tail_in_lane(t, k) * 8u, row.soff + X::tail_uni(k) * 8u);
}
template <int L>
__device__ __forceinline__ u64 ld_tail_out(const BufRow& row, u32 t, int k) {
  using X = EdgeSplitIdx<L>;
  return buf_ld64<false>(row.r, X::tail_out_lane(t, k) * 8u, row.soff + X::tail_uni(k) * 8u);
}
template <int L>
__device__ __forceinline__ void st_tail_out(const BufRow& row, u32 t, int k, u64 v) {
  using X = EdgeSplitIdx<L>;
  buf_st64<false>(row.r, X::tail_out_lane(t, k) * 8u, row.soff + X::tail_uni(k) * 8u, v);
}

#ifndef KS_XCD_ROWS
#define KS_XCD_ROWS 1  // (experiment hook: 0 = the x-fastest order of the rotation head / tail, r06 s3 ... s31)
#endif
// coefficient j of sigma_g(p) mod q, p a canonical residue row at byte offset soff of the descriptor: +- p[j * g^-1 mod 2N]
template <int L>
__device__ __forceinline__ u64 galois_gather(BufRsrc r, u32 soff, u32 j, u32 ginv, u64 q) {
  constexpr u32 N = 1u << L;
  const u32 i2 = (j * ginv) & (2u * N - 1u);
  const u64 w = buf_ld64<false>(r, (i2 & (N - 1u)) * 8u, soff);
  return (i2 & N) && w ? q - w : w;
}

__device__ __forceinline__ bool residue_is_f64(const DevMod& dm) { return dm.use_f64 && dm.split_ok; }
template <class A, int NC>
__device__ __forceinline__ void head_fwd(const A& ar, typename A::V (&v)[NC], const typename A::Tw* __restrict__ tw);

// one row of T out of a key-switch head: 8 bytes per value, or -- the rows of key prime I are packed (wave-uniform) -- 6, reduced
// first when the head's outputs exceed the packed range (context.cpp plan_f64_split)
template <int L, int PACK>
__device__ __forceinline__ void ks_head_store(const DevCtx* __restrict__ ctx, const DevMod& dm, const ArithD& ar, const BufRow& row, u32 I, u32 t,
                                              double (&v)[EdgeGeom<L>::HEAD_NC]) {
  constexpr int NC = EdgeGeom<L>::HEAD_NC;
  if constexpr (PACK == 0) {
    nat_store_head<L, false, NtSites<L>::ks_head_st>(row, t, v);
  } else {
    if (PACK == 1 || ((ctx->ks_row_mask >> I) & 1u) != 0) {
      if (dm.split_fwd_mask & kPlanStoreReduce) {
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] = ar.reduce(v[k]);
      }
      nat_store_head<L, true, NtSites<L>::ks_head_st>(row, t, v);
    } else {
      nat_store_head<L, false, NtSites<L>::ks_head_st>(row, t, v);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// key switch, head: T[op][I][J] = first three forward stages over q_I of (target_J mod q_I)
// grid: (N/8/256, K, ops)
// MIXED: some key primes take the integer policy (user primes >= 2^50, e.g. CoeffModulus::create(8192,[54,54,54,56])):
// their rows of T hold lazy u64 values in [0, 4q) instead of doubles; the policy is a property of the residue I, so the
// branch is wave-uniform.  MIXED = false is the all-FP64 kernel (every SEAL default parameter set).
// -------------------------------------------------------------------------------------------------
// PACK: DevCtx::pack_ks (0: 8-byte rows, 1: every row 48-bit packed, 2: per key prime, DevCtx::ks_row_mask)
template <int L, int PACK, bool MIXED>
__global__ __launch_bounds__(kHeadThreads) void ks_head_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                               const u64* __restrict__ target, size_t tstride, double* __restrict__ T, u32 ginv) {
  using G = EdgeGeom<L>;
  constexpr int NC = G::HEAD_NC;
  constexpr u32 N = 1u << L;
  u32 bx = blockIdx.x, J = blockIdx.y, op = blockIdx.z;
  if (KS_XCD_ROWS && ginv) {
    // r06 s32: a rotation's head GATHERS its digit row through the automorphism -- every workgroup of a row touches lines all over the
    // row.  Workgroups go to the 8 XCDs round robin in dispatch order (x fastest), so the TB workgroups of one row landed on TB different
    // XCDs and every one of their L2s fetched the row from HBM: 4.9 MB read per item at N = 16384 where 1.05 MB is compulsory
    // (profiles/r06_final_dot_prod_n16384_pmc_fetch.txt).  Dispatch index 8q + xcd -> workgroup q mod TB of row 8 (q / TB) + xcd: a row's
    // workgroups run back to back on ONE XCD.  (Wave-uniform; rows not a multiple of 8: the plain order.)
    const u32 TB = gridDim.x, R = gridDim.y * gridDim.z;
    if ((R & 7u) == 0u) {
      const u32 d = blockIdx.x + TB * (blockIdx.y + gridDim.y * blockIdx.z), q = d >> 3, row = (q / TB) * 8u + (d & 7u);
      bx = q % TB, J = row % gridDim.y, op = row / gridDim.y;
    }
  }
  const u32 t = bx * kHeadThreads + threadIdx.x;
  const u32 K = ctx->K, KK = ctx->KK;
  const u64* src = target + (size_t)op * tstride + (size_t)J * N;
  const BufRsrc rout = buf_rsrc(T + (size_t)op * KK * K * N);  // buffer addressing: see BufRow
  const u64 qJ = ctx->mod[J].q;
  u64 x[NC];
  if (ginv) {
    // a rotation's key switch: the target is sigma_g(c1), read THROUGH the automorphism (SEAL apply_galois, seal_fhe/src/
    // bfv_evaluator.rs:177-247) instead of from a rotated copy: coefficient j of sigma_g(p) is +- p[j * g^-1 mod 2N] (minus when the
    // index passes N).  An 8-byte gather per value; every line is used in full by the workgroups of this row (L2 absorbs it).
#pragma unroll
    for (int k = 0; k < NC; k++) x[k] = galois_gather<L>(buf_rsrc(src), 0u, EdgeSplitIdx<L>::head_in_lane(t, k) + EdgeSplitIdx<L>::head_uni(k), ginv, qJ);
  } else {
#pragma unroll
    for (int k = 0; k < NC; k++) x[k] = ld_head_in<L>(buf_row(buf_rsrc(src), 0), t, k);
  }
  for (u32 I = 0; I < KK; I++) {
    const DevMod& dm = ctx->mod[I];
    if constexpr (MIXED) {
      if (!residue_is_f64(dm)) {
        const ArithI ai(dm);
        const bool shrink = qJ > dm.q;
        u64 w[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) w[k] = shrink ? reduce64(x[k], dm) : x[k];
        head_fwd_owned<ArithI, L>(ai, w, twf_base + (size_t)I * N, t);
#pragma unroll
        for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, ((size_t)I * K + J) * N), t, k, w[k]);
        continue;
      }
    }
    const ArithD ar(dm);
    const double* tw = reinterpret_cast<const double*>(twf_base + (size_t)I * N);
    const bool need_reduce = qJ > dm.q;
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) {
      if constexpr (MIXED) {
        // the digit may come from an integer-policy prime (up to 61 bits): from_u64 is exact below 2^52 only, so bring it
        // under q_I with integer arithmetic first
        v[k] = ar.from_u64(need_reduce ? reduce64(x[k], dm) : x[k]);
      } else {
        const double d = ar.from_u64(x[k]);
        v[k] = need_reduce ? ar.reduce(d) : d;
      }
    }
    head_fwd_owned<ArithD, L>(ar, v, tw, t);
    ks_head_store<L, PACK>(ctx, dm, ar, buf_row(rout, ((size_t)I * K + J) * N), I, t, v);
  }
}

// Per-item keys (kernels.hpp KeyMap; multi-tenant batches).  The launch's items arrive sorted by key in km.order; each XCD takes a
// CONTIGUOUS run of that order (position w = xcd * per + slot % per) instead of every eighth item, so the key rows of one client
// are fetched into one XCD's L2 (two at a run boundary) and not into all eight.  Wave-uniform: two scalar loads per workgroup.
typedef unsigned long long key2_t __attribute__((ext_vector_type(2)));  // two key words per load (members .x, .y)
#define KS_KEYMAP_WALK(km, op, key, xcd, slot, per, ops) \
  if ((km).keys) {                                        \
    const u32 w__ = (xcd) * (per) + (slot) % (per);       \
    if (w__ >= (ops)) return;                             \
    const uint2 m__ = (km).order[w__];                    \
    op = m__.x;                                           \
    key = (km).keys[m__.y];                              \
  }

// -------------------------------------------------------------------------------------------------
// key switch, middle: per (op, I, block): finish the K forward transforms, multiply-accumulate with the
// key rows, run the block-local inverse stages of both accumulators.
// grid: ops8 * KK * NBLK (slice-major per XCD, see the index computation)
// -------------------------------------------------------------------------------------------------
// residues / nres: the key-prime indices this launch handles (DevCtx::ks_res_d: the FP64-policy ones, all of them for the
// SEAL default sets; the integer-policy ones go through ks_mid_int_kernel below)
// EPT (KS_EPT(L), experiment hook): elements per thread; 16 = two radix-8 groups per thread and half the threads per workgroup
// (N = 16384: 256-thread workgroups, of which two fit a CU's registers without the 128-VGPR ceiling of two 512-thread ones)
// PACK: the rows of T this launch reads are 48-bit packed; PACK_ACC: so are the accumulator rows it writes.  Per-row contexts
// (DevCtx::pack_ks == 2) pack T only, and launch this kernel once per kind of residue (ks_res_dp / ks_res_d):
//  - the tails read every accumulator row as doubles -- a per-row fetch cost them more than 6 of 18 shorter rows saved
//    (profiles/r06_s22_ab_ksrows_v1_*.txt);
//  - ONE launch that tests the row's kind per workgroup spilt 27 registers into the digit loop and ran 27 % slower (..._v3_*.txt).
template <int L, bool PACK, int EPT = KS_EPT(L), bool PACK_ACC = PACK>
__global__ __launch_bounds__((SplitShape<L, EPT>::TPB), KS_MID_WAVES(L)) void ks_mid_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                       const MulOp* __restrict__ twi_base, const double* __restrict__ T,
                                                                       const u64* __restrict__ key, double* __restrict__ ACC, u32 ops,
                                                                       const unsigned char* __restrict__ residues, u32 nres, KeyMap km) {
  using Sh = SplitShape<L, EPT>;
  using A = ArithD;
  __shared__ double smem[KS_GROUP_MAX(L) * Sh::BLOCK];
  const u32 tid = threadIdx.x;
  const u32 K = ctx->K, KK = ctx->KK;
  const u32 b = blockIdx.x;
  // Workgroups are dealt round-robin to the 8 XCDs.  Within one XCD consecutive workgroups walk the ops of ONE
  // (I, blk) slice, so the K * 2 key rows of that slice (the only re-used global data) stay in that XCD's L2.
  const u32 xcd = b & 7u, slot = b >> 3;
  const u32 per = (ops + 7u) >> 3;
  u32 op = (slot % per) * 8u + xcd;
  const u32 ib = slot / per;
  const u32 blk = ib % Sh::NBLK;
  const u32 I = residue_of(residues, ib / Sh::NBLK);
  (void)nres;
  KS_KEYMAP_WALK(km, op, key, xcd, slot, per, ops)
  if (op >= ops) return;
  const DevMod& dm = ctx->mod[I];
  const A ar(dm);
  const double* twf = reinterpret_cast<const double*>(twf_base + (size_t)I * Sh::N);
  const double* twi = reinterpret_cast<const double*>(twi_base + (size_t)I * Sh::N);
  constexpr int RF0 = split_fwd_radix(L, 0), LOWF0 = split_fwd_low(L, 0);
  using First = BlkPass<A, L, LOWF0, RF0, EPT>;
  constexpr int RL = split_fwd_radix(L, Sh::NPF - 1);  // last forward window: LOW = 0
  using Last = BlkPass<A, L, 0, RL, EPT>;
  double acc[2][EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) acc[0][e] = 0.0, acc[1][e] = 0.0;
  auto load_src = [&](u32 J, double(&dst)[EPT]) {
    const double* src = T + (((size_t)op * KK + I) * K + J) * Sh::N;
#pragma unroll
    for (int g = 0; g < First::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RF0); k++) dst[g * (1 << RF0) + k] = nat_load<PACK, NtSites<L>::ks_mid_ld>(src, Sh::N, First::elem(tid, blk, g, k));
  };
  // key rows of digit J for the group g of the last forward window (the elements this thread holds): 16-byte loads
  auto mac = [&](u32 J, const double(&v)[EPT]) {
    // (the key pointer may come out of the per-item key table: a loaded pointer is generic to the compiler -- name the address space,
    // or every key load is a flat_load; tests/test_isa_guards_cpu.py)
    const global_ptr<const u64> k0 = as_global(key) + (((size_t)J * 2 + 0) * KK + I) * Sh::N;
    const global_ptr<const u64> k1 = as_global(key) + (((size_t)J * 2 + 1) * KK + I) * Sh::N;
#pragma unroll
    for (int g = 0; g < Last::G; g++) {
      const u32 base = Last::elem(tid, blk, g, 0);
      constexpr int W = 1 << RL;
      constexpr int CH = KS_MAC_CHUNK(L) < W ? KS_MAC_CHUNK(L) : W;  // key words in flight per step (register diet at N = 16384)
#pragma unroll
      for (int c0 = 0; c0 < W; c0 += CH) {
        key2_t ka[CH / 2], kc[CH / 2];
#pragma unroll
        for (int k = 0; k < CH; k += 2) {
          ka[k / 2] = *reinterpret_cast<global_ptr<const key2_t>>(k0 + base + c0 + k);
          kc[k / 2] = *reinterpret_cast<global_ptr<const key2_t>>(k1 + base + c0 + k);
        }
#pragma unroll
        for (int h = 0; h < CH / 2; h++) {
          const int e = g * W + c0 + 2 * h;
          acc[0][e] += ar.mul_var(v[e], ar.from_u64(ka[h].x));
          acc[0][e + 1] += ar.mul_var(v[e + 1], ar.from_u64(ka[h].y));
          acc[1][e] += ar.mul_var(v[e], ar.from_u64(kc[h].x));
          acc[1][e + 1] += ar.mul_var(v[e + 1], ar.from_u64(kc[h].y));
        }
      }
    }
    if ((J & 3u) == 3u) {
      reduce_all<A, EPT>(ar, acc[0]);
      reduce_all<A, EPT>(ar, acc[1]);
    }
  };
  // digits are transformed in groups of 4 / 2 / 1, each group advanced pass by pass (mid_forward_multi)
  auto group = [&](u32 J0, auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    double v[NP][EPT];
#pragma unroll
    for (int i = 0; i < NP; i++) load_src(J0 + i, v[i]);
    if (J0 > 0) __syncthreads();  // the previous group's last pass may still be reading LDS
    // The twiddles do not depend on the group: without this the compiler hoists every twiddle load of all passes
    // out of the loop and keeps ~100 registers of them alive.  Re-materialise the pointer.
    const double* twf_j = opaque_uniform(twf);  // (an opaque OFFSET: the pointer keeps its address space, nttcore.hpp)
    mid_forward_multi<A, L, NP, EPT, KS_TW_PIPE(L)>(ar, v, smem, tid, blk, twf_j, dm.split_fwd_mask);
#pragma unroll
    for (int i = 0; i < NP; i++) mac(J0 + i, v[i]);
  };
  u32 J = 0;
  if constexpr (KS_GROUP_MAX(L) >= 4)
    for (; J + 4 <= K; J += 4) group(J, std::integral_constant<int, 4>{});
  if constexpr (KS_GROUP_MAX(L) >= 2)
    for (; J + 2 <= K; J += 2) group(J, std::integral_constant<int, 2>{});
  for (; J < K; J++) group(J, std::integral_constant<int, 1>{});
  if ((K & 3u) != 0u) {  // (the accumulation reduces after every fourth digit: K = 4, 8 arrive here reduced)
    reduce_all<A, EPT>(ar, acc[0]);
    reduce_all<A, EPT>(ar, acc[1]);
  }
  constexpr int RI = split_inv_radix(L, Sh::NPI - 1), LOWI = split_inv_low(L, Sh::NPI - 1);
  using Out = BlkPass<A, L, LOWI, RI, EPT>;
  __syncthreads();
  if constexpr (KS_GROUP_MAX(L) >= 2) {
    mid_inverse_multi<A, L, 2, EPT, KS_TW_PIPE(L)>(ar, acc, smem, tid, blk, twi, dm.split_inv_mask);
  } else {
    using One = double[1][EPT];
    mid_inverse_multi<A, L, 1, EPT, KS_TW_PIPE(L)>(ar, *reinterpret_cast<One*>(&acc[0]), smem, tid, blk, twi, dm.split_inv_mask);
    __syncthreads();
    mid_inverse_multi<A, L, 1, EPT, KS_TW_PIPE(L)>(ar, *reinterpret_cast<One*>(&acc[1]), smem, tid, blk, twi, dm.split_inv_mask);
  }
  if constexpr (PACK_ACC) {
    if (dm.split_inv_mask & kPlanStoreReduce) {  // primes whose last pass leaves more than a packed row holds (context.cpp plan_f64_split)
      reduce_all<A, EPT>(ar, acc[0]);
      reduce_all<A, EPT>(ar, acc[1]);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    double* dst = ACC + (((size_t)op * 2 + c) * KK + I) * Sh::N;
#pragma unroll
    for (int g = 0; g < Out::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RI); k++) nat_store<PACK_ACC, NtSites<L>::ks_mid_st>(dst, Sh::N, Out::elem(tid, blk, g, k), acc[c][g * (1 << RI) + k]);
  }
}


// The same for integer-policy key primes (Harvey butterflies on lazy u64 values, Shoup twiddles): the digit rows arrive as
// u64 in [0, 4q) from the head's integer branch; the products with the key rows are summed in 128 bits and reduced once per
// four digits (4 * 4q * q < 2^126 for q < 2^61); the accumulators leave as lazy u64 in [0, 2q).
#define KS_MID_INT_GROUP_OF(L) ((L) >= 15 ? 1 : 2)  // N = 32768: one 64 KB exchange region
// N = 32768 [r06]: a block is 8192 coefficients = ONE 1024-thread workgroup per CU with 128 registers per lane, where this kernel
// wants ~250: it spills ~300 bytes per lane.  Measured (interleaved, profiles/r06_s6_*): as it is 20.7 ms per 256 ops; without the
// twiddle look-ahead and with 64-bit sums reduced per product (216 bytes of scratch) 21.7; 512 threads of 16 elements (256
// registers, 80 bytes) 22.2; the next digit's rows requested behind the key words 28.6 -- the time does not follow the registers:
// with one resident workgroup per CU nothing overlaps a digit's load -> four exchanges -> key loads chain.
#ifndef KS_MID_INT_EPT15
#define KS_MID_INT_EPT15 8
#endif
#define KS_MID_INT_EPT(L) ((L) >= 15 ? KS_MID_INT_EPT15 : kBlkEPT)
#ifndef KS_MID_INT_NARROW15
#define KS_MID_INT_NARROW15 0
#endif
template <int L, int EPT = KS_MID_INT_EPT(L)>
__global__ __launch_bounds__((SplitShape<L, EPT>::TPB), (SplitShape<L, EPT>::TPB >= 1024 ? 4 : 2)) void ks_mid_int_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                             const MulOp* __restrict__ twi_base, const u64* __restrict__ T,
                                                                             const u64* __restrict__ key, u64* __restrict__ ACC, u32 ops,
                                                                             const unsigned char* __restrict__ residues, u32 nres, KeyMap km) {
  using Sh = SplitShape<L, EPT>;
  using A = ArithI;
  constexpr int KS_MID_INT_GROUP = KS_MID_INT_GROUP_OF(L);
  constexpr bool NARROW = L >= 15 && KS_MID_INT_NARROW15 != 0;  // experiment hook: 64-bit sums, reduced per product
  constexpr bool PIPE = !NARROW;                                 // next pass's twiddles held as well
  __shared__ u64 smem[KS_MID_INT_GROUP * Sh::BLOCK];
  const u32 tid = threadIdx.x;
  const u32 K = ctx->K, KK = ctx->KK;
  const u32 b = blockIdx.x;
  const u32 xcd = b & 7u, slot = b >> 3;
  const u32 per = (ops + 7u) >> 3;
  u32 op = (slot % per) * 8u + xcd;
  const u32 ib = slot / per;
  const u32 blk = ib % Sh::NBLK;
  const u32 I = residue_of(residues, ib / Sh::NBLK);
  (void)nres;
  KS_KEYMAP_WALK(km, op, key, xcd, slot, per, ops)
  if (op >= ops) return;
  const DevMod& dm = ctx->mod[I];
  const A ar(dm);
  const MulOp* twf = twf_base + (size_t)I * Sh::N;
  const MulOp* twi = twi_base + (size_t)I * Sh::N;
  constexpr int RF0 = split_fwd_radix(L, 0), LOWF0 = split_fwd_low(L, 0);
  using First = BlkPass<A, L, LOWF0, RF0, EPT>;
  constexpr int RL = split_fwd_radix(L, Sh::NPF - 1);
  using Last = BlkPass<A, L, 0, RL, EPT>;
  using Wide = typename std::conditional<NARROW, u64, u128>::type;
  Wide wide[2][EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) wide[0][e] = 0, wide[1][e] = 0;
  auto load_src = [&](u32 J, u64(&dst)[EPT]) {
    const u64* src = T + (((size_t)op * KK + I) * K + J) * Sh::N;
#pragma unroll
    for (int g = 0; g < First::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RF0); k++) dst[g * (1 << RF0) + k] = nt_ld<NtSites<L>::ks_mid_ld>(src + First::elem(tid, blk, g, k));
  };
  auto mac = [&](u32 J, const u64(&v)[EPT]) {
    const global_ptr<const u64> k0 = as_global(key) + (((size_t)J * 2 + 0) * KK + I) * Sh::N;
    const global_ptr<const u64> k1 = as_global(key) + (((size_t)J * 2 + 1) * KK + I) * Sh::N;
#pragma unroll
    for (int g = 0; g < Last::G; g++) {
      const u32 base = Last::elem(tid, blk, g, 0);
      constexpr int W = 1 << RL;
#pragma unroll
      for (int k = 0; k < W; k += 2) {
        const key2_t ka = *reinterpret_cast<global_ptr<const key2_t>>(k0 + base + k);
        const key2_t kc = *reinterpret_cast<global_ptr<const key2_t>>(k1 + base + k);
        const int e = g * W + k;
        if constexpr (NARROW) {
          wide[0][e] = ar.mul_add(v[e], ka.x, wide[0][e]);
          wide[0][e + 1] = ar.mul_add(v[e + 1], ka.y, wide[0][e + 1]);
          wide[1][e] = ar.mul_add(v[e], kc.x, wide[1][e]);
          wide[1][e + 1] = ar.mul_add(v[e + 1], kc.y, wide[1][e + 1]);
        } else {
          wide[0][e] += (u128)v[e] * ka.x;
          wide[0][e + 1] += (u128)v[e + 1] * ka.y;
          wide[1][e] += (u128)v[e] * kc.x;
          wide[1][e + 1] += (u128)v[e + 1] * kc.y;
        }
      }
    }
    if constexpr (!NARROW) {
      if ((J & 3u) == 3u) {
#pragma unroll
        for (int e = 0; e < EPT; e++) wide[0][e] = reduce128(wide[0][e], dm), wide[1][e] = reduce128(wide[1][e], dm);
      }
    }
  };
  auto group = [&](u32 J0, auto np_tag) {
    constexpr int NP = decltype(np_tag)::value;
    u64 v[NP][EPT];
#pragma unroll
    for (int i = 0; i < NP; i++) load_src(J0 + i, v[i]);
    if (J0 > 0) __syncthreads();
    const MulOp* twf_j = opaque_uniform(twf);
    mid_forward_multi<A, L, NP, EPT, PIPE>(ar, v, smem, tid, blk, twf_j, 0u);
#pragma unroll
    for (int i = 0; i < NP; i++) mac(J0 + i, v[i]);
  };
  u32 J = 0;
  if constexpr (KS_MID_INT_GROUP >= 2)
    for (; J + 2 <= K; J += 2) group(J, std::integral_constant<int, 2>{});
  for (; J < K; J++) group(J, std::integral_constant<int, 1>{});
  u64 acc[2][EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    if constexpr (NARROW) acc[0][e] = (u64)wide[0][e], acc[1][e] = (u64)wide[1][e];
    else acc[0][e] = reduce128((u128)wide[0][e], dm), acc[1][e] = reduce128((u128)wide[1][e], dm);
  }
  constexpr int RI = split_inv_radix(L, Sh::NPI - 1), LOWI = split_inv_low(L, Sh::NPI - 1);
  using Out = BlkPass<A, L, LOWI, RI, EPT>;
  __syncthreads();
  if constexpr (KS_MID_INT_GROUP >= 2) {
    mid_inverse_multi<A, L, 2, EPT, PIPE>(ar, acc, smem, tid, blk, twi, 0u);
  } else {
    using One = u64[1][EPT];
    mid_inverse_multi<A, L, 1, EPT, PIPE>(ar, *reinterpret_cast<One*>(&acc[0]), smem, tid, blk, twi, 0u);
    __syncthreads();
    mid_inverse_multi<A, L, 1, EPT, PIPE>(ar, *reinterpret_cast<One*>(&acc[1]), smem, tid, blk, twi, 0u);
  }
#pragma unroll
  for (int c = 0; c < 2; c++) {
    u64* dst = ACC + (((size_t)op * 2 + c) * KK + I) * Sh::N;
#pragma unroll
    for (int g = 0; g < Out::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RI); k++) nt_st<NtSites<L>::ks_mid_st>(dst + Out::elem(tid, blk, g, k), acc[c][g * (1 << RI) + k]);
  }
}

// -------------------------------------------------------------------------------------------------
// key switch, tail: last two inverse stages + n^{-1} on the coefficients {t + k*N/4}, then SEAL's mod-down
// by the special prime with rounding, added to the base ciphertext.
// grid: (N/4/256, 2, ops)
// -------------------------------------------------------------------------------------------------
template <int L, class A = ArithD>
__device__ __forceinline__ void tail_inverse4(const A& ar, typename A::V (&v)[4], const typename A::Tw* __restrict__ tw, u32 mask) {
  if ((mask >> 8) & 1u) {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = ar.reduce(v[k]);
  }
  if ((mask >> 24) & 1u) {
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = ar.reduce(v[k]);
  }
  // window [L-2, L): stage j = 0 pairs (0,1),(2,3) with twiddles 2 + (k>>1); stage j = 1 pairs (0,2),(1,3) with twiddle 1
  ar.inv(v[0], v[1], tw[2]);
  ar.inv(v[2], v[3], tw[3]);
  ar.inv(v[0], v[2], tw[1]);
  ar.inv(v[1], v[3], tw[1]);
}

// The special-prime residue of a key-switch accumulator as the mod-down wants it: tl = (a * n^-1 mod p + floor(p/2)) mod p, the
// canonical integer in [0, p), as a double (moddown_d.hpp) -- formed in FP64 from the four values the middle kernel left
// (r05: it used to go through a canonical u64 and a 64-bit add_mod, and back)
template <int L>
__device__ __forceinline__ void tail_special_d(const DevCtx* __restrict__ ctx, const DevMod& sp, double (&v)[4], const double* __restrict__ tw, u32 t,
                                               double (&tld)[4]) {
  const ArithD ar(sp);
  tail_inv_owned<ArithD, L>(ar, v, tw, sp.split_inv_mask, t);
  const double half = ArithD::from_u64(ctx->qsp_half);
  double s[4];
#pragma unroll
  for (int k = 0; k < 4; k++) s[k] = ar.mul_const(v[k], sp.ninv_d);
  if (sp.split_inv_mask & kPlanScaleReduce) {
    HIPBFV_KEEP_BRANCH();
#pragma unroll
    for (int k = 0; k < 4; k++) s[k] = ar.reduce(s[k]);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    double c = s[k] < 0.0 ? s[k] + ar.q : s[k];  // canonical: |s| < p
    c += half;
    tld[k] = c >= ar.q ? c - ar.q : c;
  }
}

// MIXED: rows of ACC that belong to integer-policy key primes hold lazy u64 values in [0, 2q) (ks_mid_int_kernel)
template <int L, bool PACK, bool MIXED>
__global__ __launch_bounds__(kHeadThreads) void ks_tail_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                               const double* __restrict__ ACC, const u64* __restrict__ base, size_t bstride,
                                                               u32 base_mask, const u64* __restrict__ extra, u64* __restrict__ out, u32 ginv) {
  using G = EdgeGeom<L>;
  constexpr u32 N = 1u << L;
  u32 bx = blockIdx.x, c = blockIdx.y, op = blockIdx.z;
  if (KS_XCD_ROWS && ginv) {  // a rotation's tail gathers sigma_g(c0): the workgroups of one (polynomial, item) on one XCD (ks_head_kernel)
    const u32 TB = gridDim.x, R = gridDim.y * gridDim.z;
    if ((R & 7u) == 0u) {
      const u32 d = blockIdx.x + TB * (blockIdx.y + gridDim.y * blockIdx.z), q = d >> 3, row = (q / TB) * 8u + (d & 7u);
      bx = q % TB, c = row % gridDim.y, op = row / gridDim.y;
    }
  }
  const u32 t = bx * kHeadThreads + threadIdx.x;
  const u32 K = ctx->K, KK = ctx->KK;
  const double* acc = ACC + ((size_t)op * 2 + c) * KK * N;
  // special prime first
  if constexpr (!MIXED) {
    const DevMod& sp = ctx->mod[KK - 1];
    // buffer addressing (BufRow): the accumulator rows, the two optional operands (absent = a descriptor of zero records) and the output
    const bool has_base = ((base_mask >> c) & 1u) != 0;
    const BufRsrc racc = buf_rsrc(acc), rout = buf_rsrc(out + ((size_t)op * 2 + c) * K * N);
    const BufRsrc rbase = buf_rsrc_opt(base + (size_t)op * bstride + (size_t)c * K * N, has_base);
    const BufRsrc rex = buf_rsrc_opt(extra + ((size_t)op * 2 + c) * K * N, extra != nullptr);
    double v[4], tld[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACK>(nat_fetch_tail<L, PACK, NtSites<L>::tail_ld>(buf_row(racc, (size_t)(KK - 1) * N), t, k));
    tail_special_d<L>(ctx, sp, v, reinterpret_cast<const double*>(twi_base + (size_t)(KK - 1) * N), t, tld);
    const u64 qsp = sp.q;
    // all-FP64 key primes: accumulator row J + 1 is requested before row J is finished, and the base / addend words of a row
    // are requested as one group at its start (r04: every one of them used to sit alone behind its own test and wait)
    NatRaw<PACK> cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nat_fetch_tail<L, PACK, NtSites<L>::tail_ld>(buf_row(racc, 0), t, k);
    for (u32 J = 0; J < K; J++) {
      const u32 Jn = J + 1 < K ? J + 1 : J;
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = nat_fetch_tail<L, PACK, NtSites<L>::tail_ld>(buf_row(racc, (size_t)Jn * N), t, k);
      const DevMod& mj = ctx->mod[J];
      const ArithD ar(mj);
      // UNCONDITIONAL requests (an absent operand is a descriptor of zero records: its loads return 0 without a memory access): a
      // load under `if (extra)` ends its basic block in a copy of the loaded value, i.e. in an s_waitcnt vmcnt(0) that also waits
      // for the row just requested
      u64 bw[4], ex[4];
      if (ginv) {  // a rotation: the base is sigma_g(c0), read through the automorphism (ks_head_kernel)
#pragma unroll
        for (int k = 0; k < 4; k++)
          bw[k] = galois_gather<L>(rbase, (u32)((size_t)J * N * 8u), EdgeSplitIdx<L>::tail_out_lane(t, k) + EdgeSplitIdx<L>::tail_uni(k), ginv, mj.q);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) bw[k] = ld_tail_out<L>(buf_row(rbase, (size_t)J * N), t, k);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) ex[k] = ld_tail_out<L>(buf_row(rex, (size_t)J * N), t, k);
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACK>(cur[k]);
      tail_inv_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twi_base + (size_t)J * N), mj.split_inv_mask, t);
      // the mod-down in exact FP64 on the representative the transform leaves (moddown_d.hpp): no canonical u64 of `a`, no 64-bit
      // Barrett / Shoup chain per output value
      const MulOpD iw = ctx->inv_qsp_mod_q_d[J];
      const double hf = ctx->qsp_half_mod_q_d[J];
      const bool p_above_q = qsp > mj.q;
      double s[4];
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = ar.mul_const(v[k], mj.ninv_d);
      if (mj.split_inv_mask & kPlanScaleReduce) {  // mod_down_d wants |s| <= 2q
        HIPBFV_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = ar.reduce(s[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const double bd = ArithD::from_u64(bw[k]) + ArithD::from_u64(ex[k]);  // the base ciphertext + a fused Add node's (absent: 0)
        const double r = mod_down_d(ar.q, ar.qinv, iw.w, iw.wq, hf, p_above_q, s[k], tld[k], bd);
        st_tail_out<L>(buf_row(rout, (size_t)J * N), t, k, ArithD::to_bits(r));
      }
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    return;
  }
  // MIXED (8-byte rows of either policy): the special prime's residue as a canonical u64
  const bool has_base = ((base_mask >> c) & 1u) != 0;
  const BufRsrc racc = buf_rsrc(acc), rout = buf_rsrc(out + ((size_t)op * 2 + c) * K * N);  // buffer addressing: see BufRow
  const BufRsrc rbase = buf_rsrc_opt(base + (size_t)op * bstride + (size_t)c * K * N, has_base);
  const BufRsrc rex = buf_rsrc_opt(extra + ((size_t)op * 2 + c) * K * N, extra != nullptr);
  u64 tl[4];
  bool sp_done = false;
  {
    const DevMod& sp = ctx->mod[KK - 1];
    if (!residue_is_f64(sp)) {
      const ArithI ai(sp);
      u64 w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = ld_tail_in<L>(buf_row(racc, (size_t)(KK - 1) * N), t, k);
      tail_inv_owned<ArithI, L>(ai, w, twi_base + (size_t)(KK - 1) * N, 0u, t);
#pragma unroll
      for (int k = 0; k < 4; k++) tl[k] = add_mod(ai.scale_canonical(w[k], sp.ninv), ctx->qsp_half, sp.q);
      sp_done = true;
    }
  }
  if (!sp_done) {
    const DevMod& sp = ctx->mod[KK - 1];
    const ArithD ar(sp);
    const double* tw = reinterpret_cast<const double*>(twi_base + (size_t)(KK - 1) * N);
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACK>(nat_fetch_tail<L, PACK, NtSites<L>::tail_ld>(buf_row(racc, (size_t)(KK - 1) * N), t, k));
    tail_inv_owned<ArithD, L>(ar, v, tw, sp.split_inv_mask, t);
#pragma unroll
    for (int k = 0; k < 4; k++) tl[k] = add_mod(ar.scale_canonical(v[k], sp.ninv_d), ctx->qsp_half, sp.q);
  }
  const u64 qsp = ctx->mod[KK - 1].q;
  // MIXED (8-byte rows of either policy): the same pipelining on raw words -- row J + 1 requested before row J is finished, the
  // base / addend words requested unconditionally at the start of the row
  {
    u64 cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = ld_tail_in<L>(buf_row(racc, 0), t, k);
    for (u32 J = 0; J < K; J++) {
      const u32 Jn = J + 1 < K ? J + 1 : J;
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(buf_row(racc, (size_t)Jn * N), t, k);
      const DevMod& mj = ctx->mod[J];
      u64 bw[4], ex[4];  // (absent operands: descriptors of zero records, the loads return 0)
      if (ginv) {
#pragma unroll
        for (int k = 0; k < 4; k++)
          bw[k] = galois_gather<L>(rbase, (u32)((size_t)J * N * 8u), EdgeSplitIdx<L>::tail_out_lane(t, k) + EdgeSplitIdx<L>::tail_uni(k), ginv, mj.q);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) bw[k] = ld_tail_out<L>(buf_row(rbase, (size_t)J * N), t, k);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) ex[k] = ld_tail_out<L>(buf_row(rex, (size_t)J * N), t, k);
      u64 av[4];
      if (!residue_is_f64(mj)) {
        const ArithI ai(mj);
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = cur[k];
        tail_inv_owned<ArithI, L>(ai, w, twi_base + (size_t)J * N, 0u, t);
#pragma unroll
        for (int k = 0; k < 4; k++) av[k] = ai.scale_canonical(w[k], mj.ninv);
      } else {
        const ArithD ar(mj);
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __longlong_as_double((long long)cur[k]);
        tail_inv_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twi_base + (size_t)J * N), mj.split_inv_mask, t);
#pragma unroll
        for (int k = 0; k < 4; k++) av[k] = ar.scale_canonical(v[k], mj.ninv_d);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u64 tk = qsp > mj.q ? reduce64(tl[k], mj) : tl[k];
        tk = sub_mod(tk, ctx->qsp_half_mod_q[J], mj.q);
        u64 d = sub_mod(av[k], tk, mj.q);
        d = mul_shoup(d, ctx->inv_qsp_mod_q[J], mj.q);
        const u64 bv = add_mod(bw[k], ex[k], mj.q);  // the base ciphertext + a fused Add node's (absent: 0)
        st_tail_out<L>(buf_row(rout, (size_t)J * N), t, k, add_mod(bv, d, mj.q));
      }
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
  }
}

// =================================================================================================
// BEHZ multiply (2 x 2 -> 3), split pipeline.  Replaces behz_extend -> ntt_fwd -> tensor -> ntt_inv ->
// behz_floor_sk of the whole-polynomial path (SEAL bfv_multiply; Evaluator_Multiply,
// seal_fhe/src/evaluator_base.rs:198-212).  Residue r < K uses prime q_r (FP64 path when its range plan
// succeeded, integer path otherwise), residue K + j uses the auxiliary prime Bsk_j: the library's own FP64-pipe base
// when DevCtx::aux_f64 (AUXD instantiations, conversions in exact FP64), SEAL's 61-bit base on the integer path otherwise.
// =================================================================================================


// the first HL = log2(NC) forward stages on the NC values {t + k*N/NC}; native (lazy) representation out
template <class A, int NC>
__device__ __forceinline__ void head_fwd(const A& ar, typename A::V (&v)[NC], const typename A::Tw* __restrict__ tw) {
  constexpr int HL = NC == 8 ? 3 : NC == 4 ? 2 : 1;
  static_assert(NC == (1 << HL), "head threads own 2, 4 or 8 coefficients");
#pragma unroll
  for (int j = 0; j < HL; j++) {
    const int half = (NC / 2) >> j;
#pragma unroll
    for (int k = 0; k < NC; k++) {
      if (k & half) continue;
      ar.fwd(v[k], v[k + half], tw[(1u << j) + (u32)(k >> (HL - j))]);
    }
  }
}

// mul head: grid (N/8/256, 4 polys (a0,a1,b0,b1), ops); ext = [ops][4][K+S][N] in native representation
// AUXD (DevCtx::aux_f64): every residue, auxiliary base included, takes the FP64 policy and the base extension
// itself runs in FP64 (behz_extend_coeff_d).
// PACK (all-FP64 instantiations): 0 = 8-byte rows, 1 = every row 48-bit packed, 2 = per row (DevCtx::mul_row_packed for the data
// rows -- a wave-uniform branch -- and every auxiliary row packed)
// HGRID (all-FP64, K <= 4): the extension's sums in exact-sum form (behz_extend_multi_d<GRID>); its own instantiation -- with both forms
// in one kernel the register allocation is the union of the two (170 VGPRs, two waves per SIMD; apart: 106 and 122, four waves)
template <int L, int KMAX, bool AUXD, int PACK, bool HGRID = false>
__global__ __launch_bounds__(kHeadThreads, (!AUXD && PACK ? HEAD_MIXED_WAVES : 1)) void mul_head_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                const u64* __restrict__ in0, const u64* __restrict__ in1,
                                                                u64* __restrict__ ext, const MemberHead* __restrict__ members, u32 mfirst, u32 mper) {
  using G = EdgeGeom<L>;
  constexpr int NC = G::HEAD_NC;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 poly = blockIdx.y, op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  // src / dst: the polynomial's rows; owned coefficients sit at G::head_in(t, k) (inputs) / G::head_out(t, k) (outputs)
  const u64* src = (poly < 2 ? in0 + ((size_t)op * 2 + poly) * K * N : in1 + ((size_t)op * 2 + (poly - 2)) * K * N);
  if (members) {  // a merged launch of the graph executor: every member's operands where they are (kernels.hpp MemberHead; wave-uniform)
    const u32 it = mfirst + op, mem = it / mper;
    const MemberHead mh = members[mem];
    src = (poly < 2 ? mh.a : mh.b) + ((size_t)(it - mem * mper) * 2 + (poly & 1u)) * K * N;
  }
  u64* dst = ext + ((size_t)op * 4 + poly) * R * N;
  const BufRsrc rin = buf_rsrc(src), rout = buf_rsrc(dst);  // buffer addressing: see BufRow
  if constexpr (AUXD) {
    // Every input word is requested before the first is used, and no request sits behind a branch: rows i >= K (instantiation
    // wider than the context) re-read row K - 1 and are discarded.  (r01-r03 wrote `i < K ? load : 0`; the compiler gave every
    // load its own basic block and an s_waitcnt vmcnt(0) -- KMAX x NC dependent HBM round trips per thread, found in r04 by
    // listing the load / wait sequence of the ISA.)
    u64 raw[KMAX][NC];
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
      const u32 row = (u32)i < K ? (u32)i : K - 1;
#pragma unroll
      for (int k = 0; k < NC; k++) raw[i][k] = ld_head_in<L>(buf_row(rin, (size_t)row * N), t, k);
    }
    // (rows i >= K hold a copy of row K - 1 and are never read: every consumer below is guarded by i < K.  r01-r04 zeroed them with
    // a select per word -- 64 v_cndmask per thread at K = KMAX, where there is nothing to zero)
    double x[KMAX][NC];
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
#pragma unroll
      for (int k = 0; k < NC; k++) x[i][k] = ArithD::from_u64(raw[i][k]);
    }
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
      if ((u32)i < K) {
        const ArithD ar(ctx->mod[i]);
        double v[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] = x[i][k];
        head_fwd_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twf_base + (size_t)i * N), t);
        const BufRow o = buf_row(rout, (size_t)i * N);
        const bool packed = PACK == 1 || (PACK == 2 && ((ctx->mul_row_mask >> i) & 1u) != 0);
        if (packed) {
          if (ctx->mod[i].split_fwd_mask & kPlanStoreReduce) {  // only primes whose head outputs exceed the packed range (context.cpp)
#pragma unroll
            for (int k = 0; k < NC; k++) v[k] = ar.reduce(v[k]);
          }
          nat_store_head<L, true, NtSites<L>::head_st>(o, t, v);
        } else {
          nat_store_head<L, false, NtSites<L>::head_st>(o, t, v);
        }
      }
    }
    // auxiliary base: extend all eight owned coefficients residue by residue (every conversion constant is
    // fetched once), and run the head stages of each auxiliary residue as soon as it is complete
    behz_extend_multi_d<KMAX, NC, HGRID>(ctx, x, [&](u32 j, double(&ev)[NC]) {
      const ArithD ar(ctx->mod[KK + j]);
      head_fwd_owned<ArithD, L>(ar, ev, reinterpret_cast<const double*>(twf_base + (size_t)(KK + j) * N), t);
      const BufRow o = buf_row(rout, (size_t)(K + j) * N);
      if constexpr (PACK != 0) {
        if (ctx->mod[KK + j].split_fwd_mask & kPlanStoreReduce) {
#pragma unroll
          for (int k = 0; k < NC; k++) ev[k] = ar.reduce(ev[k]);
        }
      }
      nat_store_head<L, (PACK != 0), NtSites<L>::head_st>(o, t, ev);
    });
    return;
  }
  u64 x[KMAX][NC];
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    const u32 row = (u32)i < K ? (u32)i : K - 1;  // branch-free requests, all in flight together (see the AUXD branch)
#pragma unroll
    for (int k = 0; k < NC; k++) x[i][k] = ld_head_in<L>(buf_row(rin, (size_t)row * N), t, k);
  }
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
#pragma unroll
    for (int k = 0; k < NC; k++) x[i][k] = (u32)i < K ? x[i][k] : 0;
  }
  // q residues: just the three head stages
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    if ((u32)i < K) {
      const DevMod& dm = ctx->mod[i];
      const MulOp* tw = twf_base + (size_t)i * N;
      if (residue_is_f64(dm)) {
        const ArithD ar(dm);
        double v[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] = ar.from_u64(x[i][k]);
        head_fwd_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(tw), t);
#pragma unroll
        for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, (size_t)i * N), t, k, (u64)__double_as_longlong(v[k]));
      } else {
        const ArithI ar(dm);
        u64 v[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) v[k] = x[i][k];
        head_fwd_owned<ArithI, L>(ar, v, tw, t);
#pragma unroll
        for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, (size_t)i * N), t, k, v[k]);
      }
    }
  }
  // auxiliary base: extend every owned coefficient, then the head stages per Bsk prime
  if constexpr (!AUXD && PACK) {  // the MIXED instantiation (PACK has no meaning of its own without AUXD)
    // mixed base (context.cpp): integer data primes, auxiliary primes on the FP64 pipe; the extension sums run in exact FP64
    behz_extend_multi_mixed<KMAX, NC>(ctx, x, [&](u32 j, double(&ev)[NC]) {
      const ArithD ar(ctx->mod[KK + j]);
      head_fwd_owned<ArithD, L>(ar, ev, reinterpret_cast<const double*>(twf_base + (size_t)(KK + j) * N), t);
#pragma unroll
      for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, (size_t)(K + j) * N), t, k, (u64)__double_as_longlong(ev[k]));
    });
    return;
  }
  u64 ev[KMAX + 2][NC];
#pragma unroll
  for (int k = 0; k < NC; k++) {
    u64 xr[KMAX], er[KMAX + 2];
#pragma unroll
    for (int i = 0; i < KMAX; i++) xr[i] = x[i][k];
#pragma unroll
    for (int j = 0; j < KMAX + 2; j++) er[j] = 0;
    behz_extend_coeff<KMAX>(ctx, xr, er);
#pragma unroll
    for (int j = 0; j < KMAX + 2; j++) ev[j][k] = er[j];
  }
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) {
    if ((u32)j < S) {
      const DevMod& dm = ctx->mod[KK + j];
      const ArithI ar(dm);
      u64 v[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) v[k] = ev[j][k];
      head_fwd_owned<ArithI, L>(ar, v, twf_base + (size_t)(KK + j) * N, t);
#pragma unroll
      for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, (size_t)(K + j) * N), t, k, v[k]);
    }
  }
}

// mul middle body for one (op, residue, block): the four forward transforms (and then the three inverse ones) advance together, pass by
// pass (mid_forward_multi): smem = 4 regions of BLOCK elements, nothing is parked.
// MODE 0: the four forward transforms together (4 exchange regions); 1: two pairs, one polynomial of the waiting pair parked in
// a third region (3 regions, N = 8192); 2: two pairs with nothing parked and the inverse transforms as a pair + one (2 regions:
// the 16-elements-per-thread geometry of N = 16384, where two such workgroups share a CU)
// SQUARE: both operands are the same ciphertext (Evaluator_Square, a program's x * x): only its two polynomials were extended
// (ext polys 0, 1), two forward transforms instead of four, d = (a0^2, a0 a1 + a0 a1, a1^2) -- the sums SEAL's bfv_multiply
// forms for equal operands, so the bits are those of multiply(x, x)
template <class A, int L, bool PACK, int EPT = kBlkEPT, int MODE = ((MID_FWD_PAIRS(L) && std::is_same<A, ArithD>::value) ? 1 : 0), bool SQUARE = false>
__device__ __forceinline__ void mul_mid_body_batched(const DevMod& dm, const typename A::Tw* twf, const typename A::Tw* twi,
                                                     const typename A::V* ext_r, size_t poly_stride, typename A::V* D_r, size_t dpoly_stride,
                                                     typename A::V* smem, u32 tid, u32 blk) {
  using Sh = SplitShape<L, EPT>;
  const A ar(dm);
  // packed rows: reduce in front of the store only where the range plan says the last pass leaves more than a packed row holds
  const bool store_reduce = (dm.split_inv_mask & kPlanStoreReduce) != 0;
  constexpr int RF0 = split_fwd_radix(L, 0), LOWF0 = split_fwd_low(L, 0);
  using First = BlkPass<A, L, LOWF0, RF0, EPT>;
  constexpr int RI = split_inv_radix(L, Sh::NPI - 1), LOWI = split_inv_low(L, Sh::NPI - 1);
  using Out = BlkPass<A, L, LOWI, RI, EPT>;
  typename A::V v[4][EPT];
  const typename A::V* ext_p = ext_r;  // re-based (pin_loads) where a load must not be scheduled above the transform before it
  auto pin_loads = [&]() {
    u32 zero = 0;
    asm volatile("" : "+s"(zero) : : "memory");
    ext_p = ext_r + zero;  // an opaque offset keeps the pointer's address space (nttcore.hpp opaque_uniform)
  };
  auto load_poly = [&](int i) {
    const typename A::V* src = ext_p + (size_t)i * poly_stride;
#pragma unroll
    for (int g = 0; g < First::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RF0); k++) {
        if constexpr (PACK && std::is_same<A, ArithD>::value)
          v[i][g * (1 << RF0) + k] = nat_load<true, NtSites<L>::mul_mid_ld>(src, Sh::N, First::elem(tid, blk, g, k));
        else
          v[i][g * (1 << RF0) + k] = nt_ld<NtSites<L>::mul_mid_ld>(src + First::elem(tid, blk, g, k));
      }
  };
  // MODE 2 loads the second pair only after the first pair's transform: at 16 elements per thread the four operands are 128
  // registers, and holding all of them through the first transform spills (the other resident workgroup covers the latency)
  using Pair = typename A::V[2][EPT];
  using One = typename A::V[1][EPT];
  if constexpr (MODE == 3 && !SQUARE) {
    // 16 elements per thread, two 256-thread workgroups per CU (r03): the four operands never sit in registers together, and
    // each product polynomial leaves as soon as it is complete.  (a0, b0) are transformed as a pair, d0 = a0 b0 is taken
    // through its inverse transform and stored at once; b1 is transformed alone, d1 = a0 b1; a1 alone, d1 += a1 b0,
    // d2 = a1 b1; (d1, d2) are inverse-transformed as a pair.  Live at the worst point: two operands, one product, the
    // transform in flight and one set of twiddles -- 184 of the 256 registers.  Two exchange regions (64 KB).
    auto store_poly = [&](int i, typename A::V(&dv)[EPT]) {
      typename A::V* dst = D_r + (size_t)i * dpoly_stride;
      if constexpr (PACK && std::is_same<A, ArithD>::value) {
        if (store_reduce) reduce_all<A, EPT>(ar, dv);
      }
#pragma unroll
      for (int g = 0; g < Out::G; g++)
#pragma unroll
        for (int k = 0; k < (1 << RI); k++) {
          if constexpr (PACK && std::is_same<A, ArithD>::value)
            nat_store<true, NtSites<L>::mul_mid_st>(dst, Sh::N, Out::elem(tid, blk, g, k), dv[g * (1 << RI) + k]);
          else
            nt_st<NtSites<L>::mul_mid_st>(dst + Out::elem(tid, blk, g, k), dv[g * (1 << RI) + k]);
        }
    };
    auto fresh = [&](const typename A::Tw* tw) { return opaque_uniform(tw); };  // keep one transform's twiddle fetches from being merged with another's
    auto load_into = [&](int i, typename A::V(&dst)[EPT]) {
      const typename A::V* src = ext_p + (size_t)i * poly_stride;
#pragma unroll
      for (int g = 0; g < First::G; g++)
#pragma unroll
        for (int k = 0; k < (1 << RF0); k++) {
          if constexpr (PACK && std::is_same<A, ArithD>::value)
            dst[g * (1 << RF0) + k] = nat_load<true, NtSites<L>::mul_mid_ld>(src, Sh::N, First::elem(tid, blk, g, k));
          else
            dst[g * (1 << RF0) + k] = nt_ld<NtSites<L>::mul_mid_ld>(src + First::elem(tid, blk, g, k));
        }
    };
    typename A::V ab[2][EPT];  // (a0, b0); b1 later takes a0's place
    load_into(0, ab[0]);
    load_into(2, ab[1]);
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, ab, smem, tid, blk, twf, dm.split_fwd_mask);
    {
      typename A::V d0[1][EPT];
#pragma unroll
      for (int e = 0; e < EPT; e++) d0[0][e] = ar.mul_var(ab[0][e], ab[1][e]);
      __syncthreads();  // the pair's last forward pass may still be reading the exchange buffer
      mid_inverse_multi<A, L, 1, EPT, MID_TW_PIPE(L)>(ar, d0, smem, tid, blk, fresh(twi), dm.split_inv_mask);
      store_poly(0, d0[0]);
    }
    typename A::V x[1][EPT], d12[2][EPT];
    pin_loads();
    load_into(3, x[0]);
    __syncthreads();
    mid_forward_multi<A, L, 1, EPT, MID3_PIPE_SINGLE>(ar, x, smem, tid, blk, fresh(twf), dm.split_fwd_mask);  // b1
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      d12[0][e] = ar.mul_var(ab[0][e], x[0][e]);  // a0 b1
      ab[0][e] = x[0][e];                          // b1 takes a0's place
    }
    typename A::V* parked = smem + 2 * Sh::BLOCK + tid;  // thread-private slots behind the two exchange regions (MulMidGeom::PARK)
#pragma unroll
    for (int e = 0; e < MID3_PARK; e++) parked[e * Sh::TPB] = d12[0][e];
    pin_loads();
    load_into(1, x[0]);
    __syncthreads();
    mid_forward_multi<A, L, 1, EPT, MID3_PIPE_SINGLE>(ar, x, smem, tid, blk, fresh(twf), dm.split_fwd_mask);  // a1
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      const typename A::V a0b1 = e < MID3_PARK ? parked[e * Sh::TPB] : d12[0][e];
      d12[0][e] = ar.mul_add(x[0][e], ab[1][e], a0b1);  // + a1 b0
      d12[1][e] = ar.mul_var(x[0][e], ab[0][e]);        // a1 b1
    }
    __syncthreads();
    mid_inverse_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, d12, smem, tid, blk, fresh(twi), dm.split_inv_mask);
    store_poly(1, d12[0]);
    store_poly(2, d12[1]);
    return;
  }
  load_poly(0);
  load_poly(1);
  if constexpr (MODE != 2 && !SQUARE) {
    load_poly(2);
    load_poly(3);
  }
  if constexpr (SQUARE && MODE == 2) {
    // r06: the squaring body of the 16-element geometry sequenced like MODE 3 -- a0^2 leaves (inverse transform, store) before
    // 2 a0 a1 and a1^2 are formed, so the three products are never live beside both operands: the 48-bit packed instantiation
    // no longer spills (94 scratch instructions before: the reason per-row packing lost on chi_sq, HISTORY.md R5).  Same sums,
    // same transforms per polynomial: the bits of the batched form.
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[0]), smem, tid, blk, twf, dm.split_fwd_mask);
    auto store_one = [&](int i, typename A::V(&dv)[EPT]) {
      typename A::V* dst = D_r + (size_t)i * dpoly_stride;
      if constexpr (PACK && std::is_same<A, ArithD>::value) {
        if (store_reduce) reduce_all<A, EPT>(ar, dv);
      }
#pragma unroll
      for (int g = 0; g < Out::G; g++)
#pragma unroll
        for (int k = 0; k < (1 << RI); k++) {
          if constexpr (PACK && std::is_same<A, ArithD>::value)
            nat_store<true, NtSites<L>::mul_mid_st>(dst, Sh::N, Out::elem(tid, blk, g, k), dv[g * (1 << RI) + k]);
          else
            nt_st<NtSites<L>::mul_mid_st>(dst + Out::elem(tid, blk, g, k), dv[g * (1 << RI) + k]);
        }
    };
    {
      typename A::V d0[1][EPT];
#pragma unroll
      for (int e = 0; e < EPT; e++) d0[0][e] = ar.mul_var(v[0][e], v[0][e]);
      __syncthreads();  // the pair's last forward pass may still be reading the exchange buffer
      mid_inverse_multi<A, L, 1, EPT, MID_TW_PIPE(L)>(ar, d0, smem, tid, blk, opaque_uniform(twi), dm.split_inv_mask);
      store_one(0, d0[0]);
    }
    typename A::V d12[2][EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      d12[0][e] = ar.mul_add(v[0][e], v[1][e], ar.mul_var(v[1][e], v[0][e]));
      d12[1][e] = ar.mul_var(v[1][e], v[1][e]);
    }
    __syncthreads();
    mid_inverse_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, d12, smem, tid, blk, opaque_uniform(twi), dm.split_inv_mask);
    store_one(1, d12[0]);
    store_one(2, d12[1]);
    return;
  }
  if constexpr (SQUARE) {
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[0]), smem, tid, blk, twf, dm.split_fwd_mask);
  } else if constexpr (MODE == 1) {
    // two pairs through 2 (of the 3) exchange regions: 48 KB of LDS per workgroup instead of 64 KB -> 3 workgroups per CU
    // The pair that is not being transformed would sit in 32 registers; one of its two polynomials waits in the third
    // exchange region instead (each thread parks and fetches its own values: no synchronisation), which keeps the kernel
    // under the 168 registers of 3 waves per SIMD without scratch spills.
    typename A::V* park = smem + 2 * Sh::BLOCK + tid;
#pragma unroll
    for (int e = 0; e < EPT; e++) park[e * Sh::TPB] = v[3][e];
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[0]), smem, tid, blk, twf, dm.split_fwd_mask);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; e++) {
      v[3][e] = park[e * Sh::TPB];
      park[e * Sh::TPB] = v[0][e];
    }
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[2]), smem, tid, blk, twf, dm.split_fwd_mask);
#pragma unroll
    for (int e = 0; e < EPT; e++) v[0][e] = park[e * Sh::TPB];
  } else if constexpr (MODE == 2) {
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[0]), smem, tid, blk, twf, dm.split_fwd_mask);
    load_poly(2);
    load_poly(3);
    __syncthreads();  // the first pair's last pass may still be reading the exchange buffer
    mid_forward_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&v[2]), smem, tid, blk, twf, dm.split_fwd_mask);
  } else {
    mid_forward_multi<A, L, 4, EPT, MID_TW_PIPE(L)>(ar, v, smem, tid, blk, twf, dm.split_fwd_mask);
  }
  typename A::V d[3][EPT];
#pragma unroll
  for (int e = 0; e < EPT; e++) {
    if constexpr (SQUARE) {
      d[0][e] = ar.mul_var(v[0][e], v[0][e]);
      d[1][e] = ar.mul_add(v[0][e], v[1][e], ar.mul_var(v[1][e], v[0][e]));
      d[2][e] = ar.mul_var(v[1][e], v[1][e]);
    } else {
      d[0][e] = ar.mul_var(v[0][e], v[2][e]);
      d[1][e] = ar.mul_add(v[0][e], v[3][e], ar.mul_var(v[1][e], v[2][e]));
      d[2][e] = ar.mul_var(v[1][e], v[3][e]);
    }
  }
  __syncthreads();  // the last forward pass may still be reading the exchange buffer
  if constexpr (MODE == 2 || MODE == 3) {
    mid_inverse_multi<A, L, 2, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<Pair*>(&d[0]), smem, tid, blk, twi, dm.split_inv_mask);
    __syncthreads();
    mid_inverse_multi<A, L, 1, EPT, MID_TW_PIPE(L)>(ar, *reinterpret_cast<One*>(&d[2]), smem, tid, blk, twi, dm.split_inv_mask);
  } else {
    mid_inverse_multi<A, L, 3, EPT, MID_TW_PIPE(L)>(ar, d, smem, tid, blk, twi, dm.split_inv_mask);
  }
  if constexpr (PACK && std::is_same<A, ArithD>::value) {
    if (store_reduce) {
#pragma unroll
      for (int i = 0; i < 3; i++) reduce_all<A, EPT>(ar, d[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; i++) {
    typename A::V* dst = D_r + (size_t)i * dpoly_stride;
#pragma unroll
    for (int g = 0; g < Out::G; g++)
#pragma unroll
      for (int k = 0; k < (1 << RI); k++) {
        if constexpr (PACK && std::is_same<A, ArithD>::value)
          nat_store<true, NtSites<L>::mul_mid_st>(dst, Sh::N, Out::elem(tid, blk, g, k), d[i][g * (1 << RI) + k]);
        else
          nt_st<NtSites<L>::mul_mid_st>(dst + Out::elem(tid, blk, g, k), d[i][g * (1 << RI) + k]);
      }
  }
}

// grid: ops * nres * NBLK workgroups of TPB threads; D = [ops][3][R][N] native representation.
// Two instantiations (separate register allocations): FP64 residues (r in [0, K)) and integer residues.
// POLICY_D selects which residues this launch handles: r0 = first residue, nres = number of residues.
// Elements per thread of the FP64 middle kernels at N = 16384.  16 (256-thread workgroups, two exchange regions, two workgroups
// per CU instead of one) was built and measured in round 2: the four operands of the tensor product are 128 registers, the 14
// vector twiddles of a pass and of the prefetched next pass another 112, the kernel spills 860 bytes per lane and mul_mid takes
// 16.5 ms instead of 6.4 (bit-exact).  Round 3 built two ways round the four operands (see below); with 16-byte
// twiddles both lost to the 8-element kernel (8.7 and 7.1 ms against 6.5), with 8-byte twiddles (nttcore.hpp) MODE 3 fits 256
// registers but for 92 bytes of scratch and wins: mul_mid 6.10-6.22 against 6.27-6.57 ms per 1024 ops (interleaved A/B on two
// boxes, mul+relin +0.3...2.4 %).  16 is the default; `-DMID_EPT_14=8` is the one-workgroup-per-CU kernel of rounds 1-2.
#define MID_EPT_14 (HIPBFV_GEOM14 == 4 ? 16 : 8)
// The SQUARING instantiation at N = 16384 does take 16 (r03): two transforms instead of four fit the 256 registers of two
// 256-thread workgroups per CU without scratch (MID_EPT_14_SQ).
#define MID_EPT_14_SQ (HIPBFV_GEOM14 == 4 ? 16 : 8)
// The general (four-operand) body at 16 elements per thread is MODE 3 of mul_mid_body_batched: operands folded into the products
// one at a time (three forward rounds).  (r03 also built a form that parked the first pair's transforms in place in `ext` and read
// them back from L2 at the tensor product: scratch-free, two rounds, and slower -- 6.7 against 6.1 ms; removed in r04.)
// Workgroup order of mul_mid: 1 = slice-major per XCD (see the kernel).  At N = 16384 the 36 vector-twiddle tables of a multiply
// (18 moduli, forward and inverse, 128 KB each) are 4.7 MB -- more than one XCD's 4 MB L2, which also has the intermediates
// streaming through it; r03's PMC passes showed mul_mid<14> fetching 1.3 MB per op beyond its rows and its scratch.
#define MUL_MID_SLICE(L) ((L) == 14)
constexpr int mid_ept_d(int logn, bool square) {
  return logn == 14 ? (square ? MID_EPT_14_SQ : MID_EPT_14) : kBlkEPT;
}
template <int L, bool POLICY_D, bool SQUARE = false, bool PACK = false>
struct MulMidGeom {
  static constexpr int EPT = POLICY_D ? mid_ept_d(L, SQUARE) : kBlkEPT;
  static constexpr int MODE = !POLICY_D ? 0 : EPT > kBlkEPT ? (SQUARE ? 2 : 3) : MID_FWD_PAIRS(L) ? 1 : 0;
  static constexpr int REGIONS = MODE >= 2 ? 2 : MODE == 1 ? 3 : 4;
  static constexpr int TPB = SplitShape<L, EPT>::TPB;
  static constexpr int PARK = (POLICY_D && MODE == 3 && !SQUARE) ? MID3_PARK * TPB : 0;  // LDS words behind the exchange regions
  static constexpr int WAVES = !POLICY_D ? MID_WAVES_I : EPT > kBlkEPT ? 2 : MID_WAVES_D(L);
};
template <int L, bool POLICY_D, bool PACK, bool SQUARE = false>
__global__ __launch_bounds__((MulMidGeom<L, POLICY_D, SQUARE, PACK>::TPB), (MulMidGeom<L, POLICY_D, SQUARE, PACK>::WAVES)) void mul_mid_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                           const MulOp* __restrict__ twi_base, const u64* __restrict__ ext,
                                                                           u64* __restrict__ D, const unsigned char* __restrict__ residues, u32 nres, u32 ops) {
  using Geo = MulMidGeom<L, POLICY_D, SQUARE, PACK>;
  using Sh = SplitShape<L, Geo::EPT>;
  __shared__ u64 smem[Geo::REGIONS * Sh::BLOCK + Geo::PARK];
  const u32 tid = threadIdx.x;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  const u32 b = blockIdx.x;
  u32 blk, r, op;
  if constexpr (MUL_MID_SLICE(L)) {
    // slice-major per XCD (the order ks_mid uses for its key rows): workgroups are dealt round-robin to the 8 XCDs, and within
    // one XCD consecutive workgroups walk the ops of ONE (residue, block) slice, so the 2 x 32 KB of vector twiddles that slice
    // uses are fetched into that XCD's L2 once per slice instead of once per workgroup
    const u32 xcd = b & 7u, slot = b >> 3;
    const u32 per = (ops + 7u) >> 3;
    op = (slot % per) * 8u + xcd;
    const u32 ib = slot / per;
    blk = ib % Sh::NBLK;
    r = residue_of(residues, ib / Sh::NBLK);
    if (op >= ops) return;
  } else {
    blk = b % Sh::NBLK;
    r = residue_of(residues, (b / Sh::NBLK) % nres);
    op = b / (Sh::NBLK * nres);
  }
  const u32 m = r < K ? r : KK + (r - K);
  const DevMod& dm = ctx->mod[m];
  const u64* ext_r = ext + ((size_t)op * 4 * R + r) * Sh::N;
  u64* D_r = D + ((size_t)op * 3 * R + r) * Sh::N;
  const size_t ps = (size_t)R * Sh::N;
  const MulOp* twf = twf_base + (size_t)m * Sh::N;
  const MulOp* twi = twi_base + (size_t)m * Sh::N;
  if constexpr (POLICY_D)
    mul_mid_body_batched<ArithD, L, PACK, Geo::EPT, Geo::MODE, SQUARE>(dm, reinterpret_cast<const double*>(twf), reinterpret_cast<const double*>(twi),
                                    reinterpret_cast<const double*>(ext_r), ps, reinterpret_cast<double*>(D_r), ps,
                                    reinterpret_cast<double*>(smem), tid, blk);
  else
    mul_mid_body_batched<ArithI, L, false, kBlkEPT, 0, SQUARE>(dm, twf, twi, ext_r, ps, D_r, ps, smem, tid, blk);
}

// the tail's inverse stages + BEHZ scaling on the 4 coefficients thread t owns (EdgeGeom): canonical residues out
// src: the residue row (not offset by t)
template <class A, int L>
__device__ __forceinline__ void tail_inv4_scale(const A& ar, const typename A::V* __restrict__ src, u32 t, const typename A::Tw* __restrict__ tw,
                                                const typename A::Sc& sc, u32 mask, u64 (&out)[4]) {
  typename A::V v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = src[EdgeGeom<L>::tail_in(t, k)];
  tail_inv_owned<A, L>(ar, v, tw, mask, t);
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = ar.scale_canonical(v[k], sc);
}

// the same from four words requested earlier (8-byte rows of either policy: doubles or lazy u64)
template <class A, int L>
__device__ __forceinline__ void tail_inv4_scale_raw(const A& ar, const u64 (&raw)[4], u32 t, const typename A::Tw* __restrict__ tw,
                                                    const typename A::Sc& sc, u32 mask, u64 (&out)[4]) {
  typename A::V v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if constexpr (std::is_same<typename A::V, double>::value) v[k] = __longlong_as_double((long long)raw[k]);
    else v[k] = raw[k];
  }
  tail_inv_owned<A, L>(ar, v, tw, mask, t);
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = ar.scale_canonical(v[k], sc);
}

// the same for the FP64 epilogue: doubles out with |out| < q -- the scaling product as ArithD::mul_const leaves it, q * (0.5 + |v| * 2^-52),
// which the range plan keeps below 0.95 q for every prime without kPlanScaleReduce (context.cpp); reduced first otherwise
template <int L, bool PACK>
__device__ __forceinline__ void tail_inv4_scale_d(const ArithD& ar, const NatRaw<PACK> (&raw)[4], const double* __restrict__ tw, const MulOpD& sc,
                                                  u32 mask, u32 t, double (&out)[4]) {
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACK>(raw[k]);
  tail_inv_owned<ArithD, L>(ar, v, tw, mask, t);
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = ar.mul_const(v[k], sc);
  if (mask & kPlanScaleReduce) {
    HIPBFV_KEEP_BRANCH();
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = ar.reduce(out[k]);
  }
}

// mul tail: grid (N/4/256, 3 polys, ops); out = [ops][3][K][N] canonical
// GRID (DevCtx::conv_grid, 8-prime all-FP64 instantiation only): floor sums formed exactly and reduced once (griddot.hpp)
// All-FP64 tail of the BEHZ multiply for the 4 coefficients {t + k*N/4} of ONE output polynomial: last two inverse stages of
// every residue, scaling, fast_floor + Shenoy-Kumaresan conversion (behz_floor_sk_multi_d): canonical data residues in res.
// d: the polynomial's R residue rows in D (not offset by t); thread t owns the coefficients EdgeGeom<L>::tail_out(t, k).
// PACK: 0 / 1 / 2 as in mul_head_kernel (2: the data rows' representation is DevCtx::mul_row_packed, the auxiliary rows are packed)
// OutT: u64 words for mul_tail_kernel to store, doubles (the same canonical integers) for the fused kernels that compute on
template <int L, int KMAX, int PACK, bool GRID, class OutT>
__device__ __forceinline__ void mul_tail_compute_d(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base, const u64* __restrict__ d, u32 t,
                                                   OutT (&res)[KMAX][4]) {
  using G = EdgeGeom<L>;
  constexpr u32 N = 1u << L;
  constexpr bool PD = PACK == 1;  // data rows, when the choice is static
  constexpr bool PA = PACK != 0;  // auxiliary rows
  const u32 K = ctx->K, KK = ctx->KK;
  const BufRsrc rd = buf_rsrc(d);  // buffer addressing: see BufRow
  double yc[KMAX][4];
  if constexpr (PACK != 2) {
    // the words of data row i + 1 are requested before row i is transformed; the requests are branch-free (rows beyond K re-read
    // row K - 1 and are dropped), so they are not held behind the `i < K` test (r04: see mul_head_kernel)
    NatRaw<PD> cur[4], nxt[4];
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nat_fetch_tail<L, PD, NtSites<L>::tail_ld>(buf_row(rd, 0), t, k);
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
      if (i + 1 < KMAX) {
        const BufRow next_row = buf_row(rd, (size_t)((u32)(i + 1) < K ? (u32)(i + 1) : K - 1) * N);
#pragma unroll
        for (int k = 0; k < 4; k++) nxt[k] = nat_fetch_tail<L, PD, NtSites<L>::tail_ld>(next_row, t, k);
      }
      if ((u32)i < K) {
        const DevMod& dm = ctx->mod[i];
        const ArithD ar(dm);
        double r4[4];
        tail_inv4_scale_d<L, PD>(ar, cur, reinterpret_cast<const double*>(twi_base + (size_t)i * N), ctx->intt_scale_q_d[i], dm.split_inv_mask, t, r4);
#pragma unroll
        for (int k = 0; k < 4; k++) yc[i][k] = r4[k] < 0.0 ? r4[k] + ar.q : r4[k];  // canonical: |r4| < q
      }
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
  } else {
    // per-row packing (r06: the default at N = 16384): the same look-ahead as above -- row i + 1 requested before row i is
    // transformed -- on a raw form that serves both representations: two dwords per value, the packed row's (u32, biased u16)
    // or the 8-byte row's (low, high) halves; which one a row is, is wave-uniform (DevCtx::mul_row_mask).  r05 had no
    // look-ahead in this arm: every row loaded, waited and computed in turn, and the packed tail ran 8 % slower than the 8-byte one.
    struct RawU {
      u32 lo, hi;
    };
    const u32 rmask = ctx->mul_row_mask;
    auto fetch_u = [&](u32 row, RawU(&r)[4]) {
      const BufRow br = buf_row(rd, (size_t)row * N);
      if ((rmask >> row) & 1u) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const NatRaw<true> x = nat_fetch_tail<L, true, NtSites<L>::tail_ld>(br, t, k);
          r[k].lo = x.lo, r[k].hi = x.hi;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const NatRaw<false> x = nat_fetch_tail<L, false, NtSites<L>::tail_ld>(br, t, k);
          r[k].lo = (u32)__double2loint(x.d), r[k].hi = (u32)__double2hiint(x.d);
        }
      }
    };
    RawU cur[4], nxt[4];
    fetch_u(0u, cur);
#pragma unroll
    for (int i = 0; i < KMAX; i++) {
      if (i + 1 < KMAX) fetch_u((u32)(i + 1) < K ? (u32)(i + 1) : K - 1, nxt);
      if ((u32)i < K) {
        const DevMod& dm = ctx->mod[i];
        const ArithD ar(dm);
        double r4[4];
        if (((rmask >> i) & 1u) != 0) {  // wave-uniform
          NatRaw<true> raw[4];
#pragma unroll
          for (int k = 0; k < 4; k++) raw[k].lo = cur[k].lo, raw[k].hi = cur[k].hi;
          tail_inv4_scale_d<L, true>(ar, raw, reinterpret_cast<const double*>(twi_base + (size_t)i * N), ctx->intt_scale_q_d[i], dm.split_inv_mask, t, r4);
        } else {
          NatRaw<false> raw[4];
#pragma unroll
          for (int k = 0; k < 4; k++) raw[k].d = __hiloint2double((int)cur[k].hi, (int)cur[k].lo);
          tail_inv4_scale_d<L, false>(ar, raw, reinterpret_cast<const double*>(twi_base + (size_t)i * N), ctx->intt_scale_q_d[i], dm.split_inv_mask, t, r4);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) yc[i][k] = r4[k] < 0.0 ? r4[k] + ar.q : r4[k];  // canonical: |r4| < q
      }
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
  }
  behz_floor_sk_multi_d<KMAX, 4, GRID, NatRaw<PA>>(
      ctx, yc,
      [&](u32 j, NatRaw<PA>(&raw)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) raw[k] = nat_fetch_tail<L, PA, NtSites<L>::tail_ld>(buf_row(rd, (size_t)(K + j) * N), t, k);
      },
      [&](u32 j, const NatRaw<PA>(&raw)[4], double(&xb)[4]) {
        const DevMod& dm = ctx->mod[KK + j];
        tail_inv4_scale_d<L, PA>(ArithD(dm), raw, reinterpret_cast<const double*>(twi_base + (size_t)(KK + j) * N), ctx->intt_scale_bsk_d[j],
                                   dm.split_inv_mask, t, xb);
      },
      res);
}

// The MIXED tail (integer-policy data primes, the library's FP64 auxiliary base: the 3 x 54-bit set) for the 4 coefficients
// {t + k*N/4} of ONE output polynomial, results in registers: res[k][i] = canonical residue i of coefficient slot k.  What
// mul_tail_kernel's mixed instantiation stores, and what the fused mixed kernels below compute on [r06].
template <int L, int KMAX>
__device__ __forceinline__ void mul_tail_compute_mixed(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base, const u64* __restrict__ d, u32 t,
                                                       u64 (&res)[4][KMAX]) {
  constexpr u32 N = 1u << L;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK;
  u64 y[4][KMAX], xb[4][KMAX + 2];
  const u32 last_row = K + S - 1;
  const BufRsrc rd = buf_rsrc(d);  // buffer addressing: see BufRow
  u64 cur[4], nxt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) cur[k] = ld_tail_in<L>(buf_row(rd, 0), t, k);
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    {
      const u32 nr = (u32)i + 1 < K ? (u32)i + 1 : K;  // after the last data row: the first auxiliary row
      const BufRow next_row = buf_row(rd, (size_t)(nr < last_row ? nr : last_row) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(next_row, t, k);
    }
    if ((u32)i < K) {
      const DevMod& dm = ctx->mod[i];
      u64 r4[4];
      if (residue_is_f64(dm)) {
        const ArithD ar(dm);
        tail_inv4_scale_raw<ArithD, L>(ar, cur, t, reinterpret_cast<const double*>(twi_base + (size_t)i * N), ctx->intt_scale_q_d[i], dm.split_inv_mask, r4);
      } else {
        const ArithI ar(dm);
        tail_inv4_scale_raw<ArithI, L>(ar, cur, t, twi_base + (size_t)i * N, ctx->intt_scale_q[i], 0u, r4);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) y[k][i] = r4[k];
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];  // (rows i >= K keep `cur`: it already holds the first auxiliary row)
    }
  }
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) {
    {
      const u32 nr = K + (u32)j + 1;
      const BufRow next_row = buf_row(rd, (size_t)(nr < last_row ? nr : last_row) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(next_row, t, k);
    }
    if ((u32)j < S) {
      const DevMod& dm = ctx->mod[KK + j];
      u64 r4[4];
      const ArithD ar(dm);  // the auxiliary rows come back from the FP64 middle kernel as doubles
      tail_inv4_scale_raw<ArithD, L>(ar, cur, t, reinterpret_cast<const double*>(twi_base + (size_t)(KK + j) * N), ctx->intt_scale_bsk_d[j], dm.split_inv_mask, r4);
#pragma unroll
      for (int k = 0; k < 4; k++) xb[k][j] = r4[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
  behz_floor_sk_coeff_mixed<KMAX, 4>(ctx, y, xb, res);
}

template <int L, int KMAX, bool AUXD, int PACK, bool GRID>
// poly0 / out_polys: the launch covers product polynomials poly0 .. poly0 + gridDim.y - 1 and writes them to
// out[op][out_polys][K][N] (3 polynomials from 0 for a stand-alone multiply; only c2, compactly, in the fused
// multiply + relinearize, whose last kernel forms c0 and c1 itself: mulrelin_tail_kernel)
__global__ EDGE_BOUNDS(KMAX) void mul_tail_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                const u64* __restrict__ D, u64* __restrict__ out, u32 poly0, u32 out_polys) {
  using G = EdgeGeom<L>;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 poly = blockIdx.y + poly0, op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  const u64* d = D + ((size_t)op * 3 + poly) * R * N;
  u64* o = out + ((size_t)op * out_polys + (poly - poly0)) * K * N;
  if constexpr (AUXD) {
    u64 res[KMAX][4];
    mul_tail_compute_d<L, KMAX, PACK, GRID>(ctx, twi_base, d, t, res);
    const BufRsrc rres = buf_rsrc(o);
#pragma unroll
    for (int i = 0; i < KMAX; i++)
      if ((u32)i < K) {
#pragma unroll
        for (int k = 0; k < 4; k++) st_tail_out<L>(buf_row(rres, (size_t)i * N), t, k, res[i][k]);
      }
    return;
  }
  constexpr bool mixed = !AUXD && PACK;  // the MIXED instantiation: integer data primes, FP64 auxiliary primes, Bsk-side sums in exact FP64
  if constexpr (mixed && TAIL_MIXED_NC == 4) {
    u64 res[4][KMAX];
    mul_tail_compute_mixed<L, KMAX>(ctx, twi_base, d, t, res);
    const BufRsrc rres = buf_rsrc(o);
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int i = 0; i < KMAX; i++)
        if ((u32)i < K) st_tail_out<L>(buf_row(rres, (size_t)i * N), t, k, res[k][i]);
    return;
  }
  u64 y[4][KMAX], xb[4][KMAX + 2];
  // The K + S rows of D are visited in order; the four words of the NEXT row are requested (branch-free: beyond the last row the
  // last row is re-read) before the current row is transformed (r04: each row used to load, wait and compute in its own block).
  const u32 last_row = K + S - 1;
  const BufRsrc rd = buf_rsrc(d), ro = buf_rsrc(o);  // buffer addressing: see BufRow
  u64 cur[4], nxt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) cur[k] = ld_tail_in<L>(buf_row(rd, 0), t, k);
#pragma unroll
  for (int i = 0; i < KMAX; i++) {
    {
      const u32 nr = (u32)i + 1 < K ? (u32)i + 1 : K;  // after the last data row: the first auxiliary row
      const BufRow next_row = buf_row(rd, (size_t)(nr < last_row ? nr : last_row) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(next_row, t, k);
    }
    if ((u32)i < K) {
      const DevMod& dm = ctx->mod[i];
      u64 r4[4];
      if (residue_is_f64(dm)) {
        const ArithD ar(dm);
        tail_inv4_scale_raw<ArithD, L>(ar, cur, t, reinterpret_cast<const double*>(twi_base + (size_t)i * N), ctx->intt_scale_q_d[i], dm.split_inv_mask, r4);
      } else {
        const ArithI ar(dm);
        tail_inv4_scale_raw<ArithI, L>(ar, cur, t, twi_base + (size_t)i * N, ctx->intt_scale_q[i], 0u, r4);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) y[k][i] = r4[k];
#pragma unroll
      for (int k = 0; k < 4; k++) cur[k] = nxt[k];  // (rows i >= K keep `cur`: it already holds the first auxiliary row)
    }
  }
#pragma unroll
  for (int j = 0; j < KMAX + 2; j++) {
    {
      const u32 nr = K + (u32)j + 1;
      const BufRow next_row = buf_row(rd, (size_t)(nr < last_row ? nr : last_row) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(next_row, t, k);
    }
    if ((u32)j < S) {
      const DevMod& dm = ctx->mod[KK + j];
      u64 r4[4];
      if constexpr (mixed) {  // the auxiliary rows come back from the FP64 middle kernel as doubles
        const ArithD ar(dm);
        tail_inv4_scale_raw<ArithD, L>(ar, cur, t, reinterpret_cast<const double*>(twi_base + (size_t)(KK + j) * N), ctx->intt_scale_bsk_d[j], dm.split_inv_mask, r4);
      } else {
        const ArithI ar(dm);
        tail_inv4_scale_raw<ArithI, L>(ar, cur, t, twi_base + (size_t)(KK + j) * N, ctx->intt_scale_bsk[j], 0u, r4);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) xb[k][j] = r4[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
  // The per-coefficient epilogue is too large to unroll four times; a rolled loop must not index y/xb by k
  // (dynamic indexing puts them in scratch), so each trip consumes row 0 and the rows rotate down.
  if constexpr (mixed) {
    // the mixed epilogue takes TAIL_MIXED_NC coefficients per trip (independent chains through the FP64 sums)
    constexpr int NCM = TAIL_MIXED_NC;
    static_assert(NCM == 1 || NCM == 2 || NCM == 4, "coefficients per trip");
#pragma unroll 1
    for (int k = 0; k < 4; k += NCM) {
      u64 yy[NCM][KMAX], xx[NCM][KMAX + 2], r[NCM][KMAX];
#pragma unroll
      for (int c = 0; c < NCM; c++) {
#pragma unroll
        for (int i = 0; i < KMAX; i++) yy[c][i] = y[c][i];
#pragma unroll
        for (int j = 0; j < KMAX + 2; j++) xx[c][j] = xb[c][j];
      }
      behz_floor_sk_coeff_mixed<KMAX, NCM>(ctx, yy, xx, r);
#pragma unroll
      for (int c = 0; c < NCM; c++)
#pragma unroll
        for (int i = 0; i < KMAX; i++)
          if ((u32)i < K) st_tail_out<L>(buf_row(ro, (size_t)i * N), t, k + c, r[c][i]);
#pragma unroll
      for (int kk = 0; kk + NCM < 4; kk++) {
#pragma unroll
        for (int i = 0; i < KMAX; i++) y[kk][i] = y[kk + NCM][i];
#pragma unroll
        for (int j = 0; j < KMAX + 2; j++) xb[kk][j] = xb[kk + NCM][j];
      }
    }
    return;
  }
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    u64 r[KMAX];
    behz_floor_sk_coeff<KMAX>(ctx, y[0], xb[0], r);
#pragma unroll
    for (int i = 0; i < KMAX; i++)
      if ((u32)i < K) st_tail_out<L>(buf_row(ro, (size_t)i * N), t, k, r[i]);
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
#pragma unroll
      for (int i = 0; i < KMAX; i++) y[kk][i] = y[kk + 1][i];
#pragma unroll
      for (int j = 0; j < KMAX + 2; j++) xb[kk][j] = xb[kk + 1][j];
    }
  }
}

// -------------------------------------------------------------------------------------------------
// multiply + relinearize, last kernel: ks_tail whose base ciphertext (c0, c1 of the product) is computed in place from the
// multiply's intermediates D instead of being read back: c0 and c1 never travel through HBM (1 MB of the 12 MB a mul+relin
// moves at N = 8192).  Thread t of polynomial c owns the same coefficients {t + k*N/4} in both halves.  All-FP64 contexts
// only (the SEAL default parameter sets).  grid: (N/4/256, 2, ops)
// -------------------------------------------------------------------------------------------------
template <int L, int KMAX, int PACKM, bool GRID, bool PACKK>
__global__ EDGE_BOUNDS(KMAX) void mulrelin_tail_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                     const u64* __restrict__ D, const double* __restrict__ ACC,
                                                                     const u64* __restrict__ extra, u64* __restrict__ out,
                                                                     const MemberTail* __restrict__ members, u32 mfirst, u32 mper) {
  using G = EdgeGeom<L>;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 c = blockIdx.y, op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  double basev[KMAX][4];  // c0 / c1 of the product: canonical integers, as doubles (the mod-down adds them in FP64)
  mul_tail_compute_d<L, KMAX, PACKM, GRID>(ctx, twi_base, D + ((size_t)op * 3 + c) * R * N, t, basev);
  const double* acc = ACC + ((size_t)op * 2 + c) * KK * N;
  // a member table (kernels.hpp MemberTail; the graph executor's merged launches): this item's own output buffer, multiplier and signed
  // addend -- wave-uniform, scalar loads
  const u64* ex_base = extra + ((size_t)op * 2 + c) * K * N;
  u64* out_base = out + ((size_t)op * 2 + c) * K * N;
  bool has_ex = extra != nullptr;
  double mmult = 1.0, msign = 1.0;
  if (members) {
    const u32 it = mfirst + op, mem = it / mper;
    const size_t at = ((size_t)(it - mem * mper) * 2 + c) * K * N;
    const MemberTail mt = members[mem];
    out_base = mt.out + at;
    ex_base = mt.extra + at;
    has_ex = mt.sign != 0;
    mmult = (double)mt.mult;
    msign = (double)mt.sign;
  }
  // buffer addressing (BufRow): the accumulator rows, the optional addend and the output
  const BufRsrc racc = buf_rsrc(acc), rout = buf_rsrc(out_base);
  const BufRsrc rex = buf_rsrc_opt(ex_base, has_ex);
  double tld[4];
  {
    const DevMod& sp = ctx->mod[KK - 1];
    double v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACKK>(nat_fetch_tail<L, PACKK, NtSites<L>::tail_ld>(buf_row(racc, (size_t)(KK - 1) * N), t, k));
    tail_special_d<L>(ctx, sp, v, reinterpret_cast<const double*>(twi_base + (size_t)(KK - 1) * N), t, tld);
  }
  const u64 qsp = ctx->mod[KK - 1].q;
  // accumulator row J + 1 is requested (branch-free: rows beyond K re-read row K - 1) before row J is finished
  NatRaw<PACKK> cur[4], nxt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) cur[k] = nat_fetch_tail<L, PACKK, NtSites<L>::tail_ld>(buf_row(racc, 0), t, k);
#pragma unroll
  for (int J = 0; J < KMAX; J++) {
    if (J + 1 < KMAX) {
      const BufRow next_row = buf_row(racc, (size_t)((u32)(J + 1) < K ? (u32)(J + 1) : K - 1) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = nat_fetch_tail<L, PACKK, NtSites<L>::tail_ld>(next_row, t, k);
    }
    if ((u32)J < K) {
      const DevMod& mj = ctx->mod[J];
      const ArithD ar(mj);
      const double* tw = reinterpret_cast<const double*>(twi_base + (size_t)J * N);
      // a ciphertext added to the result (fused Add node): its four words, requested UNCONDITIONALLY (absent: a descriptor of zero
      // records, the loads return 0 without a memory access) -- a load under `if (extra)` ends its block in an s_waitcnt vmcnt(0)
      // that would also wait for the row just requested
      u64 ex[4];
#pragma unroll
      for (int k = 0; k < 4; k++) ex[k] = ld_tail_out<L>(buf_row(rex, (size_t)J * N), t, k);
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = nat_unpack<PACKK>(cur[k]);
      tail_inv_owned<ArithD, L>(ar, v, tw, mj.split_inv_mask, t);
      const MulOpD iw = ctx->inv_qsp_mod_q_d[J];
      const double hf = ctx->qsp_half_mod_q_d[J];
      const bool p_above_q = qsp > mj.q;
      double s[4];
#pragma unroll
      for (int k = 0; k < 4; k++) s[k] = ar.mul_const(v[k], mj.ninv_d);
      if (mj.split_inv_mask & kPlanScaleReduce) {  // mod_down_d wants |s| <= 2q
        HIPBFV_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 4; k++) s[k] = ar.reduce(s[k]);
      }
      if (members) {
        // mult * (c_J of the relinearised product) + sign * addend: |.| <= 4q + q < 2^53, an exact integer, reduced once more
        HIPBFV_KEEP_BRANCH();
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const double r0 = mod_down_d(ar.q, ar.qinv, iw.w, iw.wq, hf, p_above_q, s[k], tld[k], basev[J][k]);
          const double r1 = ar.reduce(fma(mmult, r0, msign * ArithD::from_u64(ex[k])));
          st_tail_out<L>(buf_row(rout, (size_t)J * N), t, k, ArithD::to_bits(r1 < 0.0 ? r1 + ar.q : r1));
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const double bd = basev[J][k] + ArithD::from_u64(ex[k]);  // (an absent addend reads as 0)
          const double r = mod_down_d(ar.q, ar.qinv, iw.w, iw.wq, hf, p_above_q, s[k], tld[k], bd);  // moddown_d.hpp
          st_tail_out<L>(buf_row(rout, (size_t)J * N), t, k, ArithD::to_bits(r));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
}

// -------------------------------------------------------------------------------------------------
// multiply + relinearize, fourth kernel: the tail of the multiply for c2 (the key-switch target) and the head of the key
// switch in one pass -- c2 never travels through HBM either.  A key-switch head thread owns the NC = 2^head_log coefficients
// {t + k*N/NC}; those are NC/4 of the multiply tail's groups {t' + k*N/4} (t' = t + h*N/NC), computed one after the other
// into x[J][k]; then, per digit J and key prime I, the conversion and the first forward stages exactly as ks_head_kernel.
// All-FP64 contexts only.  grid: (N/NC/256, 1, ops)
// -------------------------------------------------------------------------------------------------
template <int L, int KMAX, int PACKM, bool GRID, int PACKK>
__global__ EDGE_BOUNDS(KMAX) void mulrelin_head_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                     const MulOp* __restrict__ twf_base, const u64* __restrict__ D,
                                                                     double* __restrict__ T) {
  using G = EdgeGeom<L>;
  constexpr int NC = G::HEAD_NC;
  // plain degrees: the head's NC = 8 coefficients {t + k*N/8} are two of the tail's groups {t' + k*N/4}, t' = t + h*N/8;
  // lane-split degrees: a lane owns the same 4 coefficients in the tail (its output side) and in the head (its input side)
  constexpr int GROUPS = NC / 4;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  const BufRsrc rout = buf_rsrc(T + (size_t)op * KK * K * N);  // buffer addressing: see BufRow
  double x[KMAX][NC];  // c2's residues = the key-switch digits: canonical integers, as doubles
#pragma unroll
  for (int h = 0; h < GROUPS; h++) {
    const u32 tt = t + (u32)h * (N / 8);
    double res[KMAX][4];
    mul_tail_compute_d<L, KMAX, PACKM, GRID>(ctx, twi_base, D + ((size_t)op * 3 + 2) * R * N, tt, res);
#pragma unroll
    for (int J = 0; J < KMAX; J++)
#pragma unroll
      for (int k = 0; k < 4; k++) x[J][GROUPS * k + h] = res[J][k];
  }
#pragma unroll
  for (int J = 0; J < KMAX; J++) {
    if ((u32)J >= K) break;
    const u64 qJ = ctx->mod[J].q;
    for (u32 I = 0; I < KK; I++) {
      const DevMod& dm = ctx->mod[I];
      const ArithD ar(dm);
      const double* tw = reinterpret_cast<const double*>(twf_base + (size_t)I * N);
      const bool need_reduce = qJ > dm.q;
      double v[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) v[k] = need_reduce ? ar.reduce(x[J][k]) : x[J][k];
      head_fwd_owned<ArithD, L>(ar, v, tw, t);
      ks_head_store<L, PACKK>(ctx, dm, ar, buf_row(rout, ((size_t)I * K + J) * N), I, t, v);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// multiply + relinearize for MIXED contexts (integer-policy data / key primes, FP64 auxiliary base: the north star's
// 3 x 54-bit set) [r06]: the two fused kernels above with the mixed tail (mul_tail_compute_mixed) and the MIXED arms of
// ks_head_kernel / ks_tail_kernel -- c0, c1, c2 of the product stay in registers here too (1.5 MB of HBM traffic per op at N = 8192).
// grid: head (N/NC/256, 1, ops), tail (N/4/256, 2, ops); 8-byte rows throughout.
// -------------------------------------------------------------------------------------------------
#ifndef MRH_MIXED_WAVES
#define MRH_MIXED_WAVES 2
#endif
template <int L, int KMAX>
__global__ __launch_bounds__(kHeadThreads, MRH_MIXED_WAVES) void mulrelin_head_mixed_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                           const MulOp* __restrict__ twf_base, const u64* __restrict__ D,
                                                                           double* __restrict__ T) {
  using G = EdgeGeom<L>;
  constexpr int NC = G::HEAD_NC;
  static_assert(!G::SPLIT, "lane-split degrees have no mixed instantiation");
  constexpr int GROUPS = NC / 4;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  const BufRsrc rout = buf_rsrc(T + (size_t)op * KK * K * N);
  u64 x[KMAX][NC];  // c2's residues = the key-switch digits (canonical)
  auto group = [&](auto h_tag) {  // (a compile-time group index: a rolled loop would index x dynamically -- scratch)
    constexpr int h = decltype(h_tag)::value;
    const u32 tt = t + (u32)h * (N / NC);
    u64 res[4][KMAX];
    mul_tail_compute_mixed<L, KMAX>(ctx, twi_base, D + ((size_t)op * 3 + 2) * R * N, tt, res);
#pragma unroll
    for (int J = 0; J < KMAX; J++)
#pragma unroll
      for (int k = 0; k < 4; k++) x[J][GROUPS * k + h] = res[k][J];
  };
  group(std::integral_constant<int, 0>{});
  if constexpr (GROUPS > 1) group(std::integral_constant<int, 1>{});
  static_assert(GROUPS <= 2, "a head thread owns 4 or 8 coefficients");
#pragma unroll
  for (int J = 0; J < KMAX; J++) {
    if ((u32)J >= K) break;
    const u64 qJ = ctx->mod[J].q;
    for (u32 I = 0; I < KK; I++) {  // exactly ks_head_kernel<L, false, true>
      const DevMod& dm = ctx->mod[I];
      if (!residue_is_f64(dm)) {
        const ArithI ai(dm);
        const bool shrink = qJ > dm.q;
        u64 w[NC];
#pragma unroll
        for (int k = 0; k < NC; k++) w[k] = shrink ? reduce64(x[J][k], dm) : x[J][k];
        head_fwd_owned<ArithI, L>(ai, w, twf_base + (size_t)I * N, t);
#pragma unroll
        for (int k = 0; k < NC; k++) st_head_out<L>(buf_row(rout, ((size_t)I * K + J) * N), t, k, w[k]);
        continue;
      }
      const ArithD ar(dm);
      const bool need_reduce = qJ > dm.q;
      double v[NC];
#pragma unroll
      for (int k = 0; k < NC; k++) v[k] = ar.from_u64(need_reduce ? reduce64(x[J][k], dm) : x[J][k]);
      head_fwd_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twf_base + (size_t)I * N), t);
      nat_store_head<L, false, NtSites<L>::ks_head_st>(buf_row(rout, ((size_t)I * K + J) * N), t, v);
    }
  }
}

template <int L, int KMAX>
__global__ __launch_bounds__(kHeadThreads) void mulrelin_tail_mixed_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                           const u64* __restrict__ D, const double* __restrict__ ACC,
                                                                           const u64* __restrict__ extra, u64* __restrict__ out) {
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 c = blockIdx.y, op = blockIdx.z;
  const u32 K = ctx->K, S = ctx->S, KK = ctx->KK, R = K + S;
  u64 basev[4][KMAX];  // c0 / c1 of the product, canonical
  mul_tail_compute_mixed<L, KMAX>(ctx, twi_base, D + ((size_t)op * 3 + c) * R * N, t, basev);
  const double* acc = ACC + ((size_t)op * 2 + c) * KK * N;
  const BufRsrc racc = buf_rsrc(acc), rout = buf_rsrc(out + ((size_t)op * 2 + c) * K * N);
  const BufRsrc rex = buf_rsrc_opt(extra + ((size_t)op * 2 + c) * K * N, extra != nullptr);
  // from here: the MIXED arm of ks_tail_kernel with the base ciphertext in registers
  u64 tl[4];
  {
    const DevMod& sp = ctx->mod[KK - 1];
    if (!residue_is_f64(sp)) {
      const ArithI ai(sp);
      u64 w[4];
#pragma unroll
      for (int k = 0; k < 4; k++) w[k] = ld_tail_in<L>(buf_row(racc, (size_t)(KK - 1) * N), t, k);
      tail_inv_owned<ArithI, L>(ai, w, twi_base + (size_t)(KK - 1) * N, 0u, t);
#pragma unroll
      for (int k = 0; k < 4; k++) tl[k] = add_mod(ai.scale_canonical(w[k], sp.ninv), ctx->qsp_half, sp.q);
    } else {
      const ArithD ar(sp);
      double v[4];
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = __longlong_as_double((long long)ld_tail_in<L>(buf_row(racc, (size_t)(KK - 1) * N), t, k));
      tail_inv_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twi_base + (size_t)(KK - 1) * N), sp.split_inv_mask, t);
#pragma unroll
      for (int k = 0; k < 4; k++) tl[k] = add_mod(ar.scale_canonical(v[k], sp.ninv_d), ctx->qsp_half, sp.q);
    }
  }
  const u64 qsp = ctx->mod[KK - 1].q;
  u64 cur[4], nxt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) cur[k] = ld_tail_in<L>(buf_row(racc, 0), t, k);
#pragma unroll
  for (int J = 0; J < KMAX; J++) {
    if (J + 1 < KMAX) {
      const BufRow next_row = buf_row(racc, (size_t)((u32)(J + 1) < K ? (u32)(J + 1) : K - 1) * N);
#pragma unroll
      for (int k = 0; k < 4; k++) nxt[k] = ld_tail_in<L>(next_row, t, k);
    }
    if ((u32)J < K) {
      const DevMod& mj = ctx->mod[J];
      u64 ex[4];  // (an absent addend: a descriptor of zero records, the loads return 0)
#pragma unroll
      for (int k = 0; k < 4; k++) ex[k] = ld_tail_out<L>(buf_row(rex, (size_t)J * N), t, k);
      u64 av[4];
      if (!residue_is_f64(mj)) {
        const ArithI ai(mj);
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = cur[k];
        tail_inv_owned<ArithI, L>(ai, w, twi_base + (size_t)J * N, 0u, t);
#pragma unroll
        for (int k = 0; k < 4; k++) av[k] = ai.scale_canonical(w[k], mj.ninv);
      } else {
        const ArithD ar(mj);
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = __longlong_as_double((long long)cur[k]);
        tail_inv_owned<ArithD, L>(ar, v, reinterpret_cast<const double*>(twi_base + (size_t)J * N), mj.split_inv_mask, t);
#pragma unroll
        for (int k = 0; k < 4; k++) av[k] = ar.scale_canonical(v[k], mj.ninv_d);
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        u64 tk = qsp > mj.q ? reduce64(tl[k], mj) : tl[k];
        tk = sub_mod(tk, ctx->qsp_half_mod_q[J], mj.q);
        u64 dd = sub_mod(av[k], tk, mj.q);
        dd = mul_shoup(dd, ctx->inv_qsp_mod_q[J], mj.q);
        const u64 bv = add_mod(basev[k][J], ex[k], mj.q);  // the product's c0 / c1 + a fused Add node's (absent: 0)
        st_tail_out<L>(buf_row(rout, (size_t)J * N), t, k, add_mod(bv, dd, mj.q));
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) cur[k] = nxt[k];
  }
}

// =================================================================================================
// Stand-alone two-kernel transforms for N = 32768, whose 256 KB residue polynomials do not fit one CU's LDS:
// forward = head stages (streaming) + block-local rest; inverse = block-local stages + tail stages (streaming).
// The intermediate lives in place in the data buffer, in the policy's native representation.
// =================================================================================================
__device__ __forceinline__ u32 plan_mod_split(const NttPlan& plan, u32 poly) {  // through the scalar unit: kernels.hip plan_mod
  const u32 i = __builtin_amdgcn_readfirstlane((poly / plan.div) % plan.period);
  const u32* words = reinterpret_cast<const u32*>(plan.mod);
  return (words[i >> 2] >> ((i & 3u) * 8u)) & 0xffu;
}

// grid (N/8/256, polys)
template <int L>
__global__ __launch_bounds__(kHeadThreads) void ntt_head_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base, u64* data,
                                                                NttPlan plan) {
  constexpr int NC = 1 << head_log(L);
  constexpr u32 N = 1u << L, Q = N / NC;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 poly = blockIdx.y;
  const u32 m = plan_mod_split(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * N + t;
  const MulOp* tw = twf_base + (size_t)m * N;
  if (residue_is_f64(dm)) {
    const ArithD ar(dm);
    double v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = ar.from_u64(x[(size_t)k * Q]);
    head_fwd(ar, v, reinterpret_cast<const double*>(tw));
    double* o = reinterpret_cast<double*>(x);
#pragma unroll
    for (int k = 0; k < NC; k++) o[(size_t)k * Q] = v[k];
  } else {
    const ArithI ar(dm);
    u64 v[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) v[k] = x[(size_t)k * Q];
    head_fwd(ar, v, tw);
#pragma unroll
    for (int k = 0; k < NC; k++) x[(size_t)k * Q] = v[k];
  }
}

template <class A, int L>
__device__ __forceinline__ void ntt_midfwd_body(const DevMod& dm, const typename A::Tw* tw, u64* x, typename A::V* smem, u32 tid, u32 blk) {
  using Sh = SplitShape<L>;
  const A ar(dm);
  constexpr int RF0 = split_fwd_radix(L, 0), LOWF0 = split_fwd_low(L, 0);
  using First = BlkPass<A, L, LOWF0, RF0>;
  constexpr int RL = split_fwd_radix(L, Sh::NPF - 1);
  using Last = BlkPass<A, L, 0, RL>;
  typename A::V v[kBlkEPT];
  const typename A::V* src = reinterpret_cast<const typename A::V*>(x);
#pragma unroll
  for (int g = 0; g < First::G; g++)
#pragma unroll
    for (int k = 0; k < (1 << RF0); k++) v[g * (1 << RF0) + k] = src[First::elem(tid, blk, g, k)];
  mid_forward<A, L, 0>(ar, v, smem, tid, blk, tw, dm.split_fwd_mask);
#pragma unroll
  for (int g = 0; g < Last::G; g++)
#pragma unroll
    for (int k = 0; k < (1 << RL); k++) x[Last::elem(tid, blk, g, k)] = ar.canonical(v[g * (1 << RL) + k]);
}

// grid: polys * NBLK
template <int L>
__global__ __launch_bounds__((SplitShape<L>::TPB)) void ntt_midfwd_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twf_base,
                                                                           u64* data, NttPlan plan) {
  using Sh = SplitShape<L>;
  __shared__ u64 smem[Sh::BLOCK];
  const u32 blk = blockIdx.x % Sh::NBLK, poly = blockIdx.x / Sh::NBLK;
  const u32 m = plan_mod_split(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * Sh::N;
  const MulOp* tw = twf_base + (size_t)m * Sh::N;
  if (residue_is_f64(dm))
    ntt_midfwd_body<ArithD, L>(dm, reinterpret_cast<const double*>(tw), x, reinterpret_cast<double*>(smem), threadIdx.x, blk);
  else
    ntt_midfwd_body<ArithI, L>(dm, tw, x, smem, threadIdx.x, blk);
}

template <class A, int L>
__device__ __forceinline__ void ntt_midinv_body(const DevMod& dm, const typename A::Tw* tw, u64* x, typename A::V* smem, u32 tid, u32 blk) {
  using Sh = SplitShape<L>;
  const A ar(dm);
  constexpr int R0 = split_inv_radix(L, 0);
  using In = BlkPass<A, L, 0, R0>;
  constexpr int RI = split_inv_radix(L, Sh::NPI - 1), LOWI = split_inv_low(L, Sh::NPI - 1);
  using Out = BlkPass<A, L, LOWI, RI>;
  typename A::V v[kBlkEPT];
#pragma unroll
  for (int g = 0; g < In::G; g++)
#pragma unroll
    for (int k = 0; k < (1 << R0); k++) v[g * (1 << R0) + k] = ar.from_u64(x[In::elem(tid, blk, g, k)]);
  mid_inverse<A, L, 0>(ar, v, smem, tid, blk, tw, dm.split_inv_mask);
  typename A::V* dst = reinterpret_cast<typename A::V*>(x);
#pragma unroll
  for (int g = 0; g < Out::G; g++)
#pragma unroll
    for (int k = 0; k < (1 << RI); k++) dst[Out::elem(tid, blk, g, k)] = v[g * (1 << RI) + k];
}

template <int L>
__global__ __launch_bounds__((SplitShape<L>::TPB)) void ntt_midinv_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base,
                                                                           u64* data, NttPlan plan) {
  using Sh = SplitShape<L>;
  __shared__ u64 smem[Sh::BLOCK];
  const u32 blk = blockIdx.x % Sh::NBLK, poly = blockIdx.x / Sh::NBLK;
  const u32 m = plan_mod_split(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * Sh::N;
  const MulOp* tw = twi_base + (size_t)m * Sh::N;
  if (residue_is_f64(dm))
    ntt_midinv_body<ArithD, L>(dm, reinterpret_cast<const double*>(tw), x, reinterpret_cast<double*>(smem), threadIdx.x, blk);
  else
    ntt_midinv_body<ArithI, L>(dm, tw, x, smem, threadIdx.x, blk);
}

// grid (N/4/256, polys); scale_mode as in ntt_inv_kernel
template <int L>
__global__ __launch_bounds__(kHeadThreads) void ntt_tail_kernel(const DevCtx* __restrict__ ctx, const MulOp* __restrict__ twi_base, u64* data,
                                                                NttPlan plan, int scale_mode) {
  using G = EdgeGeom<L>;
  constexpr u32 N = 1u << L;
  const u32 t = blockIdx.x * kHeadThreads + threadIdx.x;
  const u32 poly = blockIdx.y;
  const u32 m = plan_mod_split(plan, poly);
  const DevMod& dm = ctx->mod[m];
  u64* x = data + (size_t)poly * N;
  const MulOp* tw = twi_base + (size_t)m * N;
  u64 o[4];
  if (residue_is_f64(dm)) {
    const ArithD ar(dm);
    const MulOpD sc = scale_mode == 1 ? (m < ctx->KK ? ctx->intt_scale_q_d[m] : ctx->intt_scale_bsk_d[m - ctx->KK]) : dm.ninv_d;
    tail_inv4_scale<ArithD, L>(ar, reinterpret_cast<const double*>(x), t, reinterpret_cast<const double*>(tw), sc, dm.split_inv_mask, o);
  } else {
    const ArithI ar(dm);
    MulOp sc = dm.ninv;
    if (scale_mode == 1) sc = m < ctx->KK ? ctx->intt_scale_q[m] : ctx->intt_scale_bsk[m - ctx->KK];
    tail_inv4_scale<ArithI, L>(ar, x, t, tw, sc, 0u, o);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) x[G::tail_out(t, k)] = o[k];
}

template <int L>
static hipError_t ntt_split_t(const DevCtx* ctx, const MulOp* tw, u64* data, size_t polys, const NttPlan& plan, bool inverse, int scale_mode,
                              hipStream_t s) {
  using Sh = SplitShape<L>;
  // gridDim.y <= 65535: chunk the polynomial range (chunks start at multiples of div*period so the plan stays aligned)
  const size_t unit = (size_t)plan.div * plan.period;
  const size_t step = unit <= 32768 ? (32768 / unit) * unit : unit;
  for (size_t off = 0; off < polys; off += step) {
    const size_t cnt = polys - off < step ? polys - off : step;
    u64* d = data + off * Sh::N;
    if (!inverse) {
      ntt_head_kernel<L><<<dim3((Sh::N >> head_log(L)) / kHeadThreads, (unsigned)cnt), kHeadThreads, 0, s>>>(ctx, tw, d, plan);
      ntt_midfwd_kernel<L><<<dim3((unsigned)(cnt * Sh::NBLK)), Sh::TPB, 0, s>>>(ctx, tw, d, plan);
    } else {
      ntt_midinv_kernel<L><<<dim3((unsigned)(cnt * Sh::NBLK)), Sh::TPB, 0, s>>>(ctx, tw, d, plan);
      ntt_tail_kernel<L><<<dim3(Sh::N / 4 / kHeadThreads, (unsigned)cnt), kHeadThreads, 0, s>>>(ctx, tw, d, plan, scale_mode);
    }
  }
  return hipGetLastError();
}

hipError_t launch_ntt_split(const DevCtx* ctx, const MulOp* tw, u32 logn, u64* data, size_t polys, const NttPlan& plan, bool inverse,
                            int scale_mode, hipStream_t s) {
  if (logn == 15) return ntt_split_t<15>(ctx, tw, data, polys, plan, inverse, scale_mode, s);
  return hipErrorInvalidValue;
}

// scratch layout (same size as the unfused path): T = double[ops][KK][K][N], ACC = double[ops][2][KK][N]
#define SPLIT_DISPATCH(fn, ...)              \
  switch (logn) {                            \
    case 12: return fn<12>(__VA_ARGS__);     \
    case 13: return fn<13>(__VA_ARGS__);     \
    case 14: return fn<14>(__VA_ARGS__);     \
    default: return hipErrorInvalidValue;    \
  }

// The key switch also runs split at N = 32768 [r06] -- integer-policy key primes only (SEAL's 55 / 56-bit default set; the mixed
// head / tail instantiations and ks_mid_int_kernel, one digit at a time: 64 KB of LDS per 1024-thread workgroup).  The head and tail
// loop over the primes at run time, so K = 15 costs them no registers; the multiply keeps the whole-polynomial kernels there.
#define KS_DISPATCH(fn, ...)                 \
  switch (logn) {                            \
    case 12: return fn<12>(__VA_ARGS__);     \
    case 13: return fn<13>(__VA_ARGS__);     \
    case 14: return fn<14>(__VA_ARGS__);     \
    case 15: return fn<15>(__VA_ARGS__);     \
    default: return hipErrorInvalidValue;    \
  }

template <int L>
static hipError_t ks_head_t(const DevCtx* ctx, const MulOp* twf, int pack, bool mixed, u32 K, const u64* target, size_t tstride, u64* T, size_t ops, hipStream_t s,
                            u32 ginv) {
  const dim3 grid(EdgeGeom<L>::HEAD_THREADS / kHeadThreads, K, (unsigned)ops);
  if constexpr (L == 15) {
    if (!mixed) return hipErrorInvalidValue;
    ks_head_kernel<L, 0, true><<<grid, kHeadThreads, 0, s>>>(ctx, twf, target, tstride, reinterpret_cast<double*>(T), ginv);
    return hipGetLastError();
  } else if (mixed)
    ks_head_kernel<L, 0, true><<<grid, kHeadThreads, 0, s>>>(ctx, twf, target, tstride, reinterpret_cast<double*>(T), ginv);
  else if (pack == 2) {
    if constexpr (L >= 13) ks_head_kernel<L, 2, false><<<grid, kHeadThreads, 0, s>>>(ctx, twf, target, tstride, reinterpret_cast<double*>(T), ginv);
    else return hipErrorInvalidValue;  // (context.cpp: beside pack_mul == 2 only)
  } else if (pack)
    ks_head_kernel<L, 1, false><<<grid, kHeadThreads, 0, s>>>(ctx, twf, target, tstride, reinterpret_cast<double*>(T), ginv);
  else
    ks_head_kernel<L, 0, false><<<grid, kHeadThreads, 0, s>>>(ctx, twf, target, tstride, reinterpret_cast<double*>(T), ginv);
  return hipGetLastError();
}
// pack: DevCtx::pack_ks of the context behind `ctx` (48-bit packed intermediates, see nat_load); mixed: DevCtx::ks_ni != 0
// ginv != 0: the target is sigma_g(target) for the Galois element g = ginv^-1 mod 2N, read through the automorphism
hipError_t launch_ks_head(const DevCtx* ctx, const MulOp* twf, u32 logn, int pack, bool mixed, u32 K, const u64* target, size_t tstride, u64* T, size_t ops, hipStream_t s,
                          u32 ginv) {
  KS_DISPATCH(ks_head_t, ctx, twf, pack, mixed, K, target, tstride, T, ops, s, ginv)
}

// res_d / nd, res_dp / ndp, res_i / ni: device lists (inside the DevCtx) of the key primes that take the FP64 policy with 8-byte rows,
// the FP64 policy with 48-bit packed rows and the integer policy
template <int L>
static hipError_t ks_mid_t(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, bool pack_acc, const unsigned char* res_d, u32 nd, const unsigned char* res_dp,
                           u32 ndp, const unsigned char* res_i, u32 ni, const u64* T, const u64* key, u64* ACC, size_t ops, hipStream_t s, KeyMap km) {
  using Sh = SplitShape<L>;
  if (((reinterpret_cast<uintptr_t>(res_d) | reinterpret_cast<uintptr_t>(res_dp) | reinterpret_cast<uintptr_t>(res_i)) & 3u) != 0)
    return hipErrorInvalidValue;  // residue_of reads words
  const size_t ops8 = (ops + 7) / 8 * 8;
  if constexpr (L == 15) {
    if (nd || ndp) return hipErrorInvalidValue;  // integer-policy key primes only at this degree (context.cpp: ks_split_ok)
  } else {
    if (ndp) {
      const dim3 grid((unsigned)(ops8 * ndp * Sh::NBLK));
      if (pack_acc)
        ks_mid_kernel<L, true><<<grid, SplitShape<L, KS_EPT(L)>::TPB, 0, s>>>(ctx, twf, twi, reinterpret_cast<const double*>(T), key, reinterpret_cast<double*>(ACC), (u32)ops, res_dp, ndp, km);
      else
        ks_mid_kernel<L, true, KS_EPT(L), false><<<grid, SplitShape<L, KS_EPT(L)>::TPB, 0, s>>>(ctx, twf, twi, reinterpret_cast<const double*>(T), key, reinterpret_cast<double*>(ACC), (u32)ops, res_dp, ndp, km);
    }
    if (nd) {
      const dim3 grid((unsigned)(ops8 * nd * Sh::NBLK));
      ks_mid_kernel<L, false><<<grid, SplitShape<L, KS_EPT(L)>::TPB, 0, s>>>(ctx, twf, twi, reinterpret_cast<const double*>(T), key, reinterpret_cast<double*>(ACC), (u32)ops, res_d, nd, km);
    }
  }
  if (ni) {
    const dim3 grid((unsigned)(ops8 * ni * Sh::NBLK));
    ks_mid_int_kernel<L><<<grid, SplitShape<L, KS_MID_INT_EPT(L)>::TPB, 0, s>>>(ctx, twf, twi, T, key, ACC, (u32)ops, res_i, ni, km);
  }
  return hipGetLastError();
}
// the residue lists of the context behind `ctx` (DevCtx::ks_res_d / ks_res_dp / ks_res_i and their counts, read on the host); the
// accumulator rows are packed when EVERY key prime's are (pack_ks == 1)
hipError_t launch_ks_mid(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, const DevCtx& h, const u64* T, const u64* key, u64* ACC, size_t ops,
                         hipStream_t s, KeyMap km) {
  KS_DISPATCH(ks_mid_t, ctx, twf, twi, h.pack_ks == 1, ctx->ks_res_d, h.ks_nd, ctx->ks_res_dp, h.ks_ndp, ctx->ks_res_i, h.ks_ni, T, key, ACC, ops, s, km)
}

template <int L>
static hipError_t ks_tail_t(const DevCtx* ctx, const MulOp* twi, int pack_ks, bool mixed, const u64* ACC, const u64* base, size_t bstride, u32 base_mask,
                            const u64* extra, u64* out2, size_t ops, hipStream_t s, u32 ginv) {
  const dim3 grid((1u << L) / 4 / kHeadThreads, 2, (unsigned)ops);
  if constexpr (L == 15) {
    if (!mixed) return hipErrorInvalidValue;
    ks_tail_kernel<L, false, true><<<grid, kHeadThreads, 0, s>>>(ctx, twi, reinterpret_cast<const double*>(ACC), base, bstride, base_mask, extra, out2, ginv);
    return hipGetLastError();
  } else if (mixed)
    ks_tail_kernel<L, false, true><<<grid, kHeadThreads, 0, s>>>(ctx, twi, reinterpret_cast<const double*>(ACC), base, bstride, base_mask, extra, out2, ginv);
  else if (pack_ks == 1)  // (2 = per key prime: the rows of T only, the accumulator rows are doubles)
    ks_tail_kernel<L, true, false><<<grid, kHeadThreads, 0, s>>>(ctx, twi, reinterpret_cast<const double*>(ACC), base, bstride, base_mask, extra, out2, ginv);
  else
    ks_tail_kernel<L, false, false><<<grid, kHeadThreads, 0, s>>>(ctx, twi, reinterpret_cast<const double*>(ACC), base, bstride, base_mask, extra, out2, ginv);
  return hipGetLastError();
}
// extra: optional ciphertexts u64[ops][2][K][N] added to the result; ginv != 0: the base polynomials are read through sigma_g (launch_ks_head)
hipError_t launch_ks_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, int pack_ks, bool mixed, const u64* ACC, const u64* base, size_t bstride, u32 base_mask,
                          const u64* extra, u64* out2, size_t ops, hipStream_t s, u32 ginv) {
  KS_DISPATCH(ks_tail_t, ctx, twi, pack_ks, mixed, ACC, base, bstride, base_mask, extra, out2, ops, s, ginv)
}

template <int L>
static hipError_t mul_head_t(const DevCtx* ctx, const MulOp* twf, bool aux_f64, int pack, u32 kneed, const u64* a, const u64* b, u64* ext,
                             size_t ops, u32 npolys, hipStream_t s, const MemberHead* members, u32 mfirst, u32 mper) {
  const dim3 grid(EdgeGeom<L>::HEAD_THREADS / kHeadThreads, npolys, (unsigned)ops);
  const bool head_grid = (pack & 4) != 0;  // bit 2 of `pack`: DevCtx::conv_grid == 1 (evaluator.cpp), the exact-sum extension
  pack &= 3;
  if (kneed > 4) {  // only the all-FP64 instantiation exists for 5..8 data primes (evaluator.cpp checks)
    if (pack == 2)
      mul_head_kernel<L, 8, true, 2><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
    else if (pack)
      mul_head_kernel<L, 8, true, 1><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
    else
      mul_head_kernel<L, 8, true, 0><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
  } else if (aux_f64) {
    if (pack == 2) return hipErrorInvalidValue;  // per-row packing exists in the 8-prime instantiations (context.cpp)
    if (head_grid) {
      if (pack)
        mul_head_kernel<L, 4, true, 1, true><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
      else
        mul_head_kernel<L, 4, true, 0, true><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
    } else if (pack)
      mul_head_kernel<L, 4, true, 1><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
    else
      mul_head_kernel<L, 4, true, 0><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
  } else if (pack) {  // mixed base: integer data primes, FP64 auxiliary primes (DevCtx::aux_mixed)
    mul_head_kernel<L, 4, false, 1><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
  } else {
    mul_head_kernel<L, 4, false, 0><<<grid, kHeadThreads, 0, s>>>(ctx, twf, a, b, ext, members, mfirst, mper);
  }
  return hipGetLastError();
}
// aux_f64: DevCtx::aux_f64 of the context behind `ctx` (selects the all-FP64 instantiation)
// kneed: max(data primes, auxiliary primes - 2) -- selects the 4- or 8-prime instantiation
// pack: DevCtx::pack_mul (0 / 1 / 2) with aux_f64, | 4 = exact-sum extension (DevCtx::conv_grid == 1; 4-prime all-FP64 instantiation);
//       WITHOUT aux_f64, non-zero selects the mixed-base instantiation (DevCtx::aux_mixed)
// npolys: 4 = (a0, a1, b0, b1); 2 = the first operand only (squaring: ext polys 2, 3 stay unwritten and unread)
hipError_t launch_mul_head(const DevCtx* ctx, const MulOp* twf, u32 logn, bool aux_f64, int pack, u32 kneed, const u64* a, const u64* b, u64* ext,
                           size_t ops, hipStream_t s, u32 npolys, const MemberHead* members, u32 first, u32 per) {
  if (members && !per) return hipErrorInvalidValue;
  SPLIT_DISPATCH(mul_head_t, ctx, twf, aux_f64, pack, kneed, a, b, ext, ops, npolys, s, members, first, per)
}

template <int L>
static hipError_t mul_mid_t(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, const unsigned char* res_dp, u32 ndp, const unsigned char* res_d, u32 nd,
                            const unsigned char* res_i, u32 ni, const u64* ext, u64* D, size_t ops, bool square, hipStream_t s) {
  using Sh = SplitShape<L>;
  if (((reinterpret_cast<uintptr_t>(res_dp) | reinterpret_cast<uintptr_t>(res_d) | reinterpret_cast<uintptr_t>(res_i)) & 3u) != 0) return hipErrorInvalidValue;
  const size_t ops_g = MUL_MID_SLICE(L) ? (ops + 7) / 8 * 8 : ops;  // the slice-major order deals whole groups of 8 ops to the XCDs
  constexpr unsigned TDP = MulMidGeom<L, true, false, true>::TPB, TD = MulMidGeom<L, true, false, false>::TPB, TI = MulMidGeom<L, false>::TPB;
  if (square) {
    constexpr unsigned TSP = MulMidGeom<L, true, true, true>::TPB, TS = MulMidGeom<L, true, true, false>::TPB;
    if (ndp) mul_mid_kernel<L, true, true, true><<<dim3((unsigned)(ops_g * ndp * Sh::NBLK)), TSP, 0, s>>>(ctx, twf, twi, ext, D, res_dp, ndp, (u32)ops);
    if (nd) mul_mid_kernel<L, true, false, true><<<dim3((unsigned)(ops_g * nd * Sh::NBLK)), TS, 0, s>>>(ctx, twf, twi, ext, D, res_d, nd, (u32)ops);
    if (ni) mul_mid_kernel<L, false, false, true><<<dim3((unsigned)(ops_g * ni * Sh::NBLK)), TI, 0, s>>>(ctx, twf, twi, ext, D, res_i, ni, (u32)ops);
    return hipGetLastError();
  }
  if (ndp) mul_mid_kernel<L, true, true><<<dim3((unsigned)(ops_g * ndp * Sh::NBLK)), TDP, 0, s>>>(ctx, twf, twi, ext, D, res_dp, ndp, (u32)ops);
  if (nd) mul_mid_kernel<L, true, false><<<dim3((unsigned)(ops_g * nd * Sh::NBLK)), TD, 0, s>>>(ctx, twf, twi, ext, D, res_d, nd, (u32)ops);
  if (ni) mul_mid_kernel<L, false, false><<<dim3((unsigned)(ops_g * ni * Sh::NBLK)), TI, 0, s>>>(ctx, twf, twi, ext, D, res_i, ni, (u32)ops);
  return hipGetLastError();
}
// res_dp / res_d / res_i: device arrays listing the residue indices (0..R-1) handled by the FP64 instantiation with 48-bit packed
// rows, by the FP64 instantiation with 8-byte rows and by the integer instantiation (DevCtx::mid_res_*)
// square: the operands are one ciphertext -- ext holds polys 0, 1 only (launch_mul_head with npolys = 2)
hipError_t launch_mul_mid(const DevCtx* ctx, const MulOp* twf, const MulOp* twi, u32 logn, const unsigned char* res_dp, u32 ndp, const unsigned char* res_d, u32 nd,
                          const unsigned char* res_i, u32 ni, const u64* ext, u64* D, size_t ops, hipStream_t s, bool square) {
  SPLIT_DISPATCH(mul_mid_t, ctx, twf, twi, res_dp, ndp, res_d, nd, res_i, ni, ext, D, ops, square, s)
}

template <int L>
static hipError_t mul_tail_t(const DevCtx* ctx, const MulOp* twi, bool aux_f64, int pack, bool conv_grid, u32 kneed, const u64* D, u64* out, size_t ops,
                             u32 poly0, u32 npolys, hipStream_t s) {
  const dim3 grid((1u << L) / 4 / kHeadThreads, npolys, (unsigned)ops);
#define MT(KM, AD, PK, GR) mul_tail_kernel<L, KM, AD, PK, GR><<<grid, kHeadThreads, 0, s>>>(ctx, twi, D, out, poly0, npolys)
  if (kneed > 4) {
    if (pack == 2) { if (conv_grid) MT(8, true, 2, true); else MT(8, true, 2, false); }
    else if (pack) { if (conv_grid) MT(8, true, 1, true); else MT(8, true, 1, false); }
    else { if (conv_grid) MT(8, true, 0, true); else MT(8, true, 0, false); }
  } else if (aux_f64) {
    if (pack == 2) return hipErrorInvalidValue;
    if (pack) MT(4, true, 1, false); else MT(4, true, 0, false);
  } else if (pack) {  // mixed base
    MT(4, false, 1, false);
  } else {
    MT(4, false, 0, false);
  }
#undef MT
  return hipGetLastError();
}
// conv_grid: DevCtx::conv_grid (takes effect in the 8-prime instantiation, kneed > 4)
// poly0, npolys: which product polynomials to finish (0, 3 = all; 2, 1 = only c2, written compactly as out[op][K][N])
hipError_t launch_mul_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, bool aux_f64, int pack, bool conv_grid, u32 kneed, const u64* D, u64* out,
                           size_t ops, hipStream_t s, u32 poly0, u32 npolys) {
  SPLIT_DISPATCH(mul_tail_t, ctx, twi, aux_f64, pack, conv_grid && aux_f64, kneed, D, out, ops, poly0, npolys, s)
}

template <int L>
static hipError_t mulrelin_tail_t(const DevCtx* ctx, const MulOp* twi, int pack_mul, bool conv_grid, int pack_ks, u32 kneed, const u64* D, const u64* ACC,
                                  const u64* extra, u64* out2, size_t ops, hipStream_t s, const MemberTail* members, u32 mfirst, u32 mper) {
  const dim3 grid((1u << L) / 4 / kHeadThreads, 2, (unsigned)ops);
  const double* acc = reinterpret_cast<const double*>(ACC);
#define MRT(KM, PM, GR, PK) mulrelin_tail_kernel<L, KM, PM, GR, PK><<<grid, kHeadThreads, 0, s>>>(ctx, twi, D, acc, extra, out2, members, mfirst, mper)
// (pack_ks == 2, per key prime: the rows of T only -- the accumulator rows are doubles)
#define MRT_K(KM, PM, GR) do { if (pack_ks == 1) MRT(KM, PM, GR, true); else MRT(KM, PM, GR, false); } while (0)
  if (kneed > 4) {
    if (pack_mul == 2) { if (conv_grid) MRT_K(8, 2, true); else MRT_K(8, 2, false); }
    else if (pack_mul) { if (conv_grid) MRT_K(8, 1, true); else MRT_K(8, 1, false); }
    else { if (conv_grid) MRT_K(8, 0, true); else MRT_K(8, 0, false); }
  } else {
    if (pack_mul == 2) return hipErrorInvalidValue;
    if (pack_mul) MRT_K(4, 1, false); else MRT_K(4, 0, false);
  }
#undef MRT_K
#undef MRT
  return hipGetLastError();
}
template <int L>
static hipError_t mulrelin_head_t(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, int pack_mul, bool conv_grid, int pack_ks, u32 kneed, const u64* D,
                                  u64* T, size_t ops, hipStream_t s) {
  const dim3 grid(EdgeGeom<L>::HEAD_THREADS / kHeadThreads, 1, (unsigned)ops);
  double* t = reinterpret_cast<double*>(T);
#define MRH(KM, PM, GR, PK) mulrelin_head_kernel<L, KM, PM, GR, PK><<<grid, kHeadThreads, 0, s>>>(ctx, twi, twf, D, t)
#define MRH_K(KM, PM, GR) do { if (pack_ks == 2) return hipErrorInvalidValue; if (pack_ks) MRH(KM, PM, GR, 1); else MRH(KM, PM, GR, 0); } while (0)
// (per-row key-switch rows come with the per-row multiply only: context.cpp)
#define MRH_K2(GR) do { if (pack_ks == 2) MRH(8, 2, GR, 2); else if (pack_ks) MRH(8, 2, GR, 1); else MRH(8, 2, GR, 0); } while (0)
  if (kneed > 4) {
    if (pack_mul == 2) { if (conv_grid) MRH_K2(true); else MRH_K2(false); }
    else if (pack_mul) { if (conv_grid) MRH_K(8, 1, true); else MRH_K(8, 1, false); }
    else { if (conv_grid) MRH_K(8, 0, true); else MRH_K(8, 0, false); }
  } else {
    if (pack_mul == 2) return hipErrorInvalidValue;
    if (pack_mul) MRH_K(4, 1, false); else MRH_K(4, 0, false);
  }
#undef MRH_K2
#undef MRH_K
#undef MRH
  return hipGetLastError();
}
// multiply tail of c2 + key-switch head in one kernel (all-FP64 contexts)
hipError_t launch_mulrelin_head(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, u32 logn, int pack_mul, bool conv_grid, int pack_ks, u32 kneed,
                                const u64* D, u64* T, size_t ops, hipStream_t s) {
  SPLIT_DISPATCH(mulrelin_head_t, ctx, twi, twf, pack_mul, conv_grid, pack_ks, kneed, D, T, ops, s)
}

// the last kernel of the fused multiply + relinearize of all-FP64 contexts (DevCtx::aux_f64, every key prime FP64-policy)
template <int L>
static hipError_t mulrelin_head_mixed_t(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, const u64* D, u64* T, size_t ops, hipStream_t s) {
  if constexpr (EdgeGeom<L>::SPLIT) {
    return hipErrorInvalidValue;
  } else {
    const dim3 grid(EdgeGeom<L>::HEAD_THREADS / kHeadThreads, 1, (unsigned)ops);
    mulrelin_head_mixed_kernel<L, 4><<<grid, kHeadThreads, 0, s>>>(ctx, twi, twf, D, reinterpret_cast<double*>(T));
    return hipGetLastError();
  }
}
template <int L>
static hipError_t mulrelin_tail_mixed_t(const DevCtx* ctx, const MulOp* twi, const u64* D, const u64* ACC, const u64* extra, u64* out2, size_t ops, hipStream_t s) {
  if constexpr (EdgeGeom<L>::SPLIT) {
    return hipErrorInvalidValue;
  } else {
    const dim3 grid((1u << L) / 4 / kHeadThreads, 2, (unsigned)ops);
    mulrelin_tail_mixed_kernel<L, 4><<<grid, kHeadThreads, 0, s>>>(ctx, twi, D, reinterpret_cast<const double*>(ACC), extra, out2);
    return hipGetLastError();
  }
}
// the mixed forms (DevCtx::aux_mixed, K <= 4 data primes): 8-byte rows of either policy in D, T and ACC
hipError_t launch_mulrelin_head_mixed(const DevCtx* ctx, const MulOp* twi, const MulOp* twf, u32 logn, const u64* D, u64* T, size_t ops, hipStream_t s) {
  SPLIT_DISPATCH(mulrelin_head_mixed_t, ctx, twi, twf, D, T, ops, s)
}
hipError_t launch_mulrelin_tail_mixed(const DevCtx* ctx, const MulOp* twi, u32 logn, const u64* D, const u64* ACC, const u64* extra, u64* out2, size_t ops,
                                      hipStream_t s) {
  SPLIT_DISPATCH(mulrelin_tail_mixed_t, ctx, twi, D, ACC, extra, out2, ops, s)
}

hipError_t launch_mulrelin_tail(const DevCtx* ctx, const MulOp* twi, u32 logn, int pack_mul, bool conv_grid, int pack_ks, u32 kneed, const u64* D,
                                const u64* ACC, const u64* extra, u64* out2, size_t ops, hipStream_t s, const MemberTail* members, u32 first, u32 per) {
  if (members && (!per || extra)) return hipErrorInvalidValue;
  SPLIT_DISPATCH(mulrelin_tail_t, ctx, twi, pack_mul, conv_grid, pack_ks, kneed, D, ACC, extra, out2, ops, s, members, first, per)
}

}  // namespace hipbfv
