// sunscreen_amd/csrc/nttshape.hpp -- pass structure of the LDS-staged NTT, shared by the kernels
// (compile time) and by the host-side FP64 range simulation in context.cpp.
#pragma once

namespace hipbfv {

// Elements per thread (EPT) is 16 for the stand-alone transforms (N/16 threads, radix <= 16 passes) and 8
// for the fused kernels at N <= 8192 (N/8 threads, radix <= 8 passes, fewer registers per thread).
constexpr int kElemsPerThread = 16;

constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
// log2(N) radix-2 stages are grouped into ceil(logn/log2(EPT)) register passes.
constexpr int ntt_num_passes(int logn, int ept = kElemsPerThread) { return (logn + ilog2(ept) - 1) / ilog2(ept); }
constexpr int ntt_pass_radix(int logn, int p, int ept = kElemsPerThread) {
  return logn / ntt_num_passes(logn, ept) + (p < logn % ntt_num_passes(logn, ept) ? 1 : 0);
}
constexpr int ntt_stages_before(int logn, int p, int ept = kElemsPerThread) {
  return p * (logn / ntt_num_passes(logn, ept)) + (p < logn % ntt_num_passes(logn, ept) ? p : logn % ntt_num_passes(logn, ept));
}


// ---- split ("head / middle / tail") transforms -------------------------------------------------
// The first head_log(logn) forward stages (gaps >= N/8, or N/4) are done by the coefficient-parallel producer kernel,
// the last kTailLog inverse stages (gaps >= N/4) by the coefficient-parallel consumer kernel; everything
// in between is local to contiguous blocks of N/4 coefficients and runs in one "middle" kernel per block
// with 8 elements per thread.  Pass radices of the middle kernel:
// Head / tail depth per degree.  N <= 8192: the head does 3 forward stages (a head thread owns the 8 coefficients
// {t + k*N/8}), the tail the last 2 inverse stages ({t + k*N/4}), the middle kernels work on blocks of N/4 coefficients.
// N = 16384 (HIPBFV_GEOM14):
//   4 (default) = blocks of N/4 = 4096 coefficients, head 2 stages, tail 2 stages; one 128 KB-LDS middle workgroup per CU.
//   8           = the block size of N = 8192 (N/8 = 2048 coefficients, 16 KB per transform in flight, two to three middle
//                 workgroups per CU): head AND tail each take 3 stages, and because 8 coefficients per thread are too many
//                 registers at 8 data primes, a PAIR of lanes (l, l + 32) shares each set of 8 (4 each; one of the three
//                 stages sits between two halves that trade two values through v_permlane32_swap: lane_split, EdgeGeom).
//     Measured in round 2 (interleaved A/B, n = 16384, K = 8+1, per 1024 ops): the middle kernels gain what the occupancy
//     experiment predicted -- mul_mid 6.39 -> 5.28 ms, ks_mid 5.59 -> 5.01 -- but the head / tail kernels, which run at 2
//     waves per SIMD there, pay more for their third stage and the exchange than that: mul_head 2.71 -> 3.48, fused ks_head
//     2.45 -> 3.16, fused ks_tail 3.30 -> 4.02: 49.6 K -> 48.4 K mul+relin/s.  Bit-exact (the GPU suite runs the geom-8 build
//     as a variant library, tests/test_gpu_properties.py); 4 stays the default.
#ifndef HIPBFV_GEOM14
#define HIPBFV_GEOM14 4
#endif
static_assert(HIPBFV_GEOM14 == 8 || HIPBFV_GEOM14 == 4, "N = 16384 middle blocks are N/8 or N/4 coefficients");
constexpr bool lane_split(int logn) { return logn == 14 && HIPBFV_GEOM14 == 8; }
constexpr int head_log(int logn) { return logn == 14 ? (HIPBFV_GEOM14 == 8 ? 3 : 2) : 3; }
constexpr int tail_log(int logn) { return lane_split(logn) ? 3 : 2; }
constexpr int kHeadLogMax = 3;
constexpr int kTailLog = 2;  // degrees without lane splitting
// Elements per middle-kernel thread: 8 (radix <= 8 passes).  4 (radix <= 4 passes, twice the threads, about half the
// registers, ~40 % more passes) is a build-time experiment (-DHIPBFV_BLK_EPT=4), see HISTORY.md 5.5.
#ifndef HIPBFV_BLK_EPT
#define HIPBFV_BLK_EPT 8
#endif
constexpr int kBlkEPT = HIPBFV_BLK_EPT;
static_assert(kBlkEPT == 8 || kBlkEPT == 4, "middle kernels hold 8 or 4 elements per thread");
constexpr int split_fwd_passes(int logn) { return kBlkEPT == 4 ? (logn - head_log(logn) + 1) / 2 : (logn - head_log(logn) + 2) / 3; }
constexpr int split_inv_passes(int logn) { return kBlkEPT == 4 ? (logn - tail_log(logn) + 1) / 2 : (logn - tail_log(logn) + 2) / 3; }
constexpr int split_fwd_radix(int logn, int p) {
  // 4 per thread: radix 4 throughout, an odd stage count starts with one radix-2 pass
  if (kBlkEPT == 4) return (((logn - head_log(logn)) & 1) && p == 0) ? 1 : 2;
  // remaining stages: 9 -> 3,3,3 ; 10 -> 3,3,2,2 ; 11 -> 3,3,3,2 (the inverse of 11 stages starts with a radix-4 pass) ; 12 -> 3,3,3,3
  return logn - head_log(logn) == 10 ? (p < 2 ? 3 : 2) : logn - head_log(logn) == 11 ? (p < 3 ? 3 : 2) : 3;
}
constexpr int split_inv_radix(int logn, int p) {
  // 4 per thread: radix 4 throughout, an odd stage count ends with one radix-2 pass
  if (kBlkEPT == 4) return (((logn - tail_log(logn)) & 1) && p == split_inv_passes(logn) - 1) ? 1 : 2;
  // logn - tail stages, first radix == last forward radix: 10 -> 3,3,2,2 ; 11 -> 2,3,3,3 ; 12 -> 3,3,3,3 ; 13 -> 3,3,3,2,2
  return logn - tail_log(logn) == 10 ? (p < 2 ? 3 : 2) : logn - tail_log(logn) == 11 ? (p == 0 ? 2 : 3) : logn - tail_log(logn) == 13 ? (p < 3 ? 3 : 2) : 3;
}
constexpr int split_fwd_low(int logn, int p) {  // lowest index bit of the window of forward middle pass p
  int s = head_log(logn);
  for (int i = 0; i <= p; i++) s += split_fwd_radix(logn, i);
  return logn - s;
}
constexpr int split_inv_low(int logn, int p) {
  int s = 0;
  for (int i = 0; i < p; i++) s += split_inv_radix(logn, i);
  return s;
}
constexpr bool split_supported(int logn) { return logn >= 12 && logn <= 14; }

}  // namespace hipbfv
