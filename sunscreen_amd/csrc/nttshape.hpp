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
// Head depth per degree: 3 stages (8 coefficients per head thread) except at N = 16384, where 2 stages (4 coefficients)
// halve the head kernels' register footprint at K = 8 and the middle kernel absorbs the extra stage without an extra pass
// (12 stages = 3,3,3,3 instead of 11 = 2,3,3,3).
#ifndef HIPBFV_HEAD_LOG_14
#define HIPBFV_HEAD_LOG_14 2
#endif
// Depth 2 is not selectable at N = 8192: its forward sequence would end with a radix-8 pass while the inverse starts
// with a radix-4 one, and the pointwise work happens in the layout both must share (checked in SplitShape) -- and
// mul_head is already at the copy rate there.
static_assert(HIPBFV_HEAD_LOG_14 == 2 || HIPBFV_HEAD_LOG_14 == 3, "head depth at N = 16384 is 2 or 3");
constexpr int head_log(int logn) { return logn == 14 ? HIPBFV_HEAD_LOG_14 : 3; }
constexpr int kHeadLogMax = 3;
constexpr int kTailLog = 2;
// Elements per middle-kernel thread: 8 (radix <= 8 passes).  4 (radix <= 4 passes, twice the threads, about half the
// registers, ~40 % more passes) is a build-time experiment (-DHIPBFV_BLK_EPT=4), see DESIGN.md 5.5.
#ifndef HIPBFV_BLK_EPT
#define HIPBFV_BLK_EPT 8
#endif
constexpr int kBlkEPT = HIPBFV_BLK_EPT;
static_assert(kBlkEPT == 8 || kBlkEPT == 4, "middle kernels hold 8 or 4 elements per thread");
constexpr int split_fwd_passes(int logn) { return kBlkEPT == 4 ? (logn - head_log(logn) + 1) / 2 : (logn - head_log(logn) + 2) / 3; }
constexpr int split_inv_passes(int logn) { return kBlkEPT == 4 ? (logn - kTailLog + 1) / 2 : (logn - kTailLog + 2) / 3; }
constexpr int split_fwd_radix(int logn, int p) {
  // 4 per thread: radix 4 throughout, an odd stage count starts with one radix-2 pass
  if (kBlkEPT == 4) return (((logn - head_log(logn)) & 1) && p == 0) ? 1 : 2;
  // remaining stages: 9 -> 3,3,3 ; 10 -> 3,3,2,2 ; 11 -> 2,3,3,3 ; 12 -> 3,3,3,3
  return logn - head_log(logn) == 10 ? (p < 2 ? 3 : 2) : logn - head_log(logn) == 11 ? (p == 0 ? 2 : 3) : 3;
}
constexpr int split_inv_radix(int logn, int p) {
  // 4 per thread: radix 4 throughout, an odd stage count ends with one radix-2 pass
  if (kBlkEPT == 4) return (((logn - kTailLog) & 1) && p == split_inv_passes(logn) - 1) ? 1 : 2;
  // logn-2 stages, first radix == last forward radix: 10 -> 3,3,2,2 ; 11 -> 2,3,3,3 ; 12 -> 3,3,3,3 ; 13 -> 3,3,3,2,2
  return logn - kTailLog == 10 ? (p < 2 ? 3 : 2) : logn - kTailLog == 11 ? (p == 0 ? 2 : 3) : logn - kTailLog == 13 ? (p < 3 ? 3 : 2) : 3;
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
