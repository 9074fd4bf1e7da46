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

}  // namespace hipbfv
