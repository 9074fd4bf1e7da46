// sunscreen_amd/csrc/nttshape.hpp -- pass structure of the LDS-staged NTT, shared by the kernels
// (compile time) and by the host-side FP64 range simulation in context.cpp.
#pragma once

namespace hipbfv {

constexpr int kElemsPerThread = 16;

// log2(N) radix-2 stages are grouped into ceil(logn/4) register passes of 3 or 4 stages.
constexpr int ntt_num_passes(int logn) { return (logn + 3) / 4; }
constexpr int ntt_pass_radix(int logn, int p) {
  return logn / ntt_num_passes(logn) + (p < logn % ntt_num_passes(logn) ? 1 : 0);
}
constexpr int ntt_stages_before(int logn, int p) {
  return p * (logn / ntt_num_passes(logn)) + (p < logn % ntt_num_passes(logn) ? p : logn % ntt_num_passes(logn));
}

}  // namespace hipbfv
