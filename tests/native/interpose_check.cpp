// tests/native/interpose_check.cpp -- regression test for the failure of commit 6df2a2d ("one of two runs of
// examples/simple_multiply on the GPU box failed").  Root cause: libhipbfv.so exported its internal C++ classes
// (namespace hipbfv) next to the C ABI, and the first include/hipbfv.hpp used the same namespace; at -O0 the
// executable's out-of-line copy of `hipbfv::Context::~Context()` (a different class with the same mangled name) was
// bound in place of the library's own through ordinary ELF symbol interposition, and ran on the library's object inside
// SEALContext_Create.  Fix: the library is built with -fvisibility=hidden and exports the C ABI only, so nothing in it
// can be interposed.  This program DEFINES the colliding symbol on purpose and exports it (-rdynamic); with a correct
// library SEALContext_Create never reaches it.
#include <cstdio>
#include <cstdlib>

#include "hipbfv.h"

namespace hipbfv {
class Context {
 public:
  ~Context();
  int marker = 0;
};
Context::~Context() {
  std::puts("INTERPOSED: the executable's hipbfv::Context::~Context() ran inside libhipbfv.so");
  std::fflush(stdout);
  std::_Exit(42);
}
}  // namespace hipbfv

int main() {
  if (std::getenv("NEVER_SET_HIPBFV")) { hipbfv::Context keep_the_symbol_alive; (void)keep_the_symbol_alive; }
  void* params = nullptr;
  if (EncParams_Create1(1, &params)) return 2;
  if (EncParams_SetPolyModulusDegree(params, 4096)) return 3;
  uint64_t len = 0;
  if (CoeffModulus_BFVDefault(4096, 128, &len, nullptr)) return 4;
  void* coeffs[16];
  if (len > 16 || CoeffModulus_BFVDefault(4096, 128, &len, coeffs)) return 5;
  if (EncParams_SetCoeffModulus(params, len, coeffs)) return 6;
  if (EncParams_SetPlainModulus2(params, 65537)) return 7;
  void* ctx = nullptr;
  // with a GPU this succeeds and SEALContext_Destroy runs the library's destructor; without one the library builds
  // the context on the host, fails at the upload and destroys it: either way its OWN ~Context() must run
  const long hr = SEALContext_Create(params, true, 128, &ctx);
  if (hr == 0 && SEALContext_Destroy(ctx)) return 8;
  std::printf("no interposition (SEALContext_Create -> 0x%lx)\n", (unsigned long)hr);
  return 0;
}
