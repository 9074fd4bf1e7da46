#!/bin/bash
# tools/asan_cpu_suite.sh -- the host side of libhipbfv.so (capi, context, evaluator, evaluator_client, program, program_plan, wire) rebuilt
# with AddressSanitizer + UndefinedBehaviorSanitizer (g++; the hipcc-built kernel objects are linked in as they are) and the
# CPU test suite run against that build (HIPBFV_LIB).  No GPU needed: covers every host-only entry point the suite reaches --
# parameter objects, prime generation, wire format, decoders under mutation, program loading, the C ABI surface checks.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); cd $ROOT
make -s -C sunscreen_amd/csrc -j8
mkdir -p build/asan
for f in capi context evaluator evaluator_client program program_plan wire; do
  g++ -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -x c++ \
      -c sunscreen_amd/csrc/$f.cpp -o build/asan/$f.o &
done
wait
g++ -shared -fsanitize=address,undefined -o build/asan/libhipbfv_asan.so build/asan/{capi,context,evaluator,evaluator_client,program,program_plan,wire}.o \
    build/hipbfv_kernels.o build/hipbfv_kernels_split.o build/hipbfv_kernels_client.o -L/opt/rocm/lib -lamdhip64 -ldl -Wl,-rpath,/opt/rocm/lib
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  HIPBFV_LIB=$ROOT/build/asan/libhipbfv_asan.so python -m pytest tests -q -m "not gpu" --deselect tests/test_cpp_mirror.py \
  --deselect tests/test_decoder_fuzz_cpu.py::test_parsers_are_clean_under_address_and_ub_sanitizers "$@"
