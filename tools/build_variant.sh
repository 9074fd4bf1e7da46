#!/bin/bash
# tools/build_variant.sh <tag> <-D flags...> -- build sunscreen_amd/lib/variants/libhipbfv_<tag>.so with extra
# macro definitions for the split kernels (tuning A/B runs; select it with HIPBFV_LIB=<path>).  capi.cpp is rebuilt too so that
# hipbfv_build_flags() names the definitions (bench.py's kernel-source hash then refuses the default build's PMC figures).
set -e
TAG=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
make -s -C $ROOT/sunscreen_amd/csrc
mkdir -p $ROOT/sunscreen_amd/lib/variants $ROOT/build/variants
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950"
/opt/rocm/bin/hipcc $BASE "$@" -c $ROOT/sunscreen_amd/csrc/kernels_split.hip -o $ROOT/build/variants/split_$TAG.o &
# context.cpp plans the FP64 reduce masks for the same pass structure (nttshape.hpp): it must see the same macros
/opt/rocm/bin/hipcc $BASE "$@" -x hip -c $ROOT/sunscreen_amd/csrc/context.cpp -o $ROOT/build/variants/context_$TAG.o &
/opt/rocm/bin/hipcc $BASE "$@" -DHIPBFV_BUILD_FLAGS="\"variant $TAG: $*\"" -x hip -c $ROOT/sunscreen_amd/csrc/capi.cpp -o $ROOT/build/variants/capi_$TAG.o &
wait
OBJS=$(ls $ROOT/build/hipbfv_*.o | grep -v "kernels_split\|hipbfv_context\|hipbfv_capi")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$ROOT/sunscreen_amd/csrc/exports.map -o $ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so $OBJS $ROOT/build/variants/split_$TAG.o $ROOT/build/variants/context_$TAG.o $ROOT/build/variants/capi_$TAG.o -ldl
echo built $TAG
