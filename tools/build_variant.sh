#!/bin/bash
# tools/build_variant.sh <tag> <-D flags...> -- build sunscreen_amd/lib/variants/libhipbfv_<tag>.so with extra
# macro definitions for the split kernels (tuning A/B runs; select it with HIPBFV_LIB=<path>).
set -e
TAG=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd)
make -s -C $ROOT/sunscreen_amd/csrc
mkdir -p $ROOT/sunscreen_amd/lib/variants $ROOT/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 "$@" -c $ROOT/sunscreen_amd/csrc/kernels_split.hip -o $ROOT/build/variants/split_$TAG.o
# context.cpp plans the FP64 reduce masks for the same pass structure (nttshape.hpp): it must see the same macros
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 "$@" -x hip -c $ROOT/sunscreen_amd/csrc/context.cpp -o $ROOT/build/variants/context_$TAG.o
OBJS=$(ls $ROOT/build/hipbfv_*.o | grep -v "kernels_split\|hipbfv_context")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$ROOT/sunscreen_amd/csrc/exports.map -o $ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so $OBJS $ROOT/build/variants/split_$TAG.o $ROOT/build/variants/context_$TAG.o -ldl
echo built $TAG
