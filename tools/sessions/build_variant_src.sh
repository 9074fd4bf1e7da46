#!/bin/bash
# tools/build_variant_src.sh <tag> <source-stem> <-D flags...> -- like build_variant.sh, for macros of ONE other translation
# unit (kernels | kernels_client | evaluator ...): sunscreen_amd/lib/variants/libhipbfv_<tag>.so = the default objects with
# <source-stem> recompiled under the given definitions.
set -e
TAG=$1; STEM=$2; shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
make -s -C $ROOT/sunscreen_amd/csrc
mkdir -p $ROOT/sunscreen_amd/lib/variants $ROOT/build/variants
SRC=$ROOT/sunscreen_amd/csrc/$STEM.hip; X=""
[ -f $SRC ] || { SRC=$ROOT/sunscreen_amd/csrc/$STEM.cpp; X="-x hip"; }
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 "$@" $X -c $SRC -o $ROOT/build/variants/${STEM}_$TAG.o
# capi.cpp too: hipbfv_build_flags() must name the variant (bench.py's kernel-source hash)
if [ "$STEM" != capi ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950 -DHIPBFV_BUILD_FLAGS="\"variant $TAG ($STEM): $*\"" -x hip -c $ROOT/sunscreen_amd/csrc/capi.cpp -o $ROOT/build/variants/capi_$TAG.o
  CAPI=$ROOT/build/variants/capi_$TAG.o
fi
OBJS=$(ls $ROOT/build/hipbfv_*.o | grep -v "hipbfv_$STEM.o" | grep -v "hipbfv_capi.o")
[ "$STEM" = capi ] || OBJS="$OBJS $CAPI"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$ROOT/sunscreen_amd/csrc/exports.map -o $ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so $OBJS $ROOT/build/variants/${STEM}_$TAG.o -ldl
echo built $TAG
