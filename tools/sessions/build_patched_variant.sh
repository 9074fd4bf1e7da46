#!/bin/bash
# tools/build_patched_variant.sh <tag> <patch file> [-D flags...] -- sunscreen_amd/lib/variants/libhipbfv_<tag>.so from a PATCHED COPY of
# sunscreen_amd/csrc/ (build/variants/src_<tag>/): the tree -- and with it bench.kernel_source_hash() and the committed PMC figures -- stays as it is.
# Every translation unit is recompiled (a patch may touch a header); hipbfv_build_flags() of the variant names the patch, so the PMC-derived
# fields of a bench line refuse it.  Select with HIPBFV_LIB=<path> (tools/ab_libs.sh does).
set -e
TAG=$1; PATCH=$(readlink -f $2); shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
TOP=$ROOT/build/variants/src_$TAG
SRC=$TOP/sunscreen_amd/csrc   # same depth as the tree: capi.cpp includes ../../include/hipbfv.h
rm -rf $TOP; mkdir -p $SRC $ROOT/sunscreen_amd/lib/variants
ln -s $ROOT/include $TOP/include
cp $ROOT/sunscreen_amd/csrc/*.hip $ROOT/sunscreen_amd/csrc/*.hpp $ROOT/sunscreen_amd/csrc/*.cpp $ROOT/sunscreen_amd/csrc/exports.map $ROOT/sunscreen_amd/csrc/Makefile $SRC/
( cd $TOP && patch -p1 < $PATCH )
BASE="-O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950"
OBJS=""
FLAGSTR="patched variant $TAG: $(basename $PATCH) $*"
for f in $SRC/*.hip $SRC/*.cpp; do
  o=$SRC/$(basename $f).o; OBJS="$OBJS $o"
  case $f in
    */capi.cpp) /opt/rocm/bin/hipcc $BASE "$@" -DHIPBFV_BUILD_FLAGS="\"$FLAGSTR\"" -x hip -c $f -o $o & ;;
    *.cpp) /opt/rocm/bin/hipcc $BASE "$@" -x hip -c $f -o $o & ;;
    *) /opt/rocm/bin/hipcc $BASE "$@" -c $f -o $o & ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$SRC/exports.map -o $ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so $OBJS -ldl
echo built $TAG
