#!/bin/bash
# tools/gpu_variant_suites.sh -- the whole GPU suite under each environment switch that selects another code path (three suites at a
# time on the one device: the tests are small, the device is mostly idle under a single pytest process; the three start 25 s apart
# so that they do not reach the one large-batch test together -- three copies of it at once ran the device out of memory once)
# VARIANTS / TESTS in the environment select a subset of the switches / of the test files
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/variants; mkdir -p $OUT
run() {
  kv=$1
  env $kv timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -q -p no:cacheprovider > $OUT/pytest_$kv.log 2>&1
  echo "$kv: $(tail -1 $OUT/pytest_$kv.log)"
}
VARIANTS=${VARIANTS:-"HIPBFV_NO_F64=1 HIPBFV_SEAL_AUX=1 HIPBFV_NO_PACK=1 HIPBFV_NO_PACK=ks HIPBFV_PACK_ROWS=0 HIPBFV_NO_FUSED_TAIL=1 HIPBFV_NO_SMALL_BATCH=0 HIPBFV_NO_FUSED_GALOIS=1 HIPBFV_NO_MEMBER_TAILS=1"}
n=0
for v in $VARIANTS; do
  run $v &
  n=$((n + 1))
  if [ $((n % 3)) = 0 ]; then wait; else sleep 25; fi
done
wait
