#!/bin/bash
# tools/gpu_variant_suites.sh -- the whole GPU suite under each environment switch that selects another code path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/variants; mkdir -p $OUT
for v in HIPBFV_NO_F64 HIPBFV_SEAL_AUX HIPBFV_NO_FUSED_TAIL HIPBFV_NO_FUSED_HEAD HIPBFV_NO_PACK HIPBFV_NO_SPLIT_KS_INT HIPBFV_NO_GRID HIPBFV_NO_SQUARE HIPBFV_NO_FUSED_PLAIN "HIPBFV_NO_SMALL_BATCH=0"; do
  case $v in *=*) kv=$v;; *) kv=$v=1;; esac
  env $kv timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_$kv.log 2>&1
  echo "$kv: $(tail -1 $OUT/pytest_$kv.log)"
done
