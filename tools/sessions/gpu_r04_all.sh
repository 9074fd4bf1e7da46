#!/bin/bash
# round 4 evidence in ONE box session (GPU minutes: every session pays a push): PMC passes -> merge on the box -> bench lines that read the
# merged profiles/pmc_traffic.json -> the switches' suites.  The box's profiles/ does not travel back: what was merged there is copied to
# gpurun_out/<tag>/profiles_box/ and the container re-runs tools/pmc_merge.sh on the same gpurun_out files (same bytes).
TAG=${1:-r04_v3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/gpu_r04_pmc.sh $TAG
for k in mulrelin_n8192 mulrelin_n16384 ntt_n8192 mulrelin_n8192_bits54-54-54-56 pir_n16384; do bash tools/pmc_merge.sh $TAG $k; done
mkdir -p gpurun_out/$TAG/profiles_box; cp profiles/pmc_traffic.json gpurun_out/$TAG/profiles_box/
SKIP_PYTEST=1 bash tools/gpu_r04_lines.sh $TAG
( time VARIANTS="HIPBFV_NO_FUSED_TAIL=1 HIPBFV_NO_FUSED_HEAD=1 HIPBFV_NO_PACK=1 HIPBFV_PACK_ROWS=1 HIPBFV_NO_SPLIT_KS_INT=1 HIPBFV_NO_GRID=1 HIPBFV_NO_SQUARE=1 HIPBFV_NO_FUSED_PLAIN=1 HIPBFV_NO_SMALL_BATCH=0" \
  TESTS="tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_rng.py" bash tools/gpu_variant_suites.sh ) 2>&1 | tee gpurun_out/$TAG/variants_rerun.txt
