#!/bin/bash
# tools/sessions/gpu_r06_s24.sh -- an extended parameter-fuzz campaign on the final round-6 build (tests/test_gpu_fuzz.py with other seeds:
# 2 x 150 parameter sets, every evaluator operation against the oracle; the suite's own 40 sets use the default seed)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_s24; mkdir -p $OUT
for seed in ${SEEDS:-611 612}; do
  HIPBFV_FUZZ_SEED=$seed HIPBFV_FUZZ_COUNT=${COUNT:-150} timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4 > $OUT/fuzz_seed$seed.txt
  echo "seed $seed: $(tail -1 $OUT/fuzz_seed$seed.txt)"
done | tee $OUT/fuzz_campaign.txt
