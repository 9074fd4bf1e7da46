#!/bin/bash
# tools/ab_ntt.sh -- interleaved A/B of the stand-alone transform workloads: default library vs sunscreen_amd/lib/variants/libhipbfv_<tag>.so
TAG=${1:-nowp}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_ntt_$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_client.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for round in 1 2; do
  for arm in base $TAG; do
    if [ $arm = base ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so; fi
    for w in "n8192 --workload ntt" "n16384 --workload ntt --n 16384 --batch 2048" "bits54 --workload ntt --coeff-bits 54,54,54,56"; do
      set -- $w; name=$1; shift
      timeout 300 python bench.py "$@" --steps 10 --warmup 2 --no-cpu --no-secondary 2>/dev/null | tail -1 > $OUT/${arm}_${name}_$round.json
      python -c "
import json; d=json.load(open('$OUT/${arm}_${name}_$round.json')); print('$arm $name', d['value'], d['parity'][:24], d['kernels_ms_per_step'])"
    done
  done
done
