#!/bin/bash
# tools/sessions/gpu_r06_s31.sh -- the PIR product with every slice's row blocks on one XCD: the tests that run the product kernels, then three
# database shapes of 2^17 entries (rows x columns: 512 x 256, 256 x 512, 128 x 1024 = one GPU's rows of a 1024 x 1024 database) against
# the x-fastest order (libhipbfv_xcd0.so), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_s31; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_program.py tests/test_gpu_baseline_configs.py tests/test_gpu_dist.py -m gpu -x -q -k "pir or plain or matrix" 2>&1 | tail -4 | tee $OUT/pytest.txt
export HIPBFV_LIB_ALLOW_MISSING=1
for round in 1 2; do
  for shape in "256 512" "512 256" "1024 128"; do
    set -- $shape
    for arm in new xcd0; do
      if [ $arm = new ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$arm.so; fi
      timeout 600 python bench.py --workload pir --n 16384 --batch $1 --pir-rows $2 --steps 3 --warmup 1 --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 > /tmp/ab_line.json
      python -c "
import json; d=json.load(open('/tmp/ab_line.json')); print('$arm rows $2 x cols $1', d['value'], d['ms_per_step'], d['parity'][:30], dict(list(d['kernels_ms_per_step'].items())[:3]))"
    done
  done
done 2>&1 | tee $OUT/ab_pir_xcd_slices.txt
