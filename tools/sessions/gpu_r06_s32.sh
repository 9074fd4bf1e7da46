#!/bin/bash
# tools/sessions/gpu_r06_s32.sh -- a rotation's head / tail with the workgroups of one gathered row on one XCD: the rotation tests, then dot_prod
# and the rotation-only batch primitive against the x-fastest order (libhipbfv_xcdrows0.so), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_s32; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_per_key.py tests/test_gpu_program.py tests/test_gpu_fuzz.py -m gpu -x -q -k "rot or galois or dot or fuzz or every_operation or config5 or keys" 2>&1 | tail -4 | tee $OUT/pytest.txt
bash tools/ab_libs.sh xcdrows0 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_dot_prod_256.txt
bash tools/ab_libs.sh xcdrows0 --workload dot_prod --n 16384 --batch 128 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_dot_prod_128.txt
bash tools/ab_libs.sh xcdrows0 --workload dot_prod --n 8192 --batch 1024 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_dot_prod_n8192.txt
