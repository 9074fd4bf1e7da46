#!/bin/bash
# tools/gpu_r06_s22.sh -- per-row packing of the key switch's rows (DevCtx::pack_ks == 2): parity first, then interleaved A/B against
# HIPBFV_NO_PACK=ks (the same library with 8-byte key-switch rows) on the three n = 16384 workloads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_s22; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_properties.py tests/test_gpu_baseline_configs.py tests/test_gpu_per_key.py -m gpu -x -q -k "per_row or 16384 or chi_sq or dot or keys" 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
bash tools/ab_env.sh "HIPBFV_NO_PACK=ks" --n 16384 --batch 1024 --steps 5 --warmup 1 2>&1 | tee $OUT/ab_mulrelin_n16384.txt
bash tools/ab_env2.sh "HIPBFV_NO_PACK=ks" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_dot_prod.txt
bash tools/ab_env2.sh "HIPBFV_NO_PACK=ks" --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_chi_sq.txt
