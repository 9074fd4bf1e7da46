#!/bin/bash
# tools/sessions/gpu_r06_s26.sh -- the sums around fused products inside the products' last kernel (Plan::LinFold, MemberTail): the program
# tests, then chi_sq interleaved against HIPBFV_NO_MEMBER_TAILS=1 at the three batch sizes of the bench lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_s26; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_program.py tests/test_gpu_per_key.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.txt
cat $OUT/pytest.txt
bash tools/ab_env2.sh "HIPBFV_NO_MEMBER_TAILS=1" --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_chi_sq_256.txt
bash tools/ab_env2.sh "HIPBFV_NO_MEMBER_TAILS=1" --workload chi_sq --n 16384 --batch 128 --steps 3 --warmup 1 2>&1 | tee $OUT/ab_chi_sq_128.txt
bash tools/ab_env2.sh "HIPBFV_NO_MEMBER_TAILS=1" --workload chi_sq --n 16384 --batch 1024 --steps 2 --warmup 1 2>&1 | tee $OUT/ab_chi_sq_1024.txt
# s27: merged product launches at every batch size (MemberHead tables), against a launch per product
for b in 128 256 1024; do
  bash tools/ab_env2.sh "HIPBFV_NO_MERGED_PRODUCTS=1" --workload chi_sq --n 16384 --batch $b --steps 3 --warmup 1 2>&1 | tee $OUT/ab_merged_chi_sq_$b.txt
done
bash tools/ab_libs.sh prev --steps 10 --warmup 2 2>&1 | tee $OUT/ab_prev_headline.txt
