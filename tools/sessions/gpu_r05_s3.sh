#!/bin/bash
# round 5, session 3: on top of session 2's range-plan-aware stores -- biased 16-bit plane of the packed rows, no double reduction of the
# K = 4 / 8 key-switch accumulators, the product's residues handed to the fused kernels as doubles, the exact-sum extension in the
# 4-prime head (its own instantiation).  Whole suite, switch suites, then interleaved A/B: default vs session 2 (planstore) vs HEAD~ (fp64md).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s3; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="HIPBFV_NO_PACK=1 HIPBFV_NO_FUSED_TAIL=1 HIPBFV_NO_FUSED_HEAD=1 HIPBFV_SEAL_AUX=1 HIPBFV_NO_GRID=1 HIPBFV_PACK_ROWS=1" bash tools/gpu_variant_suites.sh > $O/variants.txt 2>&1; cat $O/variants.txt
bash tools/ab_libs.sh "planstore fp64md" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "planstore" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "planstore" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
bash tools/ab_libs.sh "planstore" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
