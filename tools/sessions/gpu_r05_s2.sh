#!/bin/bash
# round 5, session 2: the range-plan-aware stores (no reduction in front of packed stores / after the tail's scaling where the plan
# shows it is not needed; tight entry bound of the inverse) -- whole GPU suite, the switch suites that change which rows are packed
# or which kernels are fused, then interleaved A/B against the library of the previous commit (variant fp64md = HEAD~ arithmetic)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s2; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="HIPBFV_NO_PACK=1 HIPBFV_NO_FUSED_TAIL=1 HIPBFV_NO_FUSED_HEAD=1" bash tools/gpu_variant_suites.sh > $O/variants.txt 2>&1; cat $O/variants.txt
bash tools/ab_libs.sh "fp64md" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "fp64md" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "fp64md" --coeff-bits 54,54,54,56 --steps 5 --warmup 2 --repeats 3 > $O/ab_3x54.txt 2>&1; cat $O/ab_3x54.txt
bash tools/ab_libs.sh "fp64md" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
