#!/bin/bash
# round 5, session 4: buffer addressing in the coefficient-parallel kernels (address arithmetic moved from the VALU to the scalar unit),
# optional operands as zero-record descriptors, no zeroing selects in mul_head.  Whole suite, switch suites, interleaved A/B vs session 3.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s4; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
VARIANTS="HIPBFV_NO_PACK=1 HIPBFV_NO_FUSED_TAIL=1 HIPBFV_NO_FUSED_HEAD=1 HIPBFV_SEAL_AUX=1 HIPBFV_NO_GRID=1 HIPBFV_PACK_ROWS=1" bash tools/gpu_variant_suites.sh > $O/variants.txt 2>&1; cat $O/variants.txt
bash tools/ab_libs.sh "s3" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "s3" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "s3" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
bash tools/ab_libs.sh "s3" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
bash tools/ab_libs.sh "s3" --n 4096 --steps 5 --warmup 2 --repeats 3 > $O/ab_n4096.txt 2>&1; cat $O/ab_n4096.txt
