#!/bin/bash
# round 5, session 8: optional operands without waterfall loops (the descriptor's record count formed by a scalar instruction): key-switch
# tests, then interleaved A/B against the previous build (variant `pre`) on the headline, n = 16384, dot_prod (13 rotations: ks_tail), chi_sq
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s8; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_program.py tests/test_gpu_baseline_configs.py tests/test_gpu_properties.py -m gpu -q ) > $O/pytest_subset.log 2>&1; grep -E "passed|failed" $O/pytest_subset.log
bash tools/ab_libs.sh "pre" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "pre" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "pre" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
bash tools/ab_libs.sh "pre" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
