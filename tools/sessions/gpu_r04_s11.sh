#!/bin/bash
# round 4, GPU session 11: mixed / integer tails with pipelined row loads (new) vs the previous commit (prev4) on the 3 x 54-bit set; parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s11; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_baseline_configs.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/ab_libs.sh "prev4" --coeff-bits 54,54,54,56 --steps 10 --warmup 3 --repeats 3 --check-items 8 > $O/ab_3x54.txt 2>&1
cat $O/ab_3x54.txt
bash tools/ab_libs.sh "prev4" --steps 10 --warmup 3 --repeats 3 > $O/ab_n8192.txt 2>&1
cat $O/ab_n8192.txt
