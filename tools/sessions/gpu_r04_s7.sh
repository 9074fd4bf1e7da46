#!/bin/bash
# round 4, GPU session 7: base-conversion constants requested in one batch per trip (new) vs the previous commit (prev3); NT stores in mul_head
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_baseline_configs.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/ab_libs.sh "prev3 headnt" --steps 10 --warmup 3 --repeats 3 > $O/ab_n8192.txt 2>&1
cat $O/ab_n8192.txt
bash tools/ab_libs.sh "prev3" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_n16384.txt 2>&1
cat $O/ab_n16384.txt
