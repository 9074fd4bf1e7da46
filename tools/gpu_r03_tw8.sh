#!/bin/bash
# 8-byte FP64 twiddles against the (W, W/q) pairs of the previous commit (variants/libhipbfv_prev.so), interleaved
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab_libs.sh "prev" --workload ntt --steps 20 --warmup 3
bash tools/ab_libs.sh "prev" --steps 10 --warmup 2
bash tools/ab_libs.sh "prev" --n 16384 --batch 1024 --steps 5 --warmup 2
bash tools/ab_libs.sh "prev" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1
bash tools/ab_libs.sh "prev" --workload pir --batch 256 --steps 5 --warmup 2
