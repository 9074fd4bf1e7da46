#!/bin/bash
# tools/gpu_check.sh <tag> -- on the GPU box: the GPU suite, smoke, the default bench line.  Output under gpurun_out/<tag>/.
TAG=${1:-check}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 400 python bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_mulrelin_n8192.json; cat $OUT/bench_mulrelin_n8192.json; tail -3 $OUT/bench.err
