#!/bin/bash
# like ab_run.sh for the n=16384 configuration
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for t in "$@"; do
  HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$t.so python bench.py --n 16384 --batch 1024 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['parity'][:20], d['kernels_ms_per_step'])"
done
