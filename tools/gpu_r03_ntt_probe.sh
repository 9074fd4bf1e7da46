#!/bin/bash
# timing probes of the stand-alone transforms (wrong results by construction: --no-check)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for arm in new pr1 pr2 pr3 pr5 new pr1 pr2 pr3 pr5; do
  if [ $arm = new ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$arm.so; fi
  timeout 300 python bench.py --workload ntt --steps 20 --warmup 3 --no-cpu --no-check --no-secondary 2>/dev/null | tail -1 > /tmp/ab_line.json
  python -c "
import json; d=json.load(open('/tmp/ab_line.json')); print('$arm', d['value'], d['kernels_ms_per_step'])"
done
