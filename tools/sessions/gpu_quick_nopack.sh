#!/bin/bash
# one interleaved round, default vs HIPBFV_NO_PACK=1 at the headline configuration (5 steps, 3 regions): a hint, not a measurement
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for arm in ${ARMS:-default HIPBFV_NO_PACK=1 default HIPBFV_NO_PACK=1}; do
  if [ $arm = default ]; then E=""; else E=$arm; fi
  env $E timeout 60 python bench.py --steps 5 --warmup 2 --repeats 3 --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$arm', d['value'], d.get('spread'), d['parity'][:9], d['kernels_ms_per_step'])"
done
