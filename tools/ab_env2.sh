#!/bin/bash
# tools/ab_env2.sh "<VAR=value> ..." <bench args...> -- like ab_env.sh for ANY workload (prints value + the three largest kernels)
ARMS=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for arm in default $ARMS; do
    if [ $arm = default ]; then E=""; else E=$arm; fi
    env $E timeout 600 python bench.py "$@" --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 > /tmp/ab_line.json
    python -c "
import json; d=json.load(open('/tmp/ab_line.json')); print('$arm', d['value'], d.get('spread'), d['parity'][:20], dict(list(d['kernels_ms_per_step'].items())[:5]))"
  done
done
