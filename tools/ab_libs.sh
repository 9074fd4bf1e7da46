#!/bin/bash
# tools/ab_libs.sh "<tag1> <tag2> ..." <bench args...> -- interleaved runs of the default library and several variant libraries
TAGS=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export HIPBFV_LIB_ALLOW_MISSING=1  # an arm may be a build of an older tree (entry points added since are absent there)
for round in 1 2; do
  for arm in new $TAGS; do
    if [ $arm = new ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$arm.so; fi
    timeout 300 python bench.py "$@" --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 > /tmp/ab_line.json
    python -c "
import json; d=json.load(open('/tmp/ab_line.json')); print('$arm', d['value'], d['parity'][:20], d['kernels_ms_per_step'])"
  done
done
