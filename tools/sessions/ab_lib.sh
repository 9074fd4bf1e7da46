#!/bin/bash
# tools/ab_lib.sh <tag> <bench args...> -- interleaved A/B on one box: default library vs sunscreen_amd/lib/variants/libhipbfv_<tag>.so
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
for round in 1 2; do
  for arm in new $TAG; do
    if [ $arm = new ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$TAG.so; fi
    timeout 300 python bench.py "$@" --no-cpu --no-secondary 2>/dev/null | tail -1 > $OUT/${arm}_$round.json
    python -c "
import json; d=json.load(open('$OUT/${arm}_$round.json')); print('$arm', d['value'], d['parity'][:24], d['kernels_ms_per_step'])"
  done
done
