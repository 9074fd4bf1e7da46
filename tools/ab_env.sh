#!/bin/bash
# tools/ab_env.sh <ENVVAR> <bench args...> -- interleaved A/B on one box: default vs ENVVAR=1 (two rounds)
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/ab_$VAR; mkdir -p $OUT
for round in 1 2; do
  for arm in on off; do
    if [ $arm = off ]; then export $VAR=1; else unset $VAR; fi
    timeout 300 python bench.py "$@" --no-cpu --no-secondary 2>/dev/null | tail -1 > $OUT/${arm}_$round.json
    python -c "
import json; d=json.load(open('$OUT/${arm}_$round.json')); print('$VAR', '$arm', d['value'], d['parity'][:24], d['kernels_ms_per_step'])"
  done
done
