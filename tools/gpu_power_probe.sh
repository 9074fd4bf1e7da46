#!/bin/bash
# power / clock samples while a workload runs: $1 = tag for the output, rest = bench.py arguments
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out
( timeout 120 python bench.py "$@" --no-cpu --no-check --no-secondary > /tmp/pp_line.json 2>/dev/null ) &
BP=$!
sleep 14
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -i "power\|sclk\|mclk\|busy\|fclk" | tr '\n' ';'
  echo
  sleep 1
done > gpurun_out/power_$TAG.txt
wait $BP
tail -1 /tmp/pp_line.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$TAG', d['value'], d['kernels_ms_per_step'])" >> gpurun_out/power_$TAG.txt
cat gpurun_out/power_$TAG.txt
