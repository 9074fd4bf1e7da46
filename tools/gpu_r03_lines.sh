#!/bin/bash
# tools/gpu_r03_lines.sh <tag> -- after tools/pmc_merge.sh: the bench lines of the four PMC workloads (their `traffic` / `valu` fields now
# come from passes taken on these very kernel sources) and the default line with its `secondary` object
TAG=${1:-r03_v7}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout 600 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'valu', d['valu'] and d['valu']['frac'])"; }
run default --steps 20 --warmup 3 --power
run default_k5 --power
run sustained_n8192 --steps 400 --warmup 2 --no-secondary --no-cpu --power
run mulrelin_n8192 --steps 10 --warmup 2 --no-secondary
run mulrelin_n16384 --n 16384 --batch 1024 --steps 5 --warmup 1 --power
run ntt_n8192 --workload ntt --steps 100 --warmup 3 --power
run ntt_n8192_sustained --workload ntt --steps 5000 --warmup 3 --no-cpu --power
run mulrelin_n8192_bits54-54-54-56 --coeff-bits 54,54,54,56 --steps 5 --warmup 2
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT -o default_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/$OUT/default_trace.log 2>&1; cd $GRAFT_REPO_ROOT; python tools/rocprof_summary.py $OUT/default_trace_results.db > $OUT/default_kernel_stats.txt 2>/dev/null; rm -f $OUT/*.db; head -12 $OUT/default_kernel_stats.txt
