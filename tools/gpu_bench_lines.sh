#!/bin/bash
# tools/gpu_bench_lines.sh <tag> -- on the GPU box: the bench lines of the workloads profiles/pmc_traffic.json covers (so that their
# `traffic` / `valu` fields come from the PMC file of the same kernels) plus the secondary workloads; output gpurun_out/<tag>/bench_*.json
TAG=${1:-lines}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { name=$1; shift; timeout 400 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], 'roofline', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])"; }
run mulrelin_n8192 --steps 10 --warmup 2
run mulrelin_n16384 --n 16384 --batch 1024 --steps 5 --warmup 1
run ntt_n8192 --workload ntt --steps 10 --warmup 2
run mulrelin_n8192_bits54-54-54-56 --coeff-bits 54,54,54,56 --steps 5 --warmup 2
run mulrelin_n8192_keys4096 --keys 4096 --steps 5 --warmup 2
run mulrelin_n16384_keys1024 --n 16384 --batch 1024 --keys 1024 --steps 5 --warmup 1
if [ "$2" = all ]; then
run mulrelin_n8192_keys64 --keys 64 --steps 5 --warmup 2
run mulrelin_n4096 --n 4096 --batch 8192 --steps 5 --warmup 2
run mulrelin_n32768 --n 32768 --batch 256 --steps 2 --warmup 1
run ntt_n16384 --workload ntt --n 16384 --batch 2048 --steps 10 --warmup 2
run ntt_n8192_bits54-54-54-56 --workload ntt --coeff-bits 54,54,54,56 --steps 10 --warmup 2
run chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1
run dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1
run e2e_n8192 --workload e2e --batch 2048 --steps 5 --warmup 2
run pir_n8192 --workload pir --batch 256 --steps 5 --warmup 1
run pir_n16384 --workload pir --n 16384 --batch 1024 --pir-rows 128 --steps 3 --warmup 1
timeout 300 python tools/latency.py > $OUT/latency_n8192.json 2>$OUT/latency.err
fi
