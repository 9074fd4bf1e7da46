#!/bin/bash
# tools/gpu_r03_graph.sh <tag> -- on the GPU box: the graph executor's tests, the PIR workload through the compiled graph
# vs the hand-written primitives, chi_sq / dot_prod regression lines, single-input-set program latency.
TAG=${1:-r03_b}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 1200 python -m pytest tests/test_gpu_program.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py -m gpu -x -q ) > $OUT/pytest_graph.txt 2>&1; tail -15 $OUT/pytest_graph.txt
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], d['kernels_ms_per_step'], d['parity'][:60])" || tail -5 $OUT/$name.err; }
run pir_n8192_graph --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu
run pir_n8192_direct --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu --pir-direct
run chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
run dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
HIPBFV_PROGRAM_SERIAL=1 run chi_sq_n16384_serial --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
HIPBFV_PROGRAM_SERIAL=1 run dot_prod_n16384_serial --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
timeout 600 python tools/program_latency.py 16384 > $OUT/program_latency_n16384.json 2>$OUT/program_latency.err; cat $OUT/program_latency_n16384.json; tail -3 $OUT/program_latency.err
timeout 600 python tools/program_latency.py 8192 > $OUT/program_latency_n8192.json 2>>$OUT/program_latency.err; cat $OUT/program_latency_n8192.json
