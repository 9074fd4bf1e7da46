#!/bin/bash
TAG=${1:-r03_c}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_program.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $OUT/pytest_graph.txt 2>&1; tail -3 $OUT/pytest_graph.txt
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], sum(d['kernels_ms_per_step'].values()))" || tail -5 $OUT/$name.err; }
run pir_n8192_graph --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu
run pir_n8192_direct --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu --pir-direct
run chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
HIPBFV_PROGRAM_SERIAL=1 run chi_sq_n16384_serial --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
run dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --no-cpu
HIPBFV_PROGRAM_TRACE=1 timeout 600 python bench.py --workload chi_sq --n 16384 --batch 256 --steps 1 --warmup 1 --no-cpu --no-check 2>&1 | grep "^\[program\]" | tail -12
timeout 600 python tools/program_latency.py 16384 > $OUT/program_latency_n16384.json 2>$OUT/program_latency.err; cat $OUT/program_latency_n16384.json
