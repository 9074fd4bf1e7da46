#!/bin/bash
TAG=${1:-r03_d}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_gpu_program.py tests/test_gpu_fuzz.py -m gpu -x -q ) > $OUT/pytest_graph.txt 2>&1; tail -3 $OUT/pytest_graph.txt
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], round(sum(d['kernels_ms_per_step'].values()),3), d['kernels_ms_per_step'])" || tail -5 $OUT/$name.err; }
run pir_n8192_graph --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu
run pir_n8192_direct --workload pir --batch 256 --steps 5 --warmup 2 --no-cpu --pir-direct
run pir_n16384_2p17_graph --workload pir --n 16384 --batch 256 --pir-rows 512 --steps 3 --warmup 1 --no-cpu
run pir_n16384_2p17_direct --workload pir --n 16384 --batch 256 --pir-rows 512 --steps 3 --warmup 1 --no-cpu --pir-direct
