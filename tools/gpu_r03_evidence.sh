#!/bin/bash
# tools/gpu_r03_evidence.sh <tag> -- on the GPU box: the round's evidence set in one call: PMC passes + kernel traces of the four
# PMC workloads (fresh source hash), the GPU suite, smoke, the default bench line (with `secondary`), every workload's line, the
# single-input-set program latency, the drop-in latency, the hipGraph probe.  Afterwards, here: tools/pmc_merge.sh <tag> <key> x4.
TAG=${1:-r03_v1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
PMC_TAG=$TAG bash tools/gpu_pmc_all.sh > $OUT/pmc_all.log 2>&1; tail -2 $OUT/pmc_all.log
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
run() { name=$1; shift; timeout 900 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], (d['roofline'] or {}).get('frac'), d['cpu_baseline'] and d['cpu_baseline']['value'])" || tail -5 $OUT/$name.err; }
( time python bench.py --steps 20 --warmup 3 ) > $OUT/bench_default.txt 2>$OUT/bench_default.err; grep '^{' $OUT/bench_default.txt | tail -1 > $OUT/bench_default.json; tail -3 $OUT/bench_default.err
run mulrelin_n4096 --n 4096 --batch 8192 --steps 5 --warmup 2
run mulrelin_n32768 --n 32768 --batch 256 --steps 2 --warmup 1
run ntt_n16384 --workload ntt --n 16384 --batch 2048 --steps 50 --warmup 2
run ntt_n8192_bits54-54-54-56 --workload ntt --coeff-bits 54,54,54,56 --steps 100 --warmup 2
run chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1
run dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1
run e2e_n8192 --workload e2e --batch 2048 --steps 5 --warmup 2
run pir_n8192 --workload pir --batch 256 --steps 5 --warmup 1
run pir_n8192_direct --workload pir --batch 256 --steps 5 --warmup 1 --pir-direct
run pir_n16384_2p17 --workload pir --n 16384 --batch 256 --pir-rows 512 --steps 3 --warmup 1
timeout 300 python tools/latency.py > $OUT/latency_n8192.json 2>$OUT/latency.err; cat $OUT/latency_n8192.json | cut -c1-300
timeout 600 python tools/program_latency.py 16384 > $OUT/program_latency_n16384.json 2>$OUT/program_latency.err; cat $OUT/program_latency_n16384.json
timeout 600 python tools/program_latency.py 8192 > $OUT/program_latency_n8192.json 2>>$OUT/program_latency.err
( python tools/graph_probe.py 8192; python tools/graph_probe.py 16384 ) > $OUT/graph_probe.txt 2>/dev/null; cat $OUT/graph_probe.txt
python tools/two_stream_probe.py 16384 1024 2>/dev/null | tail -1 > $OUT/two_stream_probe.txt; cat $OUT/two_stream_probe.txt
ls $OUT | wc -l
