#!/bin/bash
# tools/gpu_r03_check.sh <tag> -- on the GPU box: the GPU suite, smoke, the default bench line (with `secondary`) and the
# secondary workloads; everything under gpurun_out/<tag>/
TAG=${1:-r03_a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1; tail -5 $OUT/pytest_gpu.txt
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 600 python bench.py --steps 20 --warmup 3 ) > $OUT/bench_default.txt 2> $OUT/bench_default.err; tail -4 $OUT/bench_default.err
grep '^{' $OUT/bench_default.txt | tail -1 > $OUT/bench_default.json
python - <<PY
import json
d=json.load(open("$OUT/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["whole_op"]["frac"], d["roofline"]["traffic"], d["parity"][:40])
for k,v in d.get("secondary",{}).items(): print(k, v["value"], v["unit"], v["roofline"]["frac"], v["cpu_baseline"] and v["cpu_baseline"]["value"], v["parity"][:50])
PY
run() { name=$1; shift; timeout 600 python bench.py "$@" 2>$OUT/$name.err | tail -1 > $OUT/bench_$name.json; python -c "
import json; d=json.load(open('$OUT/bench_$name.json')); print('$name', d['value'], d['unit'], d['ms_per_step'], d['kernels_ms_per_step'], d['parity'][:60])" || tail -5 $OUT/$name.err; }
run chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1
run dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1
run pir_n8192 --workload pir --batch 256 --steps 5 --warmup 1
run pir_n16384_2p17 --workload pir --n 16384 --batch 256 --pir-rows 512 --steps 3 --warmup 1
