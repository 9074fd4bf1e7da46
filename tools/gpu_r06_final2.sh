#!/bin/bash
# round 6 evidence, step 3 (GPU, after the PMC merge): the driver's default command (twice: one box's run-to-run spread), every other
# bench line against the merged profiles/pmc_traffic.json, single-call latencies, smoke
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06_final
O=gpurun_out/$T; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_default_run2.json 2> /dev/null
python - <<'PY'
import json
for f in ("bench_default.json", "bench_default_run2.json"):
    d = json.load(open("gpurun_out/r06_final/" + f)); print(f, len(json.dumps(d)), json.dumps(d["summary"]))
PY
bash tools/gpu_bench_lines.sh $T all
timeout 300 python tools/program_latency.py > $O/program_latency_n16384.json 2> $O/program_latency.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# the tests added after step 1's suite (the per-row pipelines down the modulus chain), under the default and the two packing switches
for e in HIPBFV_X=0 HIPBFV_NO_PACK=ks HIPBFV_PACK_ROWS=0; do
  echo "$e: $(env $e timeout 600 python -m pytest tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -k 'per_row' 2>&1 | tail -1)"
done > $O/pytest_per_row_chain.txt 2>&1; cat $O/pytest_per_row_chain.txt
