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
VARIANTS="HIPBFV_NO_PACK=1" bash tools/gpu_variant_suites.sh > $O/variant_suite_no_pack_rerun.txt 2>&1; cat $O/variant_suite_no_pack_rerun.txt
