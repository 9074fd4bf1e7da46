#!/bin/bash
# tools/gpu_extra_report.sh <tag> -- the secondary measurements of DESIGN.md section 6 (run after gpu_round_report.sh has
# refreshed profiles/pmc_traffic.json): headline line regenerated against the current PMC file, the sizes and workloads the
# main report does not cover, drop-in latency, and kernel-trace summaries of the n=16384, NTT and e2e workloads.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 400 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n8192.json
timeout 300 python bench.py --workload pir --batch 256 --steps 5 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_pir_n8192.json
timeout 400 python bench.py --n 32768 --batch 256 --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n32768.json
timeout 300 python bench.py --workload ntt --coeff-bits 54,54,54,56 --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_ntt_n8192_3x54bit.json
timeout 300 python tools/latency.py > $OUT/latency_n8192.json 2>$OUT/latency.err
for w in "mulrelin_n16384 --n 16384 --batch 1024" "ntt_n8192 --workload ntt" "e2e_n8192 --workload e2e --batch 2048"; do
  set -- $w; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_$name -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-secondary > $OUT/trace_$name.log 2>&1
  python tools/rocprof_summary.py $OUT/trace_${name}_results.db > $OUT/${name}_kernel_stats.txt 2>/dev/null
  rm -f $OUT/*.db
done
ls -la $OUT
