#!/bin/bash
# tools/gpu_trace_secondary.sh <tag> -- rocprofv3 kernel-trace summaries of the secondary workloads (the four PMC workloads have
# theirs from tools/gpu_pmc_report.sh): chi_sq, dot_prod, pir at both degrees, e2e -> gpurun_out/<tag>/<workload>_kernel_stats.txt
TAG=${1:-traces}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
for w in "chi_sq_n16384 --workload chi_sq --n 16384 --batch 256" "dot_prod_n16384 --workload dot_prod --n 16384 --batch 256" "pir_n8192 --workload pir --batch 256" "pir_n16384 --workload pir --n 16384 --batch 256" "e2e_n8192 --workload e2e --batch 2048"; do
  set -- $w; name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_$name -- python bench.py "$@" --steps 3 --warmup 1 --no-cpu --no-check --no-secondary > $OUT/trace_$name.log 2>&1
  python tools/rocprof_summary.py $OUT/trace_${name}_results.db > $OUT/${name}_kernel_stats.txt 2>/dev/null
  rm -f $OUT/*.db
  echo "== $name"; head -8 $OUT/${name}_kernel_stats.txt | cut -c1-130
done
