#!/bin/bash
# round 4 evidence, step 1 (GPU): PMC passes + kernel traces of the five PMC workloads and the kernel trace of the driver's default command.
# step 2 (container): tools/pmc_merge.sh r04_v1 <key> for every key -> profiles/pmc_traffic.json.  step 3 (GPU): tools/gpu_r04_lines.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PMC_TAG=${1:-r04_v1}
O=gpurun_out/$PMC_TAG; mkdir -p $O
( time bash tools/gpu_pmc_all.sh ) > $O/pmc_all.log 2>&1
tail -3 $O/pmc_all.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o default_bench_trace -- python bench.py --steps 20 --warmup 5 --no-cpu --no-power > $O/default_bench_trace.log 2>&1
python tools/rocprof_summary.py $O/default_bench_trace_results.db > $O/default_bench_kernel_stats.txt 2>/dev/null
rm -f $O/*.db
head -12 $O/default_bench_kernel_stats.txt
