#!/bin/bash
# round 4, GPU session 10: PMC passes + kernel traces of the five PMC workloads, and the kernel trace of the driver's default command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PMC_TAG=r04_v1
O=gpurun_out/$PMC_TAG; mkdir -p $O
( time bash tools/gpu_pmc_all.sh ) > $O/pmc_all.log 2>&1
tail -30 $O/pmc_all.log
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o default_bench_trace -- python bench.py --steps 20 --warmup 5 --no-cpu --no-power > $O/default_bench_trace.log 2>&1
python tools/rocprof_summary.py $O/default_bench_trace_results.db > $O/default_bench_kernel_stats.txt 2>/dev/null
rm -f $O/*.db
head -30 $O/default_bench_kernel_stats.txt
