#!/bin/bash
# round 6 evidence, step 1 (GPU): on the FINAL sources -- the whole suite, the nine switch suites, PMC passes + kernel traces of the seven
# PMC workloads (chi_sq and dot_prod included: VERDICT r04 #5), the stall / LDS passes of the three metric workloads, and the kernel
# trace of the driver's default command.  Step 2 (container): tools/pmc_merge.sh r06_final <key> per key -> profiles/pmc_traffic.json.
# Step 3 (GPU): tools/gpu_r06_final2.sh.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r06_final
O=gpurun_out/$T; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log
bash tools/gpu_variant_suites.sh > $O/variant_suites.txt 2>&1; cat $O/variant_suites.txt
bash tools/gpu_pmc_report.sh $T mulrelin_n8192 --steps 5 --warmup 2 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T mulrelin_n16384 --n 16384 --batch 1024 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T ntt_n8192 --workload ntt > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T mulrelin_n8192_bits54-54-54-56 --coeff-bits 54,54,54,56 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T chi_sq_n16384 --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T dot_prod_n16384 --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T pir_n16384 --workload pir --n 16384 --batch 1024 --pir-rows 128 --steps 3 --warmup 1 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T mulrelin_n8192_keys4096 --keys 4096 > /dev/null 2>&1
bash tools/gpu_pmc_report.sh $T mulrelin_n16384_keys1024 --n 16384 --batch 1024 --keys 1024 > /dev/null 2>&1
bash tools/gpu_pmc_stall.sh $T mulrelin_n8192 > /dev/null 2>&1
bash tools/gpu_pmc_stall.sh $T mulrelin_n16384 --n 16384 --batch 1024 > /dev/null 2>&1
bash tools/gpu_pmc_stall.sh $T ntt_n8192 --workload ntt > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o default_bench_trace -- python bench.py --steps 20 --warmup 5 --no-cpu --no-power > $O/default_bench_trace.log 2>&1
python tools/rocprof_summary.py $O/default_bench_trace_results.db > $O/default_bench_kernel_stats.txt 2>/dev/null
rm -f $O/*.db
ls $O | wc -l; for k in mulrelin_n8192 mulrelin_n16384 ntt_n8192 mulrelin_n8192_bits54-54-54-56 chi_sq_n16384 dot_prod_n16384 pir_n16384 mulrelin_n8192_keys4096 mulrelin_n16384_keys1024; do python -c "
import json; d=json.load(open('$O/${k}_bench.json')); print('$k', d['value'], d['unit'])"; done
