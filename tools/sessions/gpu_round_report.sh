#!/bin/bash
# tools/gpu_round_report.sh <tag> -- run on the GPU box (through gpurun): tests, smoke, an interleaved A/B of the
# base-conversion forms (HIPBFV_NO_GRID), benches and rocprofv3 kernel-trace + PMC passes for the headline workload;
# everything lands under gpurun_out/<tag>/.  Steps are ordered by importance: the call may be cut short by the budget.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# A/B at n=16384 (where the 8-prime tail uses them): grid sums (default) vs per-term reduction, interleaved on this box
for round in 1 2; do
  for arm in grid nogrid; do
    if [ $arm = nogrid ]; then export HIPBFV_NO_GRID=1; else unset HIPBFV_NO_GRID; fi
    timeout 300 python bench.py --n 16384 --batch 1024 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $OUT/ab_${arm}_n16384_$round.json
  done
done
unset HIPBFV_NO_GRID
python - <<PY | tee $OUT/ab_summary.txt
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["parity"][:24], d["kernels_ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
timeout 400 python bench.py --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n8192.json
CMD="python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
python tools/rocprof_summary.py $OUT/trace_results.db > $OUT/kernel_stats.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT -o pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT TCC_MISS -d $OUT -o pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT -o pmc_inst -- $CMD > $OUT/pmc_inst.log 2>&1
for p in pmc_fetch pmc_write pmc_inst; do python tools/rocprof_summary.py $OUT/${p}_results.db --pmc > $OUT/$p.txt 2>/dev/null; done
rm -f $OUT/*.db
timeout 400 python bench.py --n 16384 --batch 1024 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n16384.json
timeout 300 python bench.py --workload ntt --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_ntt_n8192.json
timeout 300 python bench.py --workload ntt --n 16384 --batch 2048 --steps 10 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_ntt_n16384.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT -o pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1
python tools/rocprof_summary.py $OUT/pmc_sq_results.db --pmc > $OUT/pmc_sq.txt 2>/dev/null
rm -f $OUT/*.db
# the other workloads and sizes
timeout 300 python bench.py --n 4096 --batch 8192 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n4096.json
timeout 300 python bench.py --coeff-bits 54,54,54,56 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_mulrelin_n8192_3x54bit.json
timeout 300 python bench.py --workload e2e --batch 2048 --steps 5 --warmup 2 2>/dev/null | tail -1 > $OUT/bench_e2e_n8192.json
timeout 300 python bench.py --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_chi_sq_n16384.json
timeout 300 python bench.py --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_dot_prod_n16384.json
HIPBFV_NO_GRID=1 timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_no_grid.log 2>&1; tail -1 $OUT/pytest_gpu_no_grid.log
ls $OUT; du -sh $OUT
