#!/bin/bash
# tools/gpu_pmc_stall.sh <tag> <workload-key> <bench.py args...> -- on the GPU box: two SQ-level PMC passes of ONE workload that say what
# the wavefronts of each kernel do with their cycles (MI355X_MICROARCH.md: WAIT_ANY = parked at s_waitcnt / a barrier, WAIT_INST_ANY =
# ready but not issued, ACTIVE_INST_ANY = issuing; the three are disjoint and add up to WAVE_CYCLES) and what the LDS exchanges cost
# (IDX_ACTIVE = LDS-array cycles, BANK_CONFLICT = the extra ones).  Summaries: gpurun_out/<tag>/<key>_pmc_stall.txt, <key>_pmc_lds.txt.
TAG=$1; KEY=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
CMD="python bench.py $* --steps 3 --warmup 1 --repeats 1 --settle-ms 0 --no-cpu --no-check --no-secondary --no-power"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $OUT -o ${KEY}_pmc_stall -- $CMD > $OUT/${KEY}_pmc_stall.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM_WR SQ_ACTIVE_INST_SCA -d $OUT -o ${KEY}_pmc_lds -- $CMD > $OUT/${KEY}_pmc_lds.log 2>&1
for p in pmc_stall pmc_lds; do python tools/rocprof_summary.py $OUT/${KEY}_${p}_results.db --pmc > $OUT/${KEY}_$p.txt 2>/dev/null; done
rm -f $OUT/*.db
ls -la $OUT | grep ${KEY}_pmc_
