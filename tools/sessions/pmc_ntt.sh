#!/bin/bash
# tools/pmc_ntt.sh <tag> [bench args] -- SQ-level PMC passes (wave occupancy, wait / issue breakdown, LDS) for a workload
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
CMD="python bench.py ${*:---workload ntt} --steps 2 --warmup 1 --no-cpu --no-check --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT GRBM_GUI_ACTIVE -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
for p in p1 p2 p3; do python tools/rocprof_summary.py $OUT/${p}_results.db --pmc 2>/dev/null | grep -A12 "hipbfv::" > $OUT/$p.txt; done
rm -f $OUT/*.db
cat $OUT/p1.txt $OUT/p2.txt $OUT/p3.txt
