#!/bin/bash
# tools/pmc_ntt.sh -- PMC passes for the NTT workload (run on the GPU box through gpurun).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; mkdir -p $OUT
CMD="python bench.py --workload ${2:-ntt} --steps 2 --warmup 1 --no-cpu --no-check"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT TCC_MISS -d $OUT -o p4 -- $CMD > $OUT/p4.log 2>&1
ls $OUT
