#!/bin/bash
# tools/pmc_sq.sh <tag> [HIPBFV_LIB path or ""] <bench args...> -- SQ-level PMC passes (VALU / wait breakdown, traffic) per kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=$1; LIB=$2; shift 2
if [ -n "$LIB" ]; then export HIPBFV_LIB=$GRAFT_REPO_ROOT/$LIB; fi
OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT
CMD="python bench.py $* --steps 2 --warmup 1 --no-cpu --no-check --no-secondary"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT -o p1 -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT -o p2 -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT TCC_MISS -d $OUT -o p3 -- $CMD > $OUT/p3.log 2>&1
for p in p1 p2 p3; do python tools/rocprof_summary.py $OUT/${p}_results.db --pmc 2>/dev/null | grep -A10 "hipbfv::" > $OUT/$p.txt; done
rm -f $OUT/*.db
