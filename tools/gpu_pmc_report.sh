#!/bin/bash
# tools/gpu_pmc_report.sh <tag> <workload-key> <bench.py args...> -- on the GPU box (through gpurun): the bench line, the
# rocprofv3 kernel-trace summary and three PMC passes (fetch + clock, write, VALU instruction classes) of ONE workload;
# everything under gpurun_out/<tag>/<workload-key>_*.  Afterwards (here, in the container):
#   tools/pmc_merge.sh <tag> <workload-key>    -> profiles/<tag>_<key>_*.txt + an entry in profiles/pmc_traffic.json
# PMC passes never combine with --sys-trace / hip / hsa traces (gpurun refuses that combination).
TAG=$1; KEY=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import bench; print(bench.kernel_source_hash())" > $OUT/${KEY}_source_hash.txt
timeout 400 python bench.py "$@" --no-secondary 2>$OUT/${KEY}_bench.err | tail -1 > $OUT/${KEY}_bench.json
CMD="python bench.py $* --steps 3 --warmup 1 --repeats 1 --settle-ms 0 --no-cpu --no-check --no-secondary --no-power"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o ${KEY}_trace -- $CMD > $OUT/${KEY}_trace.log 2>&1
python tools/rocprof_summary.py $OUT/${KEY}_trace_results.db > $OUT/${KEY}_kernel_stats.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT -o ${KEY}_pmc_fetch -- $CMD > $OUT/${KEY}_pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT TCC_MISS -d $OUT -o ${KEY}_pmc_write -- $CMD > $OUT/${KEY}_pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT -o ${KEY}_pmc_inst -- $CMD > $OUT/${KEY}_pmc_inst.log 2>&1
for p in pmc_fetch pmc_write pmc_inst; do python tools/rocprof_summary.py $OUT/${KEY}_${p}_results.db --pmc > $OUT/${KEY}_$p.txt 2>/dev/null; done
rm -f $OUT/*.db
ls -la $OUT | grep $KEY
