#!/bin/bash
# tools/gpu_final_evidence.sh <tag> -- on the GPU box: everything the round's evidence set holds, in one call:
# PMC passes + kernel traces of the four PMC workloads, the GPU suite log, the threaded drop-in numbers.
# (bench lines come afterwards, against the merged profiles/pmc_traffic.json: tools/gpu_bench_lines.sh <tag> all)
TAG=${1:-final}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash tools/gpu_pmc_all.sh > $OUT/pmc_all.log 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
g++ -O2 -std=c++17 -Iinclude tools/mt_dropin.cpp -Lsunscreen_amd/lib -lhipbfv -Wl,-rpath,$GRAFT_REPO_ROOT/sunscreen_amd/lib -lpthread -o /tmp/mt_dropin
( /tmp/mt_dropin 8192 1.5; HIPBFV_NO_COMBINE=1 /tmp/mt_dropin 8192 1.0; /tmp/mt_dropin 16384 1.5; /tmp/mt_dropin 16384 1.5 chi_sq; HIPBFV_NO_COMBINE=1 /tmp/mt_dropin 16384 1.0 chi_sq ) > $OUT/mt_dropin.txt 2>&1; cat $OUT/mt_dropin.txt
python tools/bench_multiply_plain.py 2>/dev/null | grep ct > $OUT/multiply_plain.txt; cat $OUT/multiply_plain.txt
