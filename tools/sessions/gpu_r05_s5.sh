#!/bin/bash
# round 5, session 5: counters on the state of commit 543b9b8 (headline + n = 16384): the three traffic / instruction passes and the two
# stall passes; program-workload tests first (the double-buffered table arena and the retired switches have not met a GPU yet)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s5; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_program.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py tests/test_gpu_properties.py -m gpu -q ) > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
bash tools/gpu_pmc_report.sh r05_s5 mulrelin_n8192 --steps 5 --warmup 2 > $O/pmc_n8192.log 2>&1
bash tools/gpu_pmc_stall.sh r05_s5 mulrelin_n8192 > $O/stall_n8192.log 2>&1
bash tools/gpu_pmc_stall.sh r05_s5 mulrelin_n16384 --n 16384 --batch 1024 > $O/stall_n16384.log 2>&1
cat $O/mulrelin_n8192_pmc_stall.txt $O/mulrelin_n8192_pmc_lds.txt | head -120
