#!/bin/bash
# round 5, session 7: per-row packing as the default at n = 16384 (pipelined per-row arm of the multiply tail): targeted tests, then
# interleaved A/B default vs HIPBFV_PACK_ROWS=0 on the workloads that multiply at n = 16384, and the headline as a sanity check
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s7; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_baseline_configs.py tests/test_gpu_parity.py tests/test_gpu_integer_model.py tests/test_gpu_program.py -m gpu -q ) > $O/pytest_subset.log 2>&1; grep -E "passed|failed" $O/pytest_subset.log
bash tools/ab_env.sh "HIPBFV_PACK_ROWS=0" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_env.sh "HIPBFV_PACK_ROWS=0" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
bash tools/ab_env.sh "HIPBFV_PACK_ROWS=0" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
bash tools/ab_env.sh "HIPBFV_PACK_ROWS=0" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
