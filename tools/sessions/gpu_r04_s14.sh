#!/bin/bash
# round 4, GPU session 14: the derived auxiliary-base bound (context.cpp) -- the tests that exercise it, then interleaved A/B against
# SEAL's sizing (HIPBFV_SEAL_BOUND=1) on the configurations whose base changes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_properties.py tests/test_gpu_program.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/ab_env.sh "HIPBFV_SEAL_BOUND=1 HIPBFV_NO_GRID=1" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_env.sh "HIPBFV_SEAL_BOUND=1" --workload chi_sq --n 16384 --batch 256 --steps 5 --warmup 2 --repeats 3 > $O/ab_chi.txt 2>&1; cat $O/ab_chi.txt
