#!/bin/bash
# round 4, GPU session 2: per-row packing of the multiply's intermediates at n = 16384 -- parity, then A/B against 8-byte rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_baseline_configs.py -m gpu -x -q ) > $O/pytest_props.log 2>&1
tail -4 $O/pytest_props.log
bash tools/ab_env.sh "HIPBFV_NO_PACK=part" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_rows.txt 2>&1
cat $O/ab_rows.txt
bash tools/ab_env.sh "HIPBFV_NO_PACK=part" --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_rows_chi.txt 2>&1
cat $O/ab_rows_chi.txt
