#!/bin/bash
# round 4, GPU session 3: the hand-pipelined PIR product kernels (JU = 2 default; variants JU = 1, 4; the r03 kernel) + parity
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_program.py tests/test_gpu_baseline_configs.py tests/test_gpu_properties.py -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/ab_libs.sh "pirold pirju1 pirju4" --workload pir --n 16384 --batch 256 --pir-rows 512 --steps 5 --warmup 2 --repeats 3 > $O/ab_pir16384.txt 2>&1
cat $O/ab_pir16384.txt
bash tools/ab_libs.sh "pirold pirju1 pirju4" --workload pir --batch 256 --steps 10 --warmup 2 --repeats 3 > $O/ab_pir8192.txt 2>&1
cat $O/ab_pir8192.txt
bash tools/ab_libs.sh "pirold pirju1" --workload pir --batch 256 --pir-direct --steps 10 --warmup 2 --repeats 3 > $O/ab_pir8192_direct.txt 2>&1
cat $O/ab_pir8192_direct.txt
