#!/bin/bash
# round 4, GPU session 5: head / tail kernels with their global loads un-serialised (r04) vs the previous kernels; full suite first
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/ab_libs.sh "prevld" --steps 10 --warmup 3 --repeats 3 > $O/ab_n8192.txt 2>&1
cat $O/ab_n8192.txt
bash tools/ab_libs.sh "prevld" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_n16384.txt 2>&1
cat $O/ab_n16384.txt
bash tools/ab_libs.sh "prevld" --coeff-bits 54,54,54,56 --steps 10 --warmup 3 --repeats 3 --check-items 8 > $O/ab_3x54.txt 2>&1
cat $O/ab_3x54.txt
bash tools/ab_libs.sh "prevld" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1
cat $O/ab_dot.txt
