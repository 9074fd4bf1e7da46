#!/bin/bash
# round 4, GPU session 8: experiment -- K known at compile time in the all-FP64 head / tail kernels (variant fullk: valid for K == KMAX only)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
bash tools/ab_libs.sh "fullk" --steps 10 --warmup 3 --repeats 3 > $O/ab_n8192.txt 2>&1
cat $O/ab_n8192.txt
bash tools/ab_libs.sh "fullk" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_n16384.txt 2>&1
cat $O/ab_n16384.txt
