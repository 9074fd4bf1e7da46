#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh "m16 m16np" --n 16384 --batch 1024 --steps 5 --warmup 2
bash tools/ab_libs.sh "m16" --workload chi_sq --n 16384 --batch 256 --steps 3 --warmup 1
