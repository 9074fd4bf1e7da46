#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/ab_libs.sh "prev" --steps 10 --warmup 2
bash tools/ab_libs.sh "prev" --n 16384 --batch 1024 --steps 5 --warmup 2
bash tools/ab_libs.sh "prev" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1
bash tools/ab_libs.sh "prev" --workload pir --batch 256 --steps 5 --warmup 2
