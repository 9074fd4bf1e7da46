#!/bin/bash
# round 4, GPU session 6: scalar residue / plan lookups, un-hoisted first NTT load (new) vs the previous commit (prev2); per-row packing again
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
bash tools/ab_libs.sh "prev2" --workload ntt --steps 200 --warmup 5 --repeats 3 > $O/ab_ntt.txt 2>&1
cat $O/ab_ntt.txt
bash tools/ab_libs.sh "prev2" --steps 10 --warmup 3 --repeats 3 > $O/ab_n8192.txt 2>&1
cat $O/ab_n8192.txt
bash tools/ab_libs.sh "prev2" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_n16384.txt 2>&1
cat $O/ab_n16384.txt
bash tools/ab_env.sh "HIPBFV_PACK_ROWS=1" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab_rows.txt 2>&1
cat $O/ab_rows.txt
bash tools/ab_env.sh "HIPBFV_NO_PACK=1" --steps 10 --warmup 3 --repeats 3 > $O/ab_pack8192.txt 2>&1
cat $O/ab_pack8192.txt
