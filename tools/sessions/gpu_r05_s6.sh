#!/bin/bash
# round 5, session 6: the LDS placement that is conflict free for 16-lane stores as well as 32-lane loads (nttcore.hpp lds_pos), the buffer
# addressing in the integer / mixed arms; variant `nomerge` = the same sources with the SI load/store merger off in kernels_split.hip
# (ds_read2st64_b64 costs 8 LDS cycles where two ds_read_b64 cost 4).  Whole suite, then interleaved A/B against session 4.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_s6; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/ab_libs.sh "nomerge s4" --steps 5 --warmup 2 --repeats 3 > $O/ab_n8192.txt 2>&1; cat $O/ab_n8192.txt
bash tools/ab_libs.sh "nomerge s4" --workload ntt --steps 100 --warmup 10 --repeats 3 > $O/ab_ntt.txt 2>&1; cat $O/ab_ntt.txt
bash tools/ab_libs.sh "nomerge s4" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 > $O/ab_n16384.txt 2>&1; cat $O/ab_n16384.txt
bash tools/ab_libs.sh "nomerge s4" --coeff-bits 54,54,54,56 --steps 5 --warmup 2 --repeats 3 > $O/ab_3x54.txt 2>&1; cat $O/ab_3x54.txt
bash tools/ab_libs.sh "nomerge s4" --workload dot_prod --n 16384 --batch 256 --steps 3 --warmup 1 --repeats 3 > $O/ab_dot.txt 2>&1; cat $O/ab_dot.txt
