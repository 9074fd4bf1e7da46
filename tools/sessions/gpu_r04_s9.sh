#!/bin/bash
# round 4, GPU session 9: per-switch full suites on the round's kernels, the threaded drop-in numbers, single-call latencies
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
( time bash tools/gpu_variant_suites.sh ) > $O/variant_suites.txt 2>&1
cat $O/variant_suites.txt
g++ -O2 -std=c++17 -Iinclude tools/mt_dropin.cpp -Lsunscreen_amd/lib -lhipbfv -Wl,-rpath,$GRAFT_REPO_ROOT/sunscreen_amd/lib -lpthread -o /tmp/mt_dropin
( /tmp/mt_dropin 8192 1.5; HIPBFV_NO_COMBINE=1 /tmp/mt_dropin 8192 1.0; /tmp/mt_dropin 16384 1.5; /tmp/mt_dropin 16384 1.5 chi_sq ) > $O/mt_dropin.txt 2>&1; cat $O/mt_dropin.txt
timeout 300 python tools/latency.py > $O/latency_n8192.json 2>$O/latency.err; cat $O/latency_n8192.json
timeout 300 python tools/latency.py --n 16384 > $O/latency_n16384.json 2>>$O/latency.err; cat $O/latency_n16384.json
timeout 300 python tools/program_latency.py > $O/program_latency_n8192.json 2>$O/program_latency.err; cat $O/program_latency_n8192.json
