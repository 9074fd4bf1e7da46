#!/bin/bash
# round 4, GPU session 12: grouped loads in ks_mac / ks_moddown (whole-polynomial key switch: n = 32768 and the single-ciphertext path)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
bash tools/ab_libs.sh "prev5" --n 32768 --batch 256 --steps 3 --warmup 1 --repeats 3 --check-items 4 > $O/ab_n32768.txt 2>&1
cat $O/ab_n32768.txt
for lib in new prev5; do
  if [ $lib = new ]; then unset HIPBFV_LIB; else export HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$lib.so; fi
  echo "== $lib"; timeout 300 python tools/latency.py 8192 2>/dev/null | tail -1; timeout 300 python tools/latency.py 16384 2>/dev/null | tail -1
done > $O/latency_ab.txt 2>&1
cat $O/latency_ab.txt
