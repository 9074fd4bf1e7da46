#!/bin/bash
# round 4, GPU session 1: suite + the default bench line (all secondaries) + the mul_mid<14> A/B (LDS parking, slice-major order)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.err
python - <<PY
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('HEAD',d['value'],d['values'],d['spread'],d['kernels_ms_per_step'],d.get('power'))
for k,v in d['secondary'].items(): print(k,v['value'],v['values'],v['spread'],v['kernels_ms_per_step'],(v['cpu_baseline'] or {}).get('value'),v['parity'][:60])
PY
bash tools/ab_libs.sh "prevmid parkonly sliceonly park8" --n 16384 --batch 1024 --steps 5 --warmup 2 --repeats 3 --check-items 8 > $O/ab16384.txt 2>&1
cat $O/ab16384.txt
