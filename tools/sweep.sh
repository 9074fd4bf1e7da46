#!/bin/bash
# tools/sweep.sh -- old vs split pipelines at the default chunk (run through gpurun)
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  HIPBFV_NO_SPLIT_MUL=$1 HIPBFV_NO_SPLIT_KS=$2 python bench.py --steps 3 --warmup 1 --no-cpu --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('nosplit_mul=$1 nosplit_ks=$2', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"
done
python bench.py --n 16384 --batch 1024 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('n16384', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['parity'])"
