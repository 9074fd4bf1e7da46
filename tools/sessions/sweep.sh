#!/bin/bash
for v in d3i3 d2i3 d3i4 d4i4 d2i4; do cp build/libhipbfv_$v.so sunscreen_amd/lib/libhipbfv.so
  python bench.py --steps 3 --warmup 1 --no-cpu --no-check 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['kernels_ms_per_step']['mul_mid'])"
done
