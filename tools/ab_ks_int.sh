cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02_d; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
for arm in split; do
  if [ $arm = whole ]; then export HIPBFV_NO_SPLIT_KS_INT=1; else unset HIPBFV_NO_SPLIT_KS_INT; fi
  timeout 300 python bench.py --coeff-bits 54,54,54,56 --steps 5 --warmup 2 --no-cpu 2>/dev/null | tail -1 > $OUT/ab_$arm.json
  python -c "
import json; d=json.load(open('$OUT/ab_$arm.json')); print('$arm', d['value'], d['parity'][:30], d['kernels_ms_per_step'])"
done
