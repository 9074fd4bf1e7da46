#!/bin/bash
# tools/gpu_quick_variant.sh <tag> -- one bench run of a variant library and one of the default (headline configuration, parity gate included), then the
# multiply + relinearize integer-model test under the variant: the shortest session that says "same bits, this fast"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$1.so
for arm in $V "" $V ""; do
  HIPBFV_LIB=$arm timeout 40 python bench.py --steps 5 --warmup 2 --repeats 3 --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${arm:+variant}', d['value'], d.get('spread'), d['parity'][:40], d['kernels_ms_per_step'])"
done
HIPBFV_LIB=$V timeout 60 python -m pytest tests/test_gpu_integer_model.py -m gpu -q -x -k "relin" 2>&1 | tail -1
