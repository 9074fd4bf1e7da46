#!/bin/bash
# tools/ab_run.sh <tags...> -- on the GPU box: bench every variant library built by tools/build_variant.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for t in "$@"; do
  HIPBFV_LIB=$GRAFT_REPO_ROOT/sunscreen_amd/lib/variants/libhipbfv_$t.so python bench.py --steps 5 --warmup 2 --no-cpu --no-secondary --no-power 2>/dev/null | tail -1 > gpurun_out/ab/$t.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab/$t.json"))
print("$t", d["value"], d["parity"][:20], d["kernels_ms_per_step"])
PY
done
