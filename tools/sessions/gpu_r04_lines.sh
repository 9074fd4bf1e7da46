#!/bin/bash
# round 4 evidence, step 3 (GPU, after the PMC merge): the driver's default command, every other bench line, the suite log, smoke
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r04_v1}
O=gpurun_out/$TAG; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
bash tools/gpu_bench_lines.sh $TAG all
if [ -z "$SKIP_PYTEST" ]; then ( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log; fi
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
