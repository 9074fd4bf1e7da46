cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r01_v11; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
for round in 1 2; do
  for c in 1024 2048 4096; do
    timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --chunk $c 2>/dev/null | tail -1 > $OUT/chunk_${c}_$round.json
  done
  for arm in grid nogrid; do
    if [ $arm = nogrid ]; then export HIPBFV_NO_GRID=1; else unset HIPBFV_NO_GRID; fi
    timeout 300 python bench.py --n 16384 --batch 1024 --steps 3 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $OUT/ab_${arm}_n16384_$round.json
  done
  unset HIPBFV_NO_GRID
done
python - <<PY | tee $OUT/summary.txt
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], d["value"], d["parity"][:24], d["kernels_ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY
