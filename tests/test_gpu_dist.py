"""`python bench.py --gpus 2` spawns its own ranks (no torchrun) and reports n_gpus = 2.

Runs on a 1-GPU box through the validation knobs bench.py keeps for that purpose: HIPBFV_BENCH_ONE_DEVICE=1 puts both
ranks on device 0 and HIPBFV_BENCH_BACKEND=gloo replaces RCCL (two ranks cannot share one device under RCCL).  The whole
N>1 path runs: launcher, rendezvous, key broadcast from rank 0 (dist.replicate_keys), per-rank shards, barriers,
MAX-over-ranks timing, the all-rank decrypt gate and the optional result gather."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HIPBFV_BENCH_ONE_DEVICE="1", HIPBFV_BENCH_BACKEND="gloo")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "4096", "--batch", "64", "--steps", "2", "--warmup", "1",
                        "--check-items", "8"] + extra, env=env, capture_output=True, text=True, timeout=600)
    return p


def test_bench_gpus_2_spawns_two_ranks_and_reports_them():
    p = _run(["--gpus", "2", "--no-cpu", "--gather"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "batch-sharded x2"
    assert "all 128 results decrypt" in d["parity"] and "bit-exact vs oracle on 8 items" in d["parity"]
    assert d["result_gather_ms"] > 0
    one = _run(["--gpus", "1", "--no-cpu"])
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1
    # weak scaling: two ranks did twice the work of one (on one shared device the rate does not double; the count does)
    assert abs(d["value"] * d["ms_per_step"] / (d1["value"] * d1["ms_per_step"]) - 2.0) < 0.05  # ms_per_step is rounded to 1 us


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    p = _run(["--gpus", "1", "--no-cpu"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29999"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
