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


def _run(extra, env_extra=None, base=("--n", "4096", "--batch", "64", "--steps", "2", "--warmup", "1", "--check-items", "8")):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(HIPBFV_BENCH_ONE_DEVICE="1", HIPBFV_BENCH_BACKEND="gloo")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(base) + extra, env=env, capture_output=True, text=True, timeout=600)
    return p


def _line(p):
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 only
    return json.loads(lines[0])


def test_bench_gpus_2_spawns_two_ranks_and_reports_them():
    d = _line(_run(["--gpus", "2", "--no-cpu", "--gather"]))
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


def test_pir_database_sharded_by_row_with_the_cross_gpu_sum():
    """SURVEY 8d config 5a / 8e "Exception": examples/pir with the database rows sharded over the ranks; every rank reduces its
    rows to one ciphertext, `dist.reduce_ciphertexts` gathers one ciphertext per rank on rank 0 and adds.  bench.py's gate
    decrypts the answer (database[sel_r][sel_c]), compares the decryption with the oracle's, and -- the database being small
    enough for one device -- compares the sharded answer with the single-GPU lookup over the whole database BIT FOR BIT.
    3 ranks over 10 rows: uneven shards (4, 3, 3)."""
    d = _line(_run(["--gpus", "3", "--workload", "pir", "--pir-rows", "10", "--no-cpu"], base=("--n", "4096", "--batch", "12", "--steps", "2", "--warmup", "1")))
    assert d["n_gpus"] == 3 and d["scaling"] == "strong"
    assert "rows sharded x3" in d["config"]["parallelism"] and "gather(dst=0)" in d["config"]["exchange"]
    assert "equals the single-GPU lookup over the whole database bit for bit" in d["parity"]
    assert "bit-exact vs the oracle" in d["parity"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 10 * 12) < 0.5  # value counts the whole database once per step


def test_total_batch_is_sharded_over_the_ranks():
    """BASELINE configs[3]: "1024-input batch sharded across 8" -- the strong-scaling form: --total-batch T items in total."""
    d = _line(_run(["--gpus", "2", "--total-batch", "65", "--no-cpu", "--no-secondary"]))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_batch"] == 65
    assert d["config"]["batch_per_gpu"] == 33  # rank 0's shard of 65
    assert "all 65 results decrypt" in d["parity"]
