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


# ---------------------------------------------------------------------------------------------------------------------
# The RCCL code path on the one GPU a box has (VERDICT r03 missing #2): a process group of ONE rank with backend "nccl" is a
# complete communicator -- every collective of sunscreen_amd/dist.py goes through librccl and completes on the device.
# HIPBFV_DIST_FORCE=1 makes the module issue them although world_size is 1.  (No 1 -> 8 scaling curve exists: the driver
# never had an 8-GPU node for this repository; this test executes the transport's code path, not its performance.)
# ---------------------------------------------------------------------------------------------------------------------
_RCCL_SCRIPT = r'''
import os, sys
ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HIPBFV_DIST_FORCE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
from sunscreen_amd import Context, GaloisKeys, KeyGenerator, RelinearizationKeys
from sunscreen_amd import dist as D
from sunscreen_amd.batch import BatchEvaluator
from sunscreen_amd.seal import CoefficientModulus, PlainModulus

rank, local_rank, world = D.init("nccl")
assert (rank, world) == (0, 1) and dist.is_initialized()
assert D.is_nccl() and not D.solo(), "the collectives below must be issued on an RCCL communicator"
dev = "cuda:0"
D.barrier()                      # dist.barrier(device_ids=[...]) under RCCL
D.barrier_sync()

n = 16384
ctx = Context.from_raw(n, [m.value() for m in CoefficientModulus.bfv_default(n)], PlainModulus.batching(n, 17).value())
ev = BatchEvaluator(ctx)
kg = KeyGenerator(ctx, seed=7)
rk = kg.create_relinearization_keys()
gk = kg.create_galois_keys()   # SEAL's default set: 2 log2(n) - 1 = 27 keys, 486 MiB in the wire format
gbytes = gk.as_bytes(compression=0)
assert len(gbytes) > (256 << 20), len(gbytes)   # more than one 256 MiB message: the chunked payload loop runs
# broadcast_bytes: length + payload as uint8 tensors ON THE DEVICE, in 256 MiB messages
for key, cls in ((rk, RelinearizationKeys), (gk, GaloisKeys)):
    blob = key.as_bytes(compression=0)
    got = D.broadcast_bytes(blob, 0, dev)
    assert got == blob, "key bytes changed in the broadcast"
    copy = cls.from_bytes(ctx, got)               # what a non-owner rank does with them
    assert copy.as_bytes(compression=0) == blob
    assert D.replicate_keys(ctx, key, cls, 0, dev) is key
# broadcast_tensor: a device tensor above 1 GiB travels in two messages
big = torch.arange((1 << 27) + 12345, dtype=torch.int64, device=dev)
ref = big.clone()
out = D.broadcast_tensor(big, big.shape, torch.int64, dev, 0)
assert out.is_cuda and torch.equal(out, ref)
del big, ref, out
# gather_results / reduce_ciphertexts on device-resident ciphertext batches
gen = torch.Generator(device=dev); gen.manual_seed(1)
q = [m.value() for m in CoefficientModulus.bfv_default(n)]
ct = torch.stack([torch.stack([torch.randint(0, q[i], (n,), generator=gen, device=dev, dtype=torch.int64) for i in range(ctx.K)]) for _ in range(6)]).reshape(3, 2, ctx.K, n)
full = D.gather_results(ct, 3)
assert full.is_cuda and torch.equal(full, ct)
summed = D.reduce_ciphertexts(ct, ev.add, 0)
assert torch.equal(summed, ct)                    # one rank: the sum of one term
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
el = D.timed_steps(lambda: ev.add(ct, ct), 3, 1, dev)
assert el > 0
torch.cuda.synchronize()
maps = open("/proc/self/maps").read()
assert "librccl" in maps, "librccl is not mapped: the collectives did not go through RCCL"
dist.destroy_process_group()
print("RCCL_PATH_OK", [l.split()[-1] for l in maps.splitlines() if "librccl" in l][0])
'''


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_every_collective_of_the_dist_module_runs_under_rccl_on_a_single_rank_group(tmp_path):
    """init (backend nccl), is_nccl, barrier(device_ids), broadcast_bytes with a 486 MB Galois key set (two 256 MiB messages on the
    device), replicate_keys, broadcast_tensor above 1 GiB, gather_results, reduce_ciphertexts, the MAX all-reduce of the timing:
    every `is_nccl()` branch of sunscreen_amd/dist.py executes, payloads come back bit-equal, librccl is mapped."""
    script = tmp_path / "rccl_path.py"
    script.write_text(_RCCL_SCRIPT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, str(script), ROOT, str(_free_port())], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "RCCL_PATH_OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_bench_single_rank_with_the_rccl_process_group_initialised():
    """bench.py --gpus 1 with HIPBFV_BENCH_FORCE_DIST=1: the process group (backend nccl) exists, the key broadcast, the barriers,
    the MAX-reduce of the elapsed time, the all-rank decrypt gate and the result gather all run on it; the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(HIPBFV_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "4096", "--batch", "64", "--steps", "2", "--warmup", "1", "--repeats", "2",
                        "--check-items", "8", "--gpus", "1", "--no-cpu", "--no-power", "--gather"], env=env, capture_output=True, text=True, timeout=900)
    d = _line(p)
    assert d["n_gpus"] == 1 and d["config"]["collectives"] == "nccl" and d["repeats"] == 2 and len(d["values"]) == 2
    assert "all 64 results decrypt" in d["parity"] and d["result_gather_ms"] > 0
