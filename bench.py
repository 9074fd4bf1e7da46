#!/usr/bin/env python3
"""bench.py -- BFV ct x ct multiply + relinearize throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: `--batch` (default 4096) independent
ciphertext pairs, n=8192, SEAL default 128-bit parameters (K=4 data primes + 1 special prime,
t = batching(8192,17) = 114689) -- BASELINE.json configs[2], the configuration the metric is quoted
on.  Inputs are resident in HBM before the timed region.

BASELINE.json's metric names three quantities ("mul+relin ops/sec at n=8192/16384; NTTs/sec") and its `configs` list the
reference's example programs.  The default run times all of them in one process: the headline line is n=8192 and its
`secondary` object carries the n=16384 mul+relin (batch/4 pairs, SEAL default K=8+1), the configs[1] transform workload
(forward+inverse NTT, n=8192, 3 primes), the north star's literal prime set (n=8192, 3 x 54-bit + special prime), and -- on one
GPU -- examples/chi_sq at n=16384 (the 1024-input batch of configs[3] and one GPU's 128-set share of it), examples/dot_prod
(256 sets) and examples/pir over 2^17 entries (128 rows x 1024 columns: one GPU's rows of the 1024 x 1024 database, 128 GiB in transform form), each with its own value, ms_per_step,
repeats, roofline, cpu_baseline and parity gate (64 items / 8 input sets against the oracle), measured with the same --steps /
--warmup / --repeats (`--no-secondary` drops them; the NTT workload times at least 100 of its 0.7 ms steps per region; the whole
default run takes 2-3 minutes).  `--workload ntt|chi_sq|dot_prod|pir|e2e` run one workload alone (same JSON contract).

The CPU oracle appears here in three roles only: client (it generates the keys and the few genuine encryptions the
parity gate needs), checker (parity gate, after the timed region) and `cpu_baseline` (host cores, OpenMP over the
batch).  Nothing in the timed region touches it.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank processes its own `--batch` items (weak
scaling, no data-path collective) or, with `--total-batch T`, its shard of T items (strong scaling: BASELINE configs[3]
"1024-input batch sharded across 8"); time = max over ranks.  Under torchrun the ranks come from the environment
(WORLD_SIZE must equal --gpus); a bare `python bench.py --gpus N` spawns the N ranks itself.  Rank 0 owns the keys and
broadcasts them once (sunscreen_amd.dist.replicate_keys); `--gather` additionally times the optional gather of the
results to rank 0 (reported apart from `value`).  `--workload pir` is the one workload with a data-path exchange
(SURVEY 8e "Exception"): the database is sharded by ROW across the ranks, every rank reduces its rows to one ciphertext
and `dist.reduce_ciphertexts` sums one ciphertext per GPU on rank 0 inside the timed step.
`--gpus N --dry-run` validates the launch environment and prints the per-rank shapes without touching a GPU.

Timing: W warmup steps, then untimed steps for about `--settle-ms` (150 ms: the shader clock settles under the package power
cap these kernels run at; the count is agreed between the ranks and reported as `settle_steps`), then `--repeats` R (5) timed
regions of EXACTLY K steps each, every region between two barriers (+ torch.cuda.synchronize) and MAX-reduced over the ranks:
`value` / `ms_per_step` are the MEDIAN region, `values` lists all R and `spread` = (slowest - fastest) / median.  At N = 1 an untimed leg after the measurement samples `rocm-smi` while the steps keep running (`power`:
package watts, shader clock, cap; `--no-power` skips it).

Prints ONE JSON line on rank 0.
"""
import argparse
import copy
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
HBM_BYTES = 288 * 10**9


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="ciphertext pairs (or polynomials for --workload ntt) per GPU per step; "
                    "pir: database columns (and rows unless --pir-rows)")
    ap.add_argument("--total-batch", type=int, default=0, help="strong-scaling form: this many items in total, sharded over the ranks "
                    "(sunscreen_amd.dist.shard_range) instead of --batch per GPU")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--workload", choices=["mulrelin", "ntt", "chi_sq", "dot_prod", "e2e", "pir"], default="mulrelin",
                    help="mulrelin = the headline; ntt = batched transforms; chi_sq / dot_prod = whole program graphs "
                         "(examples/chi_sq, examples/dot_prod) through the batch graph executor (SURVEY 8d configs 4 / 5b); "
                         "e2e = encode + encrypt both operands, multiply + relinearize, decrypt + decode, all on the device; "
                         "pir = examples/pir lookup over a (--pir-rows x --batch) plaintext database held in transform form, rows "
                         "sharded over the ranks, one cross-GPU sum (SURVEY 8d config 5a)")
    ap.add_argument("--pir-rows", type=int, default=0, help="pir: database rows in total (default = --batch: a square database)")
    ap.add_argument("--pir-direct", action="store_true", help="pir: the hand-written batch primitives (workloads.pir_lookup) instead of the "
                    "compiled `lookup` graph through hipbfv_Program_Run (the A/B arm; same bits)")
    ap.add_argument("--coeff-bits", default="", help="comma-separated prime sizes (CoeffModulus::create, last = special prime) instead of the "
                    "SEAL default set for --n, e.g. 54,54,54,56 for the 3 x 54-bit n=8192 variant BASELINE.json mentions")
    ap.add_argument("--keys", type=int, default=1, help="mulrelin: this many CLIENTS in the batch, each with its own secret, public and relinearisation "
                    "key (item i belongs to client i %% keys -- interleaved, the order a batching server sees): the per-key entry point "
                    "hipbfv_batch_multiply_relin_keys; --keys = --batch is SURVEY 8d config 3's per-ciphertext-key worst case, where the key "
                    "(16 K (K+1) N bytes) is part of every item's compulsory traffic")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of exactly --steps steps each (every one between two barriers); value = the "
                    "MEDIAN region, all of them are printed (`values`, `spread`): one box differs from the next by 2-3 %, a claim smaller than "
                    "the spread of its own repeats is noise")
    ap.add_argument("--chunk", type=int, default=0, help="override the executor's chunk size (ops per launch group)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="ops in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--settle-ms", type=float, default=150.0, help="keep running untimed steps after the W warmup steps until this much wall time "
                    "has passed (device clocks under the power cap settle in tens of ms; 0 = exactly W warmup steps)")
    ap.add_argument("--no-kernel-events", action="store_true", help="time the regions WITHOUT the HIP events the library records around every kernel launch "
                    "(hipbfv_profile_enable): the line then has no roofline (no per-kernel times) -- the one-off measurement of what the events cost")
    ap.add_argument("--no-power", dest="power", action="store_false", help="skip the `power` leg (N=1: after the measurement, rocm-smi is sampled for package "
                    "power and shader clock while the steps keep running, about 3 s)")
    ap.add_argument("--power", dest="power", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(power=True)
    ap.add_argument("--no-secondary", action="store_true", help="headline run only: skip the n=16384 mul+relin and the NTT workload "
                    "that the default run reports under `secondary`")
    ap.add_argument("--full-line", action="store_true", help="print every field of every secondary workload (the default line keeps the secondary records "
                    "compact and ends in `summary`, so that the tail a driver stores carries every quantity of the metric)")
    ap.add_argument("--check-items", type=int, default=64, help="mulrelin: items compared bit for bit with the oracle (BASELINE.md section 3: >= 64)")
    ap.add_argument("--check-sets", type=int, default=8, help="chi_sq / dot_prod: input sets compared bit for bit with the oracle's graph interpreter")
    ap.add_argument("--gather", action="store_true", help="N>1: also time one gather of the result batch to rank 0 (reported as result_gather_ms, never part of value)")
    ap.add_argument("--dry-run", action="store_true", help="validate the launch environment, per-rank shapes and HBM footprint for --gpus N and exit (no GPU needed)")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without torchrun: start one process per GPU (this file again) with the torchrun
    environment, wait for all of them; rank 0 prints the JSON line.  The parent never touches a GPU."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(args.gpus)]
    # watch ALL ranks: the first one that fails takes the others down at once (they would otherwise sit in the rendezvous or in a
    # collective until its timeout -- ten minutes for a rank that died before joining)
    rc = 0
    while True:
        states = [p.poll() for p in procs]
        failed = [c for c in states if c not in (None, 0)]
        if failed:
            rc = failed[0]
            for q in procs:
                if q.poll() is None:
                    q.kill()
            for q in procs:
                q.wait()
            break
        if all(c == 0 for c in states):
            break
        time.sleep(0.2)
    return rc


# ---------------------------------------------------------------------------------------------------------------------
# shapes (shared by the run and by --dry-run)
# ---------------------------------------------------------------------------------------------------------------------
def power_leg(step, sync, seconds=2.5):
    """Package power and shader clock while the workload runs (untimed, after the measurement): rocm-smi is sampled from a
    thread while this thread keeps launching steps.  Every workload of this repository runs at the package power cap, which
    is what bounds the VALU-issue and HBM fractions the line reports; None when rocm-smi is missing or prints nothing usable."""
    import re
    import shutil
    import subprocess
    import threading

    smi = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(smi):
        return None
    dev = os.environ.get("LOCAL_RANK", "0")
    samples, done = [], threading.Event()

    def sampler():
        time.sleep(0.6)  # past the ramp
        for _ in range(3):
            try:
                out = subprocess.run([smi, "-d", dev, "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                break
            w = re.search(r"(?:Current Socket|Average) Graphics Package Power \(W\): ([0-9.]+)", out)
            c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            cap = re.search(r"Max Graphics Package Power \(W\): ([0-9.]+)", out)
            if w and c:
                samples.append((float(w.group(1)), int(c.group(1)), float(cap.group(1)) if cap else None))
        done.set()

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    while not done.is_set() and time.perf_counter() - t0 < 30.0:
        step()
        sync()
    th.join(timeout=5)
    if not samples:
        return None
    return {"package_w": round(sum(x[0] for x in samples) / len(samples), 1), "sclk_mhz": int(sum(x[1] for x in samples) / len(samples)),
            "cap_w": samples[0][2], "samples": len(samples), "source": "rocm-smi while the workload's steps run (untimed leg after the measurement)"}


def default_K(n):
    # data primes of SEAL's 128-bit default set (CoeffModulus::BFVDefault): key level = K + 1
    return {1024: 1, 2048: 1, 4096: 2, 8192: 4, 16384: 8, 32768: 15}[n]


def rank_items(args, rank, world):
    """Items of this rank's shard: --batch each (weak) or shard_range(--total-batch) (strong)."""
    from sunscreen_amd.dist import shard_range

    if args.total_batch:
        lo, hi = shard_range(args.total_batch, rank, world)
        return hi - lo
    return args.batch


def plan_bytes(args, rank, world, K):
    """Resident HBM bytes of one rank's inputs + outputs (the pipeline scratch of one 1024-op chunk comes on top: ~4 GB)."""
    n = args.n
    ct = 2 * K * n * 8
    if args.workload == "pir":
        from sunscreen_amd.dist import shard_range

        rows = args.pir_rows or args.batch
        lo, hi = shard_range(rows, rank, world)
        return {"database_rows": [lo, hi], "database_bytes": (hi - lo) * args.batch * K * n * 8, "query_bytes": (args.batch + rows) * ct,
                "partial_sum_bytes_to_root": ct}
    B = rank_items(args, rank, world)
    if args.workload == "ntt":
        return {"items": B, "resident_bytes": 2 * B * 3 * n * 8}
    nin, nout = {"mulrelin": (2, 1), "e2e": (2, 1), "chi_sq": (3, 4), "dot_prod": (2, 1)}[args.workload]
    plan = {"items": B, "resident_bytes": B * ct * (nin + nout)}
    if args.workload == "mulrelin" and getattr(args, "keys", 1) > 1:  # one relinearisation key (+ public key) per client of this rank
        nk = min(args.keys, B)
        plan["key_sets"] = nk
        plan["key_bytes"] = nk * (16 * K * (K + 1) * n + 2 * (K + 1) * n * 8)
        plan["resident_bytes"] += plan["key_bytes"]
    return plan


def dry_run(args) -> int:
    """`--gpus N --dry-run`: everything that can be checked without a GPU -- the launch environment torchrun / the self-spawn
    provides, the per-rank shards, the HBM footprint against 288 GB, the one-time key broadcast and the collectives the run
    will issue.  Exit code 0 = the launch is consistent."""
    from sunscreen_amd.dist import shard_range

    world = args.gpus
    problems = []
    env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY",
                                          "HIPBFV_BENCH_BACKEND", "HIPBFV_DIST_TIMEOUT_S", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES")}
    if env["WORLD_SIZE"] is not None and int(env["WORLD_SIZE"]) != world:
        problems.append(f"WORLD_SIZE={env['WORLD_SIZE']} contradicts --gpus {world}")
    if env["WORLD_SIZE"] is not None and env["MASTER_ADDR"] not in (None, "127.0.0.1", "localhost"):
        problems.append(f"MASTER_ADDR={env['MASTER_ADDR']}: one node only -- use 127.0.0.1 (the container hostname may not resolve)")
    if world > 1 and (env["HSA_ENABLE_IPC_MODE_LEGACY"] or "0") != "0":
        problems.append("HSA_ENABLE_IPC_MODE_LEGACY must be 0 (the host driver only supports dmabuf IPC; RCCL fails with hipIpcGetMemHandle otherwise)")
    K = len(args.coeff_bits.split(",")) - 1 if args.coeff_bits else default_K(args.n)
    KK = K + 1
    key_bytes = 16 * K * KK * args.n
    ranks = []
    for r in range(world):
        p = plan_bytes(args, r, world, K)
        total = p.get("resident_bytes", 0) + p.get("database_bytes", 0) + p.get("query_bytes", 0)
        if total + (8 << 30) > HBM_BYTES:
            problems.append(f"rank {r}: {total / 2**30:.1f} GiB resident + scratch exceeds 288 GB of HBM")
        if args.workload != "pir" and p["items"] == 0:
            problems.append(f"rank {r} has no items (--total-batch {args.total_batch} < --gpus {world})")
        ranks.append(dict(rank=r, local_rank=r, **p))
    if args.workload == "pir" and (args.pir_rows or args.batch) < world:
        problems.append("fewer database rows than ranks")
    ngal = {"dot_prod": args.n.bit_length() - 1}.get(args.workload, 0)
    collectives = [
        "init_process_group(nccl, timeout=HIPBFV_DIST_TIMEOUT_S or 600 s)",
        f"broadcast x2 per key object, <= 256 MiB per message: relin keys {key_bytes} B, public key {2 * KK * args.n * 8} B, secret key {KK * args.n * 8} B"
        + (f", Galois keys {ngal} x {key_bytes} B = {ngal * key_bytes / 2**20:.0f} MiB" if ngal else ""),
        "barrier(device_ids=[local_rank]) before and after the timed region",
        "all_reduce(MAX) of the elapsed time (1 double); all_reduce(MIN) of the decrypt gate (1 int64)",
    ]
    if args.workload == "pir":
        collectives.append(f"per step: gather(dst=0) of one ciphertext per rank ({2 * K * args.n * 8} B each) + {world - 1} additions on rank 0")
        collectives.append("once: broadcast of the query ciphertexts from rank 0 (<= 1 GiB per message)")
    if args.gather:
        collectives.append("after the timed region: gather(dst=0) of every rank's result block (padded to the longest shard)")
    doc = {"dry_run": True, "n_gpus": world, "workload": args.workload, "scaling": scaling_of(args), "poly_modulus_degree": args.n,
           "coeff_modulus_primes": KK, "environment": env, "ranks": ranks, "collectives": collectives,
           "launch": (f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {world} --master-addr 127.0.0.1 --master-port P bench.py --gpus {world} ..."
                      if world > 1 else "python bench.py"),
           "problems": problems, "ok": not problems}
    try:
        import torch

        doc["torch"] = torch.__version__
        doc["nccl_available"] = bool(torch.distributed.is_nccl_available())
        doc["visible_devices"] = torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception as e:  # torch is plumbing; the shape checks above do not need it
        doc["torch"] = f"import failed: {e}"
    from sunscreen_amd import _lib

    try:
        _lib.load()
        doc["libhipbfv"] = "loaded, every symbol of include/hipbfv.h exported"
    except Exception as e:
        problems.append(f"libhipbfv.so: {e}")
        doc["ok"] = False
    print(json.dumps(doc))
    return 0 if doc["ok"] else 1


def scaling_of(args):
    # pir shards one fixed database by row; --total-batch shards one fixed batch: total work fixed = strong
    return "strong" if (args.total_batch or args.workload == "pir") else "weak"


class Env:
    """What main() sets up once and every measurement shares."""

    def __init__(self, rank, local_rank, world, dev):
        self.rank, self.local_rank, self.world, self.dev = rank, local_rank, world, dev


def setup(args) -> Env:
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torchrun --nproc-per-node {args.gpus}) or drop the torchrun environment"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # validation knobs for boxes with fewer GPUs than ranks (not used by the driver): all ranks on device 0, gloo
    if os.environ.get("HIPBFV_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py rank {rank}: no GPU for LOCAL_RANK={local_rank} ({torch.cuda.device_count() if torch.cuda.is_available() else 0} visible); "
                 f"--gpus {args.gpus} needs that many devices on this node")
    torch.cuda.set_device(local_rank)
    # HIPBFV_BENCH_FORCE_DIST=1 (validation on a 1-GPU box, tests/test_gpu_dist.py): a process group of ONE rank is created and every
    # collective of the N>1 path is issued on it -- the RCCL code path executes although no second GPU exists
    force = world == 1 and os.environ.get("HIPBFV_BENCH_FORCE_DIST") == "1"
    if force:
        os.environ["HIPBFV_DIST_FORCE"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force:
        from sunscreen_amd.dist import init_timeout

        dist.init_process_group(os.environ.get("HIPBFV_BENCH_BACKEND", "nccl"), rank=rank, world_size=world, timeout=init_timeout())
    from sunscreen_amd import _lib

    _lib.load().hipbfv_set_device(local_rank)
    return Env(rank, local_rank, world, f"cuda:{local_rank}")


def kernel_source_hash() -> str:
    """sha256 over the device code and the host code that picks launch sequences and reduce masks: the PMC-derived fields of
    the bench line are only valid for the kernels they were measured on (profiles/pmc_traffic.json records this hash)."""
    d = os.path.join(ROOT, "sunscreen_amd", "csrc")
    device_headers = ("devctx.hpp", "devarith.hpp", "griddot.hpp", "moddown_d.hpp", "nttshape.hpp", "nttcore.hpp", "behzcore.hpp", "kernels.hpp", "rng.hpp")
    names = sorted(f for f in os.listdir(d) if f.endswith(".hip") or f in device_headers or f in ("context.cpp", "evaluator.cpp", "Makefile"))
    h = hashlib.sha256()
    for f in names:
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read() + b"\0")
    # ... and the flag string the LOADED library was compiled with (hipbfv_build_flags: the Makefile's CXXFLAGS, or the -D set of a
    # tools/build_variant.sh library selected through HIPBFV_LIB): a -DMID_EPT_14=8 build has the same sources and other kernels
    from sunscreen_amd import _lib

    h.update(b"flags\0" + _lib.build_flags().encode() + b"\0")
    return h.hexdigest()[:16]


def measure(args, env: Env, secondary: bool = False):
    """Run ONE workload: W warm-up steps, K timed steps between barriers, parity gate, roofline, CPU baseline.
    Returns the JSON object on rank 0, None on the other ranks."""
    import torch
    import torch.distributed as dist

    from sunscreen_amd import Context, RelinearizationKeys
    from sunscreen_amd import dist as D
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    # the oracle is the checker and the CPU baseline only
    from oracle import bfv_oracle as O

    rank, world, dev = env.rank, env.world, env.dev
    collective = not D.solo()  # world > 1, or a forced single-rank group (HIPBFV_BENCH_FORCE_DIST)
    n = args.n
    primes = O.coeff_modulus_create(n, [int(b) for b in args.coeff_bits.split(",")]) if args.coeff_bits else O.bfv_default(n)
    pset = ("CoeffModulus::create(" + args.coeff_bits + ")") if args.coeff_bits else "SEAL default 128-bit"
    t = O.plain_batching(n, 17)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    if args.chunk:
        ev.set_chunk_ops(args.chunk)
    K, KK = ctx.K, ctx.KK
    B = rank_items(args, rank, world)
    total_items = args.total_batch if args.total_batch else B * world
    share = f"{args.total_batch} in total, sharded" if args.total_batch else f"{B}/GPU"
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EA10001 + rank)
    cdev = dev if D.is_nccl() else "cpu"  # where small collective payloads live

    def uniform_residues(shape_prefix, nres, primes_):
        # uniform canonical residues per prime: valid ciphertext / polynomial bit patterns
        out = torch.empty(shape_prefix + (nres, n), dtype=torch.int64, device=dev)
        for i in range(nres):
            out[..., i, :] = torch.randint(0, primes_[i % len(primes_)], shape_prefix + (n,), generator=gen, device=dev, dtype=torch.int64)
        return out

    def owner_keys(seed, galois_elts=None):
        """Rank 0 is the key owner (the oracle stands in for the client's key generator); the other ranks receive the
        SEAL-format bytes once and build their device copies from them -- the deployment form, SURVEY 8(e)."""
        from sunscreen_amd import GaloisKeys, PublicKey, SecretKey

        sk = pk = rk = gk = None
        rkd = skd = pkd = gkd = None
        if rank == 0:
            O.seed(seed)
            sk, pk, rk, gk = o.keygen(galois_elts=galois_elts)
            rkd, skd, pkd = RelinearizationKeys.from_array(ctx, rk), SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)
            gkd = GaloisKeys.from_arrays(ctx, gk) if gk else None
        rkd = D.replicate_keys(ctx, rkd, RelinearizationKeys, 0, cdev)
        skd = D.replicate_keys(ctx, skd, SecretKey, 0, cdev)
        pkd = D.replicate_keys(ctx, pkd, PublicKey, 0, cdev)
        if galois_elts:
            gkd = D.replicate_keys(ctx, gkd, GaloisKeys, 0, cdev)
        return sk, pk, rk, gk, rkd, skd, pkd, gkd

    exchange = None
    nkeys = 1
    if args.workload == "mulrelin" and args.keys > 1:
        # ---- per-key batch: `keys` clients, item i belongs to client i % keys (hipbfv_batch_multiply_relin_keys) ----
        from sunscreen_amd import KeyGenerator

        o = O.Oracle(n, primes, t)
        nkeys = min(args.keys, B)
        key_index = (np.arange(B) % nkeys).astype(np.uint32)
        # every client generates its own secret, public and relinearisation key (the library's device key generator: one SEAL
        # KeyGenerator per client) and encrypts its own items; a rank's clients are its own (no key crosses ranks)
        clients = []
        va = torch.randint(-128, 129, (B, n), generator=gen, device=dev, dtype=torch.int64)
        vb = torch.randint(-128, 129, (B, n), generator=gen, device=dev, dtype=torch.int64)
        a = torch.empty((B, 2, K, n), dtype=torch.int64, device=dev)
        b = torch.empty((B, 2, K, n), dtype=torch.int64, device=dev)
        pa, pb = ev.encode(va, signed=True), ev.encode(vb, signed=True)
        for c in range(nkeys):
            kg = KeyGenerator(ctx, seed=0x6E75 + 7919 * rank + c)
            cl = {"sk": kg.secret_key(), "pk": kg.create_public_key(), "rk": kg.create_relinearization_keys()}
            clients.append(cl)
            mine = torch.arange(c, B, nkeys, device=dev)
            a[mine] = ev.encrypt(pa[mine], cl["pk"], seed=0xA000 + c)
            b[mine] = ev.encrypt(pb[mine], cl["pk"], seed=0xB000 + c)
        del pa, pb
        rk_sets = [cl["rk"] for cl in clients]
        ncheck = 0 if args.no_check else min(args.check_items, B)
        out = torch.empty((B, 2, K, n), dtype=torch.int64, device=dev)

        def step():
            ev.multiply_relin_keys(a, b, rk_sets, key_index, out=out)

        key_bytes = 16 * K * KK * n
        # SURVEY 8(d) + the keys that are now compulsory: read 2 ciphertexts, write 1, and every DISTINCT key of the step once
        unit_bytes = 48 * K * n + key_bytes * nkeys // B
        units_per_step = B
        metric, unit = "bfv_mul_relin_ops_per_sec", "ops/s"
        workload = (f"BFV ct*ct multiply+relinearize, n={n}, K={K}+1 {pset} primes, t={t}, batch={share} pairs of {nkeys} clients "
                    f"(item i -> client i mod {nkeys}; {nkeys * key_bytes / 2**30:.2f} GiB of relinearisation keys resident)")
    elif args.workload == "mulrelin":
        o = O.Oracle(n, primes, t)
        sk, pk, rk, _, rkd, skd, pkd, _ = owner_keys(0xBF5 + 17)
        # the whole batch is genuine: slot vectors in [-128, 128] (products stay below t/2),
        # batch-encoded and encrypted under the public key by the library's own encryptor, resident in HBM
        va = torch.randint(-128, 129, (B, n), generator=gen, device=dev, dtype=torch.int64)
        vb = torch.randint(-128, 129, (B, n), generator=gen, device=dev, dtype=torch.int64)
        a = ev.encrypt(ev.encode(va, signed=True), pkd, seed=0xA0 + 2 * rank)
        b = ev.encrypt(ev.encode(vb, signed=True), pkd, seed=0xA1 + 2 * rank)
        ncheck = 0 if args.no_check else min(args.check_items, B)
        out = torch.empty((B, 2, K, n), dtype=torch.int64, device=dev)

        def step():
            ev.multiply_relin(a, b, rkd, out=out)

        unit_bytes = 48 * K * n  # read 2 ciphertexts, write 1 (compulsory HBM traffic per op)
        units_per_step = B
        metric, unit = "bfv_mul_relin_ops_per_sec", "ops/s"
        workload = f"BFV ct*ct multiply+relinearize, n={n}, K={K}+1 {pset} primes, t={t}, batch={share} pairs"
    elif args.workload == "e2e":
        o = O.Oracle(n, primes, t)
        sk, pk, rk, _, rkd, skd, pkd, _ = owner_keys(0xE2E + 17)
        va = torch.randint(0, 257, (B, n), generator=gen, device=dev, dtype=torch.int64)
        vb = torch.randint(0, 257, (B, n), generator=gen, device=dev, dtype=torch.int64)
        holder = {}

        def step():
            ca = ev.encrypt(ev.encode(va), pkd, seed=0xA + rank, first_op=0)
            cb = ev.encrypt(ev.encode(vb), pkd, seed=0xB + rank, first_op=0)
            prod = ev.multiply_relin(ca, cb, rkd)
            holder["ct"] = prod
            holder["out"] = ev.decode(ev.decrypt(prod, skd))

        unit_bytes = 32 * n + 8 * n  # compulsory: read two slot vectors, write one
        units_per_step = B
        metric, unit = "bfv_encrypt_mulrelin_decrypt_per_sec", "ops/s"
        workload = (f"encode+encrypt x2 -> multiply+relinearize -> decrypt+decode on the device, n={n}, K={K}+1 SEAL default primes, "
                    f"t={t}, batch={share} slot-vector pairs")
    elif args.workload == "pir":
        from sunscreen_amd.program import FheProgram, TransformedPlaintext
        from sunscreen_amd.workloads import pir_lookup, pir_lookup_graph

        cols = args.batch
        rows = args.pir_rows or args.batch
        lo, hi = D.shard_range(rows, rank, world)
        o = O.Oracle(n, primes, t)
        sk, pk, rk, _, rkd, skd, pkd, _ = owner_keys(0x914 + 17)
        # the database: scalar entries (value in coefficient 0) known to every rank by (row, column); rank r holds rows
        # [lo, hi) in transform form -- the server's static state, SURVEY 8(d) config 5a "pre-NTT'd and sharded by row"
        cgen = torch.Generator(device="cpu")
        cgen.manual_seed(0x914DB)
        vals = torch.randint(1, 1000, (rows, cols), generator=cgen, dtype=torch.int64)
        db_ntt = torch.empty((hi - lo, cols, K, n), dtype=torch.int64, device=dev)
        row = torch.zeros((cols, n), dtype=torch.int64, device=dev)
        for i in range(lo, hi):
            row[:, 0] = vals[i].to(dev)
            db_ntt[i - lo] = ev.plain_to_ntt(row)
        del row
        sel_r, sel_c = 7 % rows, (cols // 3) % cols
        # the client's query (rank 0 stands in for it) reaches every shard once, before the timed region
        cq = rq = None
        if rank == 0:
            onehot_c = torch.zeros((cols, n), dtype=torch.int64, device=dev)
            onehot_c[sel_c, 0] = 1
            onehot_r = torch.zeros((rows, n), dtype=torch.int64, device=dev)
            onehot_r[sel_r, 0] = 1
            cq = ev.encrypt(onehot_c, pkd, seed=0xC0)
            rq = ev.encrypt(onehot_r, pkd, seed=0xD0)
        cq = D.broadcast_tensor(cq, (cols, 2, K, n), torch.int64, dev, 0)
        rq = D.broadcast_tensor(rq, (rows, 2, K, n), torch.int64, dev, 0)
        rq_local = rq[lo:hi].contiguous()
        holder = {}
        if args.pir_direct:
            run_shard = lambda: pir_lookup(ev, cq, rq_local, db_ntt, rkd)  # noqa: E731
            via = "hand-written batch primitives (workloads.pir_lookup)"
        else:
            # the reference's `lookup` fhe_program (examples/pir/src/main.rs:16-45) for this shard's rows, in the serde JSON form
            # the compiler emits, run unchanged by the graph executor; one input set (the reference's call shape); the database
            # entries are plaintext ARGUMENTS, handed over already transformed (the server's static state)
            prog = FheProgram.from_json(pir_lookup_graph(hi - lo, cols).to_json())
            pargs = ([cq[j : j + 1] for j in range(cols)] + [rq_local[i : i + 1] for i in range(hi - lo)]
                     + [TransformedPlaintext(db_ntt[i, j]) for i in range(hi - lo) for j in range(cols)])
            bound = prog.prepare(ev, pargs, rkd)
            run_shard = lambda: bound()[0]  # noqa: E731
            via = f"compiled `lookup` graph ({len(prog.nodes)} nodes) through hipbfv_Program_Run; schedule: " + "; ".join(prog.describe()[:3])

        def step():
            # rows of this shard -> one partial ciphertext; one ciphertext per GPU summed on rank 0 (SURVEY 8e "Exception")
            holder["out"] = D.reduce_ciphertexts(run_shard(), ev.add, 0)

        exchange = f"gather(dst=0) of one ciphertext per rank ({2 * K * n * 8} B) + {world - 1} additions on rank 0, inside every timed step" if world > 1 else None
        unit_bytes = 8 * K * n  # compulsory traffic per database entry: its transform-domain residues, read once
        units_per_step = (hi - lo) * cols
        total_items = rows * cols
        metric, unit = "pir_db_entries_per_sec", "entries/s"
        workload = (f"examples/pir lookup: {rows}x{cols} plaintext database in transform form ({rows * cols * K * n * 8 / 2**30:.1f} GiB"
                    + (f", rows sharded over {world} GPUs" if world > 1 else "") + f"), one encrypted query per step, n={n}, K={K}+1 SEAL default primes, t={t}; "
                    + via)
    elif args.workload in ("chi_sq", "dot_prod"):
        from oracle.program_interp import run_program
        from sunscreen_amd.workloads import chi_sq_optimized, dot_product

        o = O.Oracle(n, primes, t)
        if args.workload == "chi_sq":
            prog, nin, elts = chi_sq_optimized(), 3, None
            lanes = 0
        else:
            lanes = n // 2
            prog, nin = dot_product(lanes), 2
            elts = sorted({o.galois_elt_from_step(1 << i) for i in range(lanes.bit_length() - 1)} | {2 * n - 1})
        sk, pk, rk, gk, rkd, skd, pkd, gkd = owner_keys(0xC415 + 17, elts)
        ins = [uniform_residues((B, 2), K, primes) for _ in range(nin)]
        ncheck = 0 if args.no_check else min(args.check_sets, B)
        # the first `ncheck` input sets are genuine encryptions (the library's encryptor, under the owner's public key) of small
        # slot vectors: they decrypt to the program's value AND are compared bit for bit with the oracle's graph interpreter
        vals = torch.randint(0, 7, (nin, max(ncheck, 1), n), generator=gen, device=dev, dtype=torch.int64)
        for k in range(nin):
            if ncheck:
                ins[k][:ncheck] = ev.encrypt(ev.encode(vals[k, :ncheck]), pkd, seed=0xC0 + 8 * rank + k)
        outs_holder = []

        def step():
            outs_holder[:] = prog.run(ev, ins, rkd, gkd)

        nout = prog.num_outputs()
        unit_bytes = 16 * K * n * (nin + nout)  # compulsory traffic of one program run: read the inputs, write the outputs
        units_per_step = B
        metric, unit = f"fhe_program_{args.workload}_runs_per_sec", "programs/s"
        nmul = sum(1 for op, _ in prog.nodes if op == "Multiply")
        nrot = sum(1 for op, _ in prog.nodes if op in ("ShiftLeft", "ShiftRight", "SwapRows"))
        workload = (f"FheProgram graph examples/{args.workload} ({len(prog.nodes)} nodes: {nmul} mul+relin, {nrot} rotations), n={n}, "
                    f"K={K}+1 SEAL default primes, t={t}, batch={share} input sets")
    else:
        nprimes = 3
        data = uniform_residues((B,), nprimes, primes[:nprimes]).reshape(B * nprimes, n).contiguous()
        ref = data.clone()
        fwd_keep = None if args.no_check else torch.empty((min(64, B) * nprimes, n), dtype=torch.int64, device=dev)

        def step():
            ev.ntt(data, nprimes, inverse=False)
            ev.ntt(data, nprimes, inverse=True)

        unit_bytes = 16 * n  # one single-residue transform: read + write
        units_per_step = 2 * B * nprimes
        total_items = total_items * 2 * nprimes
        metric, unit = "ntt_single_residue_transforms_per_sec", "NTT/s"
        workload = f"batched forward+inverse negacyclic NTT, n={n}, {nprimes} primes ({pset}), batch={share} polys"

    def barrier():
        D.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # Every workload here runs at the package power cap (profiles/r03_power_samples.txt: 1365-1395 W of 1400 W, sclk 1.98-2.15
    # GHz), and the clock the firmware settles on takes tens of milliseconds to reach: W steps of a sub-millisecond workload
    # (the NTT: 0.87 ms) end before that, and the K steps after them were measured 12 % below the sustained rate.  Untimed
    # steps continue for about --settle-ms of device time; the timed region is still exactly K steps.
    settle_steps = 0
    if args.settle_ms > 0 and args.warmup > 0:
        # the number of extra steps is agreed between the ranks (a step may contain a collective): W more steps are timed,
        # their rate sizes the rest, the slowest rank's count is the one every rank runs
        torch.cuda.synchronize()
        t_settle = time.perf_counter()
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        per_step = max((time.perf_counter() - t_settle) / args.warmup, 1e-5)
        more = min(max(int(args.settle_ms * 1e-3 / per_step + 0.999) - args.warmup, 0), 100000)
        if collective:
            mt = torch.tensor([more], dtype=torch.int64, device=cdev)
            dist.all_reduce(mt, op=dist.ReduceOp.MAX)
            more = int(mt.item())
        for _ in range(more):
            step()
        settle_steps = args.warmup + more
    # R timed regions of EXACTLY K steps each, every one bracketed by a barrier + torch.cuda.synchronize() on both sides and
    # reduced with MAX over the ranks; the line reports the MEDIAN region (`values` lists them all).  The HIP-event kernel
    # times accumulate over all R regions.
    repeats = max(1, args.repeats)
    regions = []
    barrier()
    ev.profile(not args.no_kernel_events)
    ev.profile_reset()
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if collective:
            tt = torch.tensor([el], dtype=torch.float64, device=cdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        regions.append(el)
    prof = ev.profile_read()
    ev.profile(False)
    elapsed = sorted(regions)[len(regions) // 2] if len(regions) % 2 else sum(sorted(regions)[len(regions) // 2 - 1 : len(regions) // 2 + 1]) / 2

    # ---- optional result gather (only if the consumer wants every result on one device), timed apart ----
    gather_ms = None
    if args.gather and collective and args.workload == "mulrelin":
        barrier()
        t0 = time.perf_counter()
        full = D.gather_results(out, total_items)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t0)
        barrier()
        if rank == 0:
            assert full.shape[0] == total_items and torch.equal(full[:B], out)
        del full

    # ---- parity gate (after timing so that the timed region is exactly K steps) ----
    parity = "skipped"
    if args.workload == "mulrelin" and nkeys > 1 and not args.no_check:
        # (1) every client's items decrypt under THAT client's secret key to the slot-wise products
        ok = True
        for c, cl in enumerate(clients):
            mine = torch.arange(c, B, nkeys, device=dev)
            dec = ev.decode(ev.decrypt(out[mine].contiguous(), cl["sk"]), signed=True)
            ok = ok and bool(torch.equal(dec, va[mine] * vb[mine]))
        for i in range(K):
            ok = ok and int(out[:, :, i, :].max()) < primes[i] and int(out[:, :, i, :].min()) >= 0
        if collective:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        assert ok, "a per-key multiply+relinearize result does not decrypt to the slot-wise product under its client's key (or is not canonical)"
        # (2) bit-exact vs the oracle on the first `ncheck` items, each relinearised by the oracle with its own client's key
        if rank == 0:
            from concurrent.futures import ThreadPoolExecutor

            ha, hb, got = to_host(a[:ncheck]), to_host(b[:ncheck]), to_host(out[:ncheck])
            hk = {c: clients[c]["rk"].to_array(ctx) for c in sorted({int(key_index[i]) for i in range(ncheck)})}
            with ThreadPoolExecutor(min(ncheck, os.cpu_count() or 1)) as ex:  # the oracle's C calls release the GIL
                refs = list(ex.map(lambda i: o.relinearize(o.multiply(ha[i], hb[i]), hk[int(key_index[i])]), range(ncheck)))
            for i in range(ncheck):
                assert (got[i] == refs[i]).all(), "HIP per-key result differs from the CPU oracle"
        parity = (f"bit-exact vs oracle on {ncheck} items of {len({int(k) for k in key_index[:ncheck]})} clients (each with its own key); all {total_items} "
                  f"results decrypt under their client's secret key to the slot-wise products; all outputs canonical")
    elif args.workload == "mulrelin" and not args.no_check:
        # (1) decrypt-correct on ALL items of every rank (device decryptor, itself bit-exact vs the oracle: tests/test_gpu_client.py)
        dec = ev.decode(ev.decrypt(out, skd), signed=True)
        ok = bool(torch.equal(dec, va * vb))
        # (2) every output word is a canonical residue
        for i in range(K):
            ok = ok and int(out[:, :, i, :].max()) < primes[i] and int(out[:, :, i, :].min()) >= 0
        if collective:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        assert ok, "a multiply+relinearize result does not decrypt to the slot-wise product (or is not canonical)"
        # (3) bit-exact vs the oracle on the first `ncheck` items of rank 0's shard (OpenMP over the items)
        if rank == 0:
            ha, hb, got = to_host(a[:ncheck]), to_host(b[:ncheck]), to_host(out[:ncheck])
            _, refo = o.bench_mul_relin(ha, hb, rk, threads=min(os.cpu_count() or 1, 64))
            assert (got == refo).all(), "HIP result differs from the CPU oracle"
            assert (refo[0] == o.relinearize(o.multiply(ha[0], hb[0]), rk)).all()
            budget = o.noise_budget(got[0], sk)
            assert budget > 0
        parity = f"bit-exact vs oracle on {ncheck} items; all {total_items} results decrypt to the slot-wise products; all outputs canonical"
    elif args.workload == "pir" and not args.no_check:
        if rank == 0:
            res = holder["out"]
            got = to_host(ev.decrypt(res, skd))[0]
            assert int(got[0]) == int(vals[sel_r, sel_c]) and not got[1:].any(), "PIR lookup returned the wrong entry"
            assert (got == o.decrypt(to_host(res)[0], sk)).all()
        parity = f"lookup decrypts to database[{sel_r}][{sel_c}]; decryption bit-exact vs the oracle"
        # the sharded answer against the unsharded lookup, bit for bit, whenever rank 0 can hold the whole database
        full_bytes = rows * cols * K * n * 8
        if world > 1 and full_bytes <= (8 << 30):
            if rank == 0:
                from sunscreen_amd.workloads import pir_lookup

                full_db = torch.empty((rows, cols, K, n), dtype=torch.int64, device=dev)
                rowbuf = torch.zeros((cols, n), dtype=torch.int64, device=dev)
                for i in range(rows):
                    rowbuf[:, 0] = vals[i].to(dev)
                    full_db[i] = ev.plain_to_ntt(rowbuf)
                single = pir_lookup(ev, cq, rq, full_db, rkd)
                assert torch.equal(single, holder["out"]), "row-sharded lookup differs from the single-GPU lookup"
                del full_db
            parity += f"; the {world}-way row-sharded answer equals the single-GPU lookup over the whole database bit for bit"
        else:
            parity += " (matrix-vector and program bits vs the oracle: tests/test_gpu_program.py)"
    elif args.workload == "e2e" and not args.no_check:
        assert torch.equal(holder["out"], (va * vb) % t), "decoded products differ from the slot-wise products"
        if rank == 0:
            got = to_host(holder["ct"][:2])
            pl = to_host(ev.decrypt(holder["ct"][:2], skd))
            for i in range(2):
                assert (pl[i] == o.decrypt(got[i], sk)).all(), "HIP decryption differs from the CPU oracle"
        parity = f"all {B} decoded results equal the slot-wise products mod t; decryption bit-exact vs the oracle on 2 items"
    elif args.workload in ("chi_sq", "dot_prod") and not args.no_check:
        if rank == 0 and ncheck:
            from concurrent.futures import ThreadPoolExecutor

            hin = [to_host(ins[k][:ncheck]) for k in range(nin)]
            hout = [to_host(outs_holder[k][:ncheck]) for k in range(nout)]
            with ThreadPoolExecutor(min(ncheck, os.cpu_count() or 1)) as ex:  # the oracle's C calls release the GIL
                refs = list(ex.map(lambda i: run_program(o, prog.nodes, prog.edges, [h[i] for h in hin], rk, gk), range(ncheck)))
            for i in range(ncheck):
                for k in range(nout):
                    assert (hout[k][i] == refs[i][k]).all(), "HIP program result differs from the CPU oracle"
            # ... and the genuine sets decrypt to the program's value on the slot vectors
            v = [x[:ncheck].to(torch.int64) for x in vals]
            if args.workload == "chi_sq":
                x_, y_ = 2 * v[0] + v[1], 2 * v[2] + v[1]
                expect = [(4 * v[0] * v[2] - v[1] * v[1]) ** 2, 2 * x_ * x_, x_ * y_, 2 * y_ * y_]
            else:
                # examples/dot_prod: every slot of the result holds the sum of the slot-wise products of its row half... the
                # rotate-and-add ladder leaves the full dot product of BOTH rows in every slot after swap_rows + add
                expect = [(v[0] * v[1]).sum(dim=1, keepdim=True).expand(-1, n)]
            for k in range(nout):
                dec = ev.decode(ev.decrypt(outs_holder[k][:ncheck], skd))
                assert torch.equal(dec, expect[k] % t), "program output does not decrypt to the expected slot values"
        parity = f"bit-exact vs the oracle graph interpreter on {ncheck} input sets x {nout} outputs; those sets decrypt to the program's slot values"
    elif args.workload == "ntt" and not args.no_check:
        assert torch.equal(data, ref), "INTT(NTT(x)) != x"
        # forward transform of the first polynomials against the oracle (SEAL NTTTables convention, pinned by the key fixture)
        m = fwd_keep.shape[0]
        fwd_keep.copy_(data[:m])
        ev.ntt(fwd_keep, nprimes, inverse=False)
        if rank == 0:
            oo, href, hgot = O.Oracle(n, primes, t), to_host(ref[:m]), to_host(fwd_keep)
            for i in range(m):  # polynomial i belongs to key prime i % nprimes (the layout of hipbfv_batch_ntt)
                assert (oo.ntt(i % nprimes, href[i]) == hgot[i]).all(), "HIP forward NTT differs from the CPU oracle"
        parity = f"INTT(NTT(x)) == x on the whole batch; NTT(x) bit-exact vs the oracle on {m} polynomials"

    if rank != 0:
        return None

    total_units = total_items * args.steps  # every rank's units of every timed step
    value = total_units / elapsed
    # dominant kernel by accumulated HIP-event time
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
    roofline = None
    valu = None
    op_rate_gbs = unit_bytes * (value / world) / 1e9
    if dom:
        name, rec = dom
        # algorithmic bytes of one launch of that kernel
        S_, R_ = ctx_S(ctx), K + ctx_S(ctx)
        bm, bk = (6 if getattr(ctx, "packed_mul", False) else 8), (6 if getattr(ctx, "packed_ks", False) else 8)
        if getattr(ctx, "packed_mul_rows", False):
            # per-row packing (N = 16384 default set: 13 of 18 rows travel as 6 bytes): the average bytes per value over the K + S rows
            rows_ = list(primes[:K]) + list(ctx.aux_primes)
            bm = sum(6 if p < (1 << 48) else 8 for p in rows_) / len(rows_)
        ba = bk  # the accumulator rows
        if getattr(ctx, "packed_ks_rows", False):  # the rows of T per key prime (3 of 9 at N = 16384), the accumulator rows as doubles
            bk, ba = sum(6 if p < (1 << 48) else 8 for p in primes[:KK]) / KK, 8
        per_unit = {
            # whole-polynomial path (kernels.hip)
            "ntt_fwd": 16 * n, "ntt_inv": 16 * n,                    # per residue polynomial: read + write
            "behz_extend": 8 * n * (K + R_),                         # per polynomial: read K, write K+S residues
            "tensor": 8 * n * 7 * R_,                                # per op: read 4, write 3 extended polys
            "behz_floor_sk": 8 * n * (R_ + K),                       # per polynomial
            "ks_decompose": 8 * n * (K + KK * K), "ks_mac": 8 * n * (KK * K + 2 * KK), "ks_moddown": 8 * n * (2 * KK + 4 * K),
            # split path (kernels_split.hip); units: polynomials for mul_head / mul_tail, ops otherwise
            # (intermediates travel as 6 bytes per value when the context packs them, 8 otherwise: bm / bk)
            "mul_head": n * (8 * K + bm * R_), "mul_mid": bm * n * 7 * R_, "mul_tail": n * (bm * R_ + 8 * K),
            "ks_head": n * (8 * K + bk * KK * K), "ks_mid": n * (bk * KK * K + ba * 2 * KK),
            "ks_tail": n * (ba * 2 * KK + 32 * K),
            "galois": 16 * n * K, "eltwise": 24 * n,                 # per polynomial / per residue polynomial (2 reads + 1 write)
            "plain": 8 * K * n,                                      # dot_plain_ntt, per database entry: its K transform-domain residues
        }.get(name, 16 * n)
        avg_ms = rec["ms"] / rec["launches"]
        kernel_bytes_per_launch = per_unit * rec["units"] / rec["launches"]
        kernel_rate = kernel_bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # SURVEY 8(d) ALGORITHMIC bytes: the per-unit compulsory figure x the units ONE launch of this kernel processes.
        # A unit is one op / program run / database entry (the dominant kernel sees this rank's units / launches of them per
        # launch); for the transform workload the unit is one single-residue transform = the kernel's own work unit.
        units_per_launch = rec["units"] / rec["launches"] if args.workload == "ntt" else units_per_step * args.steps * repeats / rec["launches"]
        achieved = unit_bytes * units_per_launch / (avg_ms * 1e-3) / 1e9
        # measured HBM bytes per launch and VALU issue occupancy from the committed PMC profile of THIS workload
        # (FETCH_SIZE x2 + WRITE_SIZE, SQ_INSTS_VALU*, GRBM_GUI_ACTIVE in separate rocprofv3 --pmc passes:
        # tools/gpu_pmc_report.sh, tools/pmc_traffic.py); PMC needs rocprofv3, so it is not re-measured live -- and it is
        # only reported while the kernels are still the ones the passes ran on (source hash recorded beside the passes)
        traffic = None
        traffic_note = None
        wkey = pmc_workload_key(args, n)
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pmc = doc["workloads"][wkey]["kernels"]
            taken_at = doc.get("source_hash", {}).get(wkey)
            if taken_at != kernel_source_hash():
                traffic_note = (f"profiles/pmc_traffic.json [{wkey}] was measured on kernel sources {taken_at}, this build is {kernel_source_hash()}: "
                                "stale PMC figures are not reported (re-run tools/gpu_pmc_report.sh + tools/pmc_merge.sh)")
            elif name in pmc:
                traffic = int(pmc[name]["hbm_bytes_per_unit"] * rec["units"] / rec["launches"])
                if "valu_issue_frac" in pmc[name]:
                    # SURVEY 8(d) asks for the VALU bound beside the HBM one: this path is FP64-issue-bound before it is
                    # HBM-bound.  Issue cycles per wave64 instruction by class: 4 for FP64, 2 for everything else
                    # (MI355X_MICROARCH.md, wave scheduling) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
                    valu = {"bound": "valu_issue", "kernel": name, "frac": pmc[name]["valu_issue_frac"],
                            "wave_insts_per_launch": int(pmc[name]["valu_wave_insts_per_dispatch"]),
                            "f64_wave_insts_per_launch": int(pmc[name].get("valu_f64_wave_insts_per_dispatch", 0)),
                            "shader_cycles_per_launch": int(pmc[name]["shader_cycles_per_dispatch"]),
                            "source": f"profiles/pmc_traffic.json [{wkey}] (rocprofv3 --pmc passes, kernel sources {taken_at})"}
        except Exception:
            traffic = None
        roofline = {
            "kernel": name,
            "bound": "hbm",
            # the task contract's figure: SURVEY 8(d) bytes of the units ONE launch of the dominant kernel serves / that launch's
            # duration.  It prices one kernel of a multi-kernel pipeline against the whole operation's compulsory bytes, so it
            # is NOT the operation's fraction of the roof: that is `whole_op` below (compulsory bytes x ops/s / peak).
            "scope": "dominant kernel launch",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "avg_launch_ms": round(avg_ms, 4),
            "launches": rec["launches"],
            "algorithmic_bytes_per_unit": unit_bytes,
            "units_per_launch": round(units_per_launch, 3),
            "algorithmic_bytes_per_launch": int(unit_bytes * units_per_launch),
            "whole_op": {"achieved": round(op_rate_gbs, 1), "frac": round(op_rate_gbs / HBM_PEAK_GBS, 4),
                         "definition": "algorithmic_bytes_per_unit x units/s per GPU over ALL kernels of the pipeline (SURVEY 8d: bytes_algorithmic x ops/s / 8e12)"},
            # the kernel's OWN reads and writes (pipeline intermediates included) over the same
            # launch time: what it keeps in flight, not what the operation has to move
            "kernel_hbm": {"bytes_per_launch": int(kernel_bytes_per_launch), "achieved": round(kernel_rate, 1),
                           "frac": round(kernel_rate / HBM_PEAK_GBS, 4)},
        }
        if traffic_note:
            roofline["traffic_note"] = traffic_note
    cpu = None
    if not args.no_cpu and world == 1:
        cpu = cpu_baseline(args, O, n, primes, t, small=secondary)
    power = None
    if args.power and world == 1 and not secondary:
        power = power_leg(step, torch.cuda.synchronize)
    line = {
        "metric": metric,
        "value": round(value, 2),
        "unit": unit,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "settle_steps": settle_steps,
        "repeats": repeats,
        "values": [round(total_units / r, 2) for r in regions],
        "spread": round((max(regions) - min(regions)) / elapsed, 4),
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": scaling_of(args),
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "poly_modulus_degree": n, "coeff_modulus_primes": KK,
                   "plain_modulus": t, "parallelism": (f"database rows sharded x{world}, one cross-GPU sum per query" if args.workload == "pir"
                                                       else f"batch-sharded x{world}"), "chunk_ops": args.chunk or "auto",
                   "collectives": (dist.get_backend() if collective else "none (single process, no process group)")},
        "roofline": roofline,
        "valu": valu,
        "whole_op_hbm": {"algorithmic_bytes_per_unit": unit_bytes, "achieved_GBps_per_gpu": round(op_rate_gbs, 1),
                         "frac_of_peak": round(op_rate_gbs / HBM_PEAK_GBS, 4)},
        "kernels_ms_per_step": {k: round(v["ms"] / (args.steps * repeats), 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        "kernel_units_per_launch": {k: v["units"] / v["launches"] for k, v in prof.items()},
        "cpu_baseline": cpu,
        "parity": parity,
        "kernel_source_hash": kernel_source_hash(),  # the sources this line (and the PMC file it may quote) was measured on
    }
    if power:
        line["power"] = power
    if args.total_batch:
        line["config"]["total_batch"] = args.total_batch
    if nkeys > 1:
        line["config"]["key_sets"] = nkeys
        line["roofline_note"] = ("algorithmic bytes per op = 48 K N (two inputs, one output) + 16 K (K+1) N x distinct keys / batch (every distinct "
                                 "relinearisation key is read at least once per step)")
    if exchange:
        line["config"]["exchange"] = exchange
    if gather_ms is not None:
        line["result_gather_ms"] = round(gather_ms, 3)
    return line


def main():
    args = parse()
    if args.dry_run:
        sys.exit(dry_run(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    env = setup(args)
    import torch
    import torch.distributed as dist

    line = measure(args, env)
    # BASELINE.json's metric also names n=16384 and NTTs/sec: the default (headline) run times them too, in this process
    headline = args.workload == "mulrelin" and args.n == 8192 and not args.coeff_bits and not args.chunk and args.keys <= 1
    if headline and not args.no_secondary:
        second = {}
        q4 = max(args.batch // 4, 1)
        jobs = [
            # the other two quantities of the metric
            ("mulrelin_n16384", dict(n=16384, batch=q4, total_batch=args.total_batch // 4)),
            ("ntt_n8192", dict(workload="ntt", steps=max(args.steps, 100))),  # a step is 0.7 ms: time at least 70 ms per region
            # the north star's literal prime set ("n=8192, 3 x 54-bit RNS primes")
            ("mulrelin_n8192_bits54-54-54-56", dict(coeff_bits="54,54,54,56")),
        ]
        if not args.total_batch:
            # SURVEY 8(d) config 3 "also report the per-ciphertext-key worst case": the same batch with one key set per 64 items, one
            # per item (every item reads its own 2.5 MiB key), and the n = 16384 worst case (18 MiB of key per item)
            jobs += [
                ("mulrelin_n8192_keys64", dict(keys=max(args.batch // 64, 1), no_cpu=True)),
                ("mulrelin_n8192_keys4096", dict(keys=args.batch, no_cpu=True)),
                ("mulrelin_n16384_keys1024", dict(n=16384, batch=q4, keys=q4, no_cpu=True)),
            ]
        if env.world == 1 and not args.total_batch:
            # BASELINE.json configs[3], [4]: the reference's example programs at n = 16384 (single-GPU forms; the sharded forms are
            # `--workload chi_sq --total-batch 1024 --gpus 8` and `--workload pir --n 16384 --batch 1024 --gpus 8`)
            jobs += [
                ("chi_sq_n16384", dict(workload="chi_sq", n=16384, batch=q4)),                       # configs[3]: the 1024-input batch
                ("chi_sq_n16384_share128", dict(workload="chi_sq", n=16384, batch=max(q4 // 8, 1), no_cpu=True)),  # ... one GPU's share of it on 8
                ("dot_prod_n16384", dict(workload="dot_prod", n=16384, batch=max(q4 // 4, 1))),
                # configs[4]: DB = 2^20 entries = a 1024 x 1024 database (the example's database is square) sharded by row over 8 GPUs: ONE GPU's
                # share is 128 rows x 1024 columns = 2^17 entries, 128 GiB -- exactly what `--workload pir --n 16384 --batch 1024 --gpus 8` gives rank r
                # (r01 ... r06 s30 ran 512 x 256 here: the same entries, four times the row-side multiplies a shard of the real database has)
                ("pir_n16384_2p17", dict(workload="pir", n=16384, batch=q4, pir_rows=max(q4 // 8, 1))),
            ]
        failed = False
        for key, over in jobs:
            sub = copy.copy(args)
            for k, v in over.items():
                setattr(sub, k, v)
            import gc

            gc.collect()
            torch.cuda.empty_cache()
            if sub.workload == "pir":
                # the database alone is rows x cols x K x n words in transform form (128 GiB for 128 x 1024 at n = 16384): on a device
                # that does not have that much free, say so instead of running out of memory with the headline unprinted
                from oracle import bfv_oracle as O_

                # + the queries, the executor's chunk scratch for the row-side multiply + relinearize (~28 MB per row at n = 16384) and a margin
                need = (sub.pir_rows or sub.batch) * sub.batch * (len(O_.bfv_default(sub.n)) - 1) * sub.n * 8 + (28 << 30)
                free = torch.cuda.mem_get_info()[0]
                if free < need:
                    second[key] = {"skipped": f"needs {need >> 30} GiB of device memory, {free >> 30} GiB free"}
                    continue
            # a failure in one secondary job (out of memory next to another process, a parity assertion) must not lose the headline
            # measured above: it is recorded under the job's key, the line is printed, and the exit status says so
            try:
                rec = measure(sub, env, secondary=True)
            except Exception as e:  # noqa: BLE001 -- recorded, and the process exits non-zero after printing
                if env.world > 1:
                    raise  # one rank skipping ahead would leave the others in this job's barrier: fail the launch as before
                failed = True
                if env.rank == 0:
                    second[key] = {"error": f"{type(e).__name__}: {e}"[:300], "parity_ok": False}
                continue
            if rec is not None:
                second[key] = rec if args.full_line else compact_secondary(rec)
        if line is not None:
            line["secondary"] = second
            # a job that was skipped (device memory) or failed is named here as well as under its key: a reader of the exit status and
            # this one field knows whether every quantity of the metric was measured
            line["skipped_jobs"] = sorted(k for k, v in second.items() if "value" not in v)
            line["complete"] = not line["skipped_jobs"]
    if line is not None:
        # LAST key: the quantities BASELINE.json's metric names, one small object each (the driver keeps the tail of the line)
        line["summary"] = summary_of(line)
        if not args.full_line:
            trim_headline(line)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()
    if headline and not args.no_secondary and failed:
        sys.exit(1)


SUMMARY_KEYS = {"mulrelin_n8192_keys64": "keys64", "mulrelin_n8192_keys4096": "keys4096", "mulrelin_n16384_keys1024": "n16384_keys1024",
                "mulrelin_n16384": "mulrelin_n16384", "ntt_n8192": "ntt_n8192", "mulrelin_n8192_bits54-54-54-56": "3x54", "chi_sq_n16384": "chi_sq_1024",
                "chi_sq_n16384_share128": "chi_sq_128", "dot_prod_n16384": "dot_prod", "pir_n16384_2p17": "pir_2p17"}


def summary_entry(rec):
    """{value, ms_per_step, frac, whole_op_frac, traffic_ratio, parity_ok} of one measured workload."""
    if "value" not in rec:
        return {k: rec[k] for k in ("error", "skipped") if k in rec} | {"parity_ok": False}
    roof = rec.get("roofline") or {}
    whole = roof.get("whole_op", {}).get("frac", roof.get("whole_op_frac"))
    ratio = roof.get("traffic_ratio")
    if ratio is None and roof.get("traffic") and roof.get("algorithmic_bytes_per_launch"):
        ratio = round(roof["traffic"] / roof["algorithmic_bytes_per_launch"], 3)
    parity = rec.get("parity", "")
    return {"value": rec["value"], "ms_per_step": rec["ms_per_step"], "frac": roof.get("frac"), "whole_op_frac": whole, "traffic_ratio": ratio,
            "parity_ok": rec.get("parity_ok", parity.startswith(("bit-exact", "INTT(NTT(x)) == x", "lookup decrypts", "all ")))}


def summary_of(line):
    name = "mulrelin_n8192" if line["config"].get("poly_modulus_degree") == 8192 and line["metric"] == "bfv_mul_relin_ops_per_sec" else "headline"
    if line["config"].get("key_sets"):  # a stand-alone --keys run
        name = f"mulrelin_n{line['config'].get('poly_modulus_degree')}_keys{line['config']['key_sets']}"
    out = {name: summary_entry(line)}
    for key, rec in (line.get("secondary") or {}).items():
        out[SUMMARY_KEYS.get(key, key)] = summary_entry(rec)
    return out


def compact_secondary(rec):
    """A secondary workload's record without the prose (profiles/README.md explains every field; `--full-line` prints everything):
    what was run, the measurement, the roofline figures, the CPU baseline and whether the parity gate passed."""
    roof = rec.get("roofline") or {}
    cpu = rec.get("cpu_baseline") or {}
    kern = sorted((rec.get("kernels_ms_per_step") or {}).items(), key=lambda kv: -kv[1])[:3]
    out = {"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
           "spread": rec["spread"],
           "config": {"n": rec["config"].get("poly_modulus_degree"), "primes": rec["config"].get("coeff_modulus_primes"), "batch": rec["config"].get("batch_per_gpu")},
           "roofline": {"kernel": roof.get("kernel"), "bound": roof.get("bound"), "achieved": roof.get("achieved"), "frac": roof.get("frac"),
                        "traffic": roof.get("traffic"), "avg_launch_ms": roof.get("avg_launch_ms"),
                        "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch"), "whole_op_frac": roof.get("whole_op", {}).get("frac")},
           "valu_issue_frac": (rec.get("valu") or {}).get("frac"),
           "kernels_ms_per_step": dict(kern),
           "cpu_baseline": ({"value": cpu.get("value"), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind")} if cpu else None),
           "parity_ok": summary_entry(rec)["parity_ok"]}
    return out


def trim_headline(line):
    """The headline keeps every field of the contract; the explanatory strings inside roofline / valu go (they are in profiles/README.md)."""
    roof = line.get("roofline") or {}
    if "whole_op" in roof:
        roof["whole_op"].pop("definition", None)
    if line.get("valu"):
        line["valu"].pop("source", None)


def pmc_workload_key(args, n):
    """Key of this run's workload in profiles/pmc_traffic.json (the PMC passes are per workload)."""
    k = f"{args.workload}_n{n}"
    if getattr(args, "keys", 1) > 1:
        k += f"_keys{args.keys}"
    if args.coeff_bits:
        k += "_bits" + args.coeff_bits.replace(",", "-")
    return k


def ctx_S(ctx):
    # |Bsk| of the context's auxiliary base (hipbfv_Context_AuxBase)
    return len(ctx.aux_primes)


def cpu_baseline(args, O, n, primes, t, small=False):
    """The CPU oracle (a port of SEAL's algorithms, NOT SEAL itself -- SEAL's source is absent from the
    reference tree) timed on this host on a bounded sample of the same workload.  small: the secondary measurements of the
    default run take a quarter of the sample so that the whole run stays within a minute."""
    cores = os.cpu_count() or 1
    threads = min(cores, 256)  # all host cores; the sample scales with them so that every thread gets several items
    o = O.Oracle(n, primes, t)
    rng = np.random.default_rng(1)
    K = o.K
    if args.workload == "mulrelin":
        O.seed(99)
        sk, pk, rk, _ = o.keygen()
        sample = args.cpu_sample or max(threads * (8 if n <= 8192 else 4) // (4 if small else 1), 64)
        sample = max(64, min(sample, (6 << 30) // (3 * 2 * K * n * 8)))  # operands + results of the sample stay below 6 GB of host memory
        a = np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2 * sample)]).reshape(sample, 2, K, n)
        b = a[::-1].copy()
        secs1, _ = o.bench_mul_relin(a[: max(8, sample // threads)], b[: max(8, sample // threads)], rk, threads=1)
        one = max(8, sample // threads) / secs1
        # OpenMP over the batch on every host core AND on 64 threads: the 256-CPU boxes run this memory-bound kernel faster
        # on a quarter of their hardware threads (measured 840 vs 1700 ops/s); the better of the two is the baseline
        trials = {}
        for th in sorted({threads, min(threads, 64)}):
            secs, _ = o.bench_mul_relin(a, b, rk, threads=th)
            trials[th] = sample / secs
        best = max(trials, key=trials.get)
        return {"value": round(trials[best], 2), "unit": "ops/s", "cores": best, "kind": "port",
                "sample": f"{sample} mul+relin ops (same parameters) with OpenMP over the batch; threads -> ops/s: "
                          + ", ".join(f"{th} -> {v:.0f}" for th, v in sorted(trials.items()))
                          + f"; single-thread rate {one:.2f} ops/s on {max(8, sample // threads)} ops",
                "single_thread_value": round(one, 2), "host_cpus": cores}
    threads = min(threads, 64)  # the secondary workloads keep the thread count round 1 measured them with
    if args.workload == "pir":
        from concurrent.futures import ThreadPoolExecutor

        O.seed(96)
        sk, pk, rk, _ = o.keygen()
        rows, cols = threads, 16  # a bounded slab of the same computation: rows x cols entries + one mul+relin per row
        plains = np.zeros((rows, cols, n), dtype=np.uint64)
        plains[:, :, 0] = rng.integers(1, 1000, (rows, cols))
        zero = np.zeros(n, dtype=np.uint64)
        cq = [o.encrypt(pk, zero) for _ in range(cols)]
        rq = [o.encrypt(pk, zero) for _ in range(rows)]

        def one_row(i):
            col = o.multiply_plain(cq[0], plains[i, 0])
            for j in range(1, cols):
                col = o.add(col, o.multiply_plain(cq[j], plains[i, j]))
            return o.relinearize(o.multiply(col, rq[i]), rk)

        t0 = time.perf_counter()
        one_row(0)
        single = cols / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one_row, range(rows)))
        secs = time.perf_counter() - t0
        return {"value": round(rows * cols / secs, 1), "unit": "entries/s", "cores": threads, "kind": "port",
                "sample": f"{rows} database rows x {cols} columns (multiply_plain + add per entry, one mul+relin per row, node by node as "
                          f"run.rs does) on {threads} host threads; single-thread rate {single:.1f} entries/s",
                "single_thread_value": round(single, 1), "host_cpus": cores}
    if args.workload == "e2e":
        from concurrent.futures import ThreadPoolExecutor

        O.seed(97)
        sk, pk, rk, _ = o.keygen()
        sample = args.cpu_sample or threads
        vals = rng.integers(0, 257, (sample, 2, n)).astype(np.uint64)

        def one(v):
            ca, cb = o.encrypt(pk, o.batch_encode(v[0])), o.encrypt(pk, o.batch_encode(v[1]))
            return o.batch_decode(o.decrypt(o.relinearize(o.multiply(ca, cb), rk), sk))

        t0 = time.perf_counter()
        one(vals[0])
        single = 1.0 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(one, vals))
        secs = time.perf_counter() - t0
        return {"value": round(sample / secs, 2), "unit": "ops/s", "cores": threads, "kind": "port",
                "sample": f"{sample} encode+encrypt x2 / mul+relin / decrypt+decode pipelines on {threads} host threads; "
                          f"single-thread rate {single:.2f} ops/s",
                "single_thread_value": round(single, 2), "host_cpus": cores}
    if args.workload in ("chi_sq", "dot_prod"):
        from concurrent.futures import ThreadPoolExecutor

        from oracle.program_interp import run_program
        from sunscreen_amd.workloads import chi_sq_optimized, dot_product

        O.seed(98)
        if args.workload == "chi_sq":
            prog, nin, elts = chi_sq_optimized(), 3, None
        else:
            lanes = n // 2
            prog, nin = dot_product(lanes), 2
            elts = sorted({o.galois_elt_from_step(1 << i) for i in range(lanes.bit_length() - 1)} | {2 * n - 1})
        sk, pk, rk, gk = o.keygen(galois_elts=elts)
        sample = args.cpu_sample or threads
        ins = [[np.stack([np.stack([rng.integers(0, primes[i], n, dtype=np.uint64) for i in range(K)]) for _ in range(2)]) for _ in range(nin)]
               for _ in range(sample)]
        t0 = time.perf_counter()
        run_program(o, prog.nodes, prog.edges, ins[0], rk, gk)
        one = 1.0 / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        with ThreadPoolExecutor(threads) as ex:  # the oracle's C calls release the GIL
            list(ex.map(lambda x: run_program(o, prog.nodes, prog.edges, x, rk, gk), ins))
        secs = time.perf_counter() - t0
        return {"value": round(sample / secs, 3), "unit": "programs/s", "cores": threads, "kind": "port",
                "sample": f"{sample} program runs on {threads} host threads (one oracle call per graph node, as run.rs does); "
                          f"single-thread rate {one:.3f} programs/s",
                "single_thread_value": round(one, 3), "host_cpus": cores}
    nprimes = 3
    sample = args.cpu_sample or threads * (64 if small else 256)
    x = np.stack([rng.integers(0, primes[i % nprimes], n, dtype=np.uint64) for i in range(sample)])
    secs1, _ = o.bench_ntt(x[: sample // threads], nprimes, threads=1)
    secs, _ = o.bench_ntt(x, nprimes, threads=threads)
    return {"value": round(2 * sample / secs, 2), "unit": "NTT/s", "cores": threads, "kind": "port",
            "sample": f"{sample} polynomials forward+inverse on {threads} threads",
            "single_thread_value": round(2 * (sample // threads) / secs1, 2), "host_cpus": cores}


if __name__ == "__main__":
    main()
