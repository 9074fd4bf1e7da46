#!/usr/bin/env python3
"""bench.py -- BFV ct x ct multiply + relinearize throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: `--batch` (default 4096) independent
ciphertext pairs, n=8192, SEAL default 128-bit parameters (K=4 data primes + 1 special prime,
t = batching(8192,17) = 114689) -- BASELINE.json configs[2], the configuration the metric is quoted
on.  Inputs are resident in HBM before the timed region.  `--workload ntt` runs configs[1]
(batched forward+inverse NTT, n=8192, 3 primes, 4096 polynomials) instead; chi_sq / dot_prod / pir / e2e run the
reference's example programs and the client-side steps (secondary workloads, same JSON contract).

The CPU oracle appears here in three roles only: client (it generates the keys and the few genuine encryptions the
parity gate needs -- the library has no key generator), checker (parity gate, after the timed region) and
`cpu_baseline` (all host cores, OpenMP over the batch).  Nothing in the timed region touches it.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL); every rank processes its own
`--batch` items (weak scaling, no data-path collective); time = max over ranks.  Under torchrun the ranks come from the
environment (WORLD_SIZE must equal --gpus); a bare `python bench.py --gpus N` spawns the N ranks itself.  Rank 0 owns the
keys and broadcasts them once (sunscreen_amd.dist.replicate_keys); `--gather` additionally times the optional all_gather
of the results (reported apart from `value`).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4096, help="ciphertext pairs (or polynomials for --workload ntt) per GPU per step")
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--workload", choices=["mulrelin", "ntt", "chi_sq", "dot_prod", "e2e", "pir"], default="mulrelin",
                    help="mulrelin = the headline; ntt = batched transforms; chi_sq / dot_prod = whole program graphs "
                         "(examples/chi_sq, examples/dot_prod) through the batch graph executor (SURVEY 8d configs 4 / 5b); "
                         "e2e = encode + encrypt both operands, multiply + relinearize, decrypt + decode, all on the device; "
                         "pir = examples/pir lookup over a (--batch x --batch) plaintext database held in transform form (SURVEY 8d config 5a)")
    ap.add_argument("--coeff-bits", default="", help="comma-separated prime sizes (CoeffModulus::create, last = special prime) instead of the "
                    "SEAL default set for --n, e.g. 54,54,54,56 for the 3 x 54-bit n=8192 variant BASELINE.json mentions")
    ap.add_argument("--chunk", type=int, default=0, help="override the executor's chunk size (ops per launch group)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="ops in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--check-items", type=int, default=64, help="mulrelin: items compared bit for bit with the oracle (BASELINE.md section 3: >= 64)")
    ap.add_argument("--gather", action="store_true", help="N>1: also time one all_gather of the result batch (reported as result_gather_ms, never part of value)")
    return ap.parse_args()


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
    import time

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


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
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
    dev = f"cuda:{local_rank}"
    if world > 1:
        dist.init_process_group(os.environ.get("HIPBFV_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    from sunscreen_amd import Context, RelinearizationKeys, _lib
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    _lib.load().hipbfv_set_device(local_rank)
    # the oracle is the checker and the CPU baseline only
    from oracle import bfv_oracle as O

    n = args.n
    primes = O.coeff_modulus_create(n, [int(b) for b in args.coeff_bits.split(",")]) if args.coeff_bits else O.bfv_default(n)
    pset = ("CoeffModulus::create(" + args.coeff_bits + ")") if args.coeff_bits else "SEAL default 128-bit"
    t = O.plain_batching(n, 17)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    if args.chunk:
        ev.set_chunk_ops(args.chunk)
    K, KK = ctx.K, ctx.KK
    B = args.batch
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EA10001 + rank)

    def uniform_residues(shape_prefix, nres, primes_):
        # uniform canonical residues per prime: valid ciphertext / polynomial bit patterns
        out = torch.empty(shape_prefix + (nres, n), dtype=torch.int64, device=dev)
        for i in range(nres):
            out[..., i, :] = torch.randint(0, primes_[i % len(primes_)], shape_prefix + (n,), generator=gen, device=dev, dtype=torch.int64)
        return out

    result = {}
    if args.workload == "mulrelin":
        from sunscreen_amd import PublicKey, SecretKey
        from sunscreen_amd import dist as D

        o = O.Oracle(n, primes, t)
        # rank 0 is the key owner (the oracle stands in for the client's key generator); the other ranks receive the
        # SEAL-format bytes once and build their device copies from them -- the deployment form, SURVEY 8(e)
        rk = sk = None
        rkd = skd = pkd = None
        if rank == 0:
            O.seed(0xBF5 + 17)
            sk, pk, rk, _ = o.keygen()
            rkd, skd, pkd = RelinearizationKeys.from_array(ctx, rk), SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)
        cdev = dev if (world > 1 and dist.get_backend() == "nccl") else "cpu"
        rkd = D.replicate_keys(ctx, rkd, RelinearizationKeys, 0, cdev)
        skd = D.replicate_keys(ctx, skd, SecretKey, 0, cdev)
        pkd = D.replicate_keys(ctx, pkd, PublicKey, 0, cdev)
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
        workload = f"BFV ct*ct multiply+relinearize, n={n}, K={K}+1 {pset} primes, t={t}, batch={B} pairs/GPU"
    elif args.workload == "e2e":
        from sunscreen_amd import PublicKey, SecretKey

        o = O.Oracle(n, primes, t)
        O.seed(0xE2E + 17)
        sk, pk, rk, _ = o.keygen()
        rkd = RelinearizationKeys.from_array(ctx, rk)
        skd, pkd = SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)
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
                    f"t={t}, batch={B} slot-vector pairs/GPU")
    elif args.workload == "pir":
        from sunscreen_amd import PublicKey, SecretKey
        from sunscreen_amd.workloads import pir_lookup

        side = B  # sqrt(database size): --batch rows x --batch columns of plaintext entries
        o = O.Oracle(n, primes, t)
        O.seed(0x914 + 17)
        sk, pk, rk, _ = o.keygen()
        rkd = RelinearizationKeys.from_array(ctx, rk)
        skd, pkd = SecretKey.from_array(ctx, sk), PublicKey.from_array(ctx, pk)
        vals = torch.randint(1, 1000, (side, side), generator=gen, device=dev, dtype=torch.int64)
        db_ntt = torch.empty((side, side, K, n), dtype=torch.int64, device=dev)
        for i in range(side):  # scalar entries (value in coefficient 0), transformed once: the server's static state
            row = torch.zeros((side, n), dtype=torch.int64, device=dev)
            row[:, 0] = vals[i]
            db_ntt[i] = ev.plain_to_ntt(row)
        sel_r, sel_c = (7 + rank) % side, (side // 3 + rank) % side
        onehot_c = torch.zeros((side, n), dtype=torch.int64, device=dev)
        onehot_c[sel_c, 0] = 1
        onehot_r = torch.zeros((side, n), dtype=torch.int64, device=dev)
        onehot_r[sel_r, 0] = 1
        cq = ev.encrypt(onehot_c, pkd, seed=0xC0 + rank)
        rq = ev.encrypt(onehot_r, pkd, seed=0xD0 + rank)
        holder = {}

        def step():
            holder["out"] = pir_lookup(ev, cq, rq, db_ntt, rkd)

        unit_bytes = 8 * K * n  # compulsory traffic per database entry: its transform-domain residues, read once
        units_per_step = side * side
        metric, unit = "pir_db_entries_per_sec", "entries/s"
        workload = (f"examples/pir lookup: {side}x{side} plaintext database in transform form ({side * side * K * n * 8 / 2**30:.1f} GiB), "
                    f"one encrypted query per step, n={n}, K={K}+1 SEAL default primes, t={t}")
    elif args.workload in ("chi_sq", "dot_prod"):
        from sunscreen_amd import GaloisKeys
        from sunscreen_amd.workloads import chi_sq_optimized, dot_product
        from oracle.program_interp import run_program

        o = O.Oracle(n, primes, t)
        O.seed(0xC415 + 17)
        if args.workload == "chi_sq":
            prog, nin, elts = chi_sq_optimized(), 3, None
            lanes = 0
        else:
            lanes = n // 2
            prog, nin = dot_product(lanes), 2
            elts = sorted({o.galois_elt_from_step(1 << i) for i in range(lanes.bit_length() - 1)} | {2 * n - 1})
        sk, pk, rk, gk = o.keygen(galois_elts=elts)
        rkd = RelinearizationKeys.from_array(ctx, rk)
        gkd = GaloisKeys.from_arrays(ctx, gk) if gk else None
        ins = [uniform_residues((B, 2), K, primes) for _ in range(nin)]
        ncheck = 0 if args.no_check else 2
        rng = np.random.default_rng(rank)
        vals = rng.integers(0, 7, (nin, ncheck, n)).astype(np.uint64)
        enc = [np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]]) if ncheck else None for a in range(nin)]
        for a in range(nin):
            if ncheck:
                ins[a][:ncheck] = to_device(enc[a], dev)
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
                    f"K={K}+1 SEAL default primes, t={t}, batch={B} input sets/GPU")
    else:
        nprimes = 3
        data = uniform_residues((B,), nprimes, primes[:nprimes]).reshape(B * nprimes, n).contiguous()
        ref = data.clone()

        def step():
            ev.ntt(data, nprimes, inverse=False)
            ev.ntt(data, nprimes, inverse=True)

        unit_bytes = 16 * n  # one single-residue transform: read + write
        units_per_step = 2 * B * nprimes
        metric, unit = "ntt_single_residue_transforms_per_sec", "NTT/s"
        workload = f"batched forward+inverse negacyclic NTT, n={n}, {nprimes} primes ({pset}), batch={B} polys/GPU"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev.profile(True)
    ev.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    prof = ev.profile_read()
    ev.profile(False)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- optional result gather (only if the consumer wants every result on one device), timed apart ----
    gather_ms = None
    if args.gather and world > 1 and args.workload == "mulrelin":
        from sunscreen_amd import dist as D

        barrier()
        t0 = time.perf_counter()
        full = D.gather_results(out, B * world)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - t0)
        assert full.shape[0] == B * world and torch.equal(full[rank * B : (rank + 1) * B], out)
        del full

    # ---- parity gate (after timing so that the timed region is exactly K steps) ----
    parity = "skipped"
    if args.workload == "mulrelin" and not args.no_check:
        # (1) decrypt-correct on ALL items of every rank (device decryptor, itself bit-exact vs the oracle: tests/test_gpu_client.py)
        dec = ev.decode(ev.decrypt(out, skd), signed=True)
        ok = bool(torch.equal(dec, va * vb))
        # (2) every output word is a canonical residue
        for i in range(K):
            ok = ok and int(out[:, :, i, :].max()) < primes[i] and int(out[:, :, i, :].min()) >= 0
        if world > 1:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
        assert ok, "a multiply+relinearize result does not decrypt to the slot-wise product (or is not canonical)"
        # (3) bit-exact vs the oracle on the first `ncheck` items of rank 0's shard (OpenMP over the items)
        if rank == 0:
            ha, hb, got = to_host(a[:ncheck]), to_host(b[:ncheck]), to_host(out[:ncheck])
            _, ref = o.bench_mul_relin(ha, hb, rk, threads=min(os.cpu_count() or 1, 64))
            assert (got == ref).all(), "HIP result differs from the CPU oracle"
            assert (ref[0] == o.relinearize(o.multiply(ha[0], hb[0]), rk)).all()
            budget = o.noise_budget(got[0], sk)
            assert budget > 0
        parity = f"bit-exact vs oracle on {ncheck} items; all {B * world} results decrypt to the slot-wise products; all outputs canonical"
    elif args.workload == "pir" and not args.no_check:
        got = to_host(ev.decrypt(holder["out"], skd))[0]
        assert int(got[0]) == int(vals[sel_r, sel_c]) and not got[1:].any(), "PIR lookup returned the wrong entry"
        assert (got == o.decrypt(to_host(holder["out"])[0], sk)).all()
        parity = f"lookup decrypts to database[{sel_r}][{sel_c}]; decryption bit-exact vs the oracle (matrix-vector bits: tests/test_gpu_program.py)"
    elif args.workload == "e2e" and not args.no_check:
        assert torch.equal(holder["out"], (va * vb) % t), "decoded products differ from the slot-wise products"
        got = to_host(holder["ct"][:2])
        pl = to_host(ev.decrypt(holder["ct"][:2], skd))
        for i in range(2):
            assert (pl[i] == o.decrypt(got[i], sk)).all(), "HIP decryption differs from the CPU oracle"
        parity = f"all {B} decoded results equal the slot-wise products mod t; decryption bit-exact vs the oracle on 2 items"
    elif args.workload in ("chi_sq", "dot_prod") and not args.no_check:
        for i in range(ncheck):
            refs = run_program(o, prog.nodes, prog.edges, [e[i] for e in enc], rk, gk)
            for k in range(nout):
                assert (to_host(outs_holder[k][i : i + 1])[0] == refs[k]).all(), "HIP program result differs from the CPU oracle"
        parity = f"bit-exact vs the oracle graph interpreter on {ncheck} input sets x {nout} outputs"
    elif args.workload == "ntt" and not args.no_check:
        assert torch.equal(data, ref), "INTT(NTT(x)) != x"
        parity = "INTT(NTT(x)) == x on the whole batch"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_units = units_per_step * args.steps * world
    value = total_units / elapsed
    # dominant kernel by accumulated HIP-event time
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
    roofline = None
    if dom:
        name, rec = dom
        # algorithmic bytes of one launch of that kernel
        S_, R_ = ctx_S(ctx), K + ctx_S(ctx)
        bm, bk = (6 if getattr(ctx, "packed_mul", False) else 8), (6 if getattr(ctx, "packed_ks", False) else 8)
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
            "ks_head": n * (8 * K + bk * KK * K), "ks_mid": n * bk * (KK * K + 2 * KK),
            "ks_tail": n * (bk * 2 * KK + 32 * K),
            "galois": 16 * n * K, "eltwise": 24 * n,                 # per polynomial / per residue polynomial (2 reads + 1 write)
            "plain": 8 * K * n,                                      # dot_plain_ntt, per database entry: its K transform-domain residues
        }.get(name, 16 * n)
        avg_ms = rec["ms"] / rec["launches"]
        kernel_bytes_per_launch = per_unit * rec["units"] / rec["launches"]
        kernel_rate = kernel_bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # SURVEY 8(d) ALGORITHMIC bytes: the per-unit compulsory figure x the units ONE launch of this kernel processes.
        # A unit is one op / program run / database entry (the dominant kernel sees total_units / launches of them per
        # launch); for the transform workload the unit is one single-residue transform = the kernel's own work unit.
        units_per_launch = rec["units"] / rec["launches"] if args.workload == "ntt" else units_per_step * args.steps / rec["launches"]
        achieved = unit_bytes * units_per_launch / (avg_ms * 1e-3) / 1e9
        # measured HBM bytes per launch and VALU issue occupancy from the committed PMC profile of THIS workload
        # (FETCH_SIZE x2 + WRITE_SIZE, SQ_INSTS_VALU*, GRBM_GUI_ACTIVE in separate rocprofv3 --pmc passes:
        # tools/gpu_pmc_report.sh, tools/pmc_traffic.py); PMC needs rocprofv3, so it is not re-measured live
        traffic = None
        valu = None
        wkey = pmc_workload_key(args, n)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["workloads"][wkey]["kernels"]
            if name in pmc:
                traffic = int(pmc[name]["hbm_bytes_per_unit"] * rec["units"] / rec["launches"])
                if "valu_issue_frac" in pmc[name]:
                    # SURVEY 8(d) asks for the VALU bound beside the HBM one: this path is FP64-issue-bound before it is
                    # HBM-bound.  Issue cycles per wave64 instruction by class: 4 for FP64, 2 for everything else
                    # (MI355X_MICROARCH.md, wave scheduling) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
                    valu = {"bound": "valu_issue", "kernel": name, "frac": pmc[name]["valu_issue_frac"],
                            "wave_insts_per_launch": int(pmc[name]["valu_wave_insts_per_dispatch"]),
                            "f64_wave_insts_per_launch": int(pmc[name].get("valu_f64_wave_insts_per_dispatch", 0)),
                            "shader_cycles_per_launch": int(pmc[name]["shader_cycles_per_dispatch"]),
                            "source": f"profiles/pmc_traffic.json [{wkey}] (rocprofv3 --pmc passes)"}
        except Exception:
            traffic = None
        roofline = {
            "kernel": name,
            "bound": "hbm",
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
            # the kernel's OWN reads and writes (pipeline intermediates included) over the same
            # launch time: what it keeps in flight, not what the operation has to move
            "kernel_hbm": {"bytes_per_launch": int(kernel_bytes_per_launch), "achieved": round(kernel_rate, 1),
                           "frac": round(kernel_rate / HBM_PEAK_GBS, 4)},
        }
    op_rate_gbs = unit_bytes * (value / world) / 1e9
    cpu = None
    if not args.no_cpu and world == 1:
        cpu = cpu_baseline(args, O, n, primes, t)
    line = {
        "metric": metric,
        "value": round(value, 2),
        "unit": unit,
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": workload, "batch_per_gpu": B, "poly_modulus_degree": n, "coeff_modulus_primes": KK,
                   "plain_modulus": t, "parallelism": f"batch-sharded x{world}", "chunk_ops": args.chunk or "auto"},
        "roofline": roofline,
        "valu": valu if dom else None,
        "whole_op_hbm": {"algorithmic_bytes_per_unit": unit_bytes, "achieved_GBps_per_gpu": round(op_rate_gbs, 1),
                         "frac_of_peak": round(op_rate_gbs / HBM_PEAK_GBS, 4)},
        "kernels_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        "kernel_units_per_launch": {k: v["units"] / v["launches"] for k, v in prof.items()},
        "cpu_baseline": cpu,
        "parity": parity,
    }
    if gather_ms is not None:
        line["result_gather_ms"] = round(gather_ms, 3)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def pmc_workload_key(args, n):
    """Key of this run's workload in profiles/pmc_traffic.json (the PMC passes are per workload)."""
    k = f"{args.workload}_n{n}"
    if args.coeff_bits:
        k += "_bits" + args.coeff_bits.replace(",", "-")
    return k


def ctx_S(ctx):
    # |Bsk| of the context's auxiliary base (hipbfv_Context_AuxBase)
    return len(ctx.aux_primes)


def cpu_baseline(args, O, n, primes, t):
    """The CPU oracle (a port of SEAL's algorithms, NOT SEAL itself -- SEAL's source is absent from the
    reference tree) timed on this host on a bounded sample of the same workload."""
    cores = os.cpu_count() or 1
    threads = min(cores, 256)  # all host cores; the sample scales with them so that every thread gets several items
    o = O.Oracle(n, primes, t)
    rng = np.random.default_rng(1)
    K = o.K
    if args.workload == "mulrelin":
        O.seed(99)
        sk, pk, rk, _ = o.keygen()
        sample = args.cpu_sample or max(threads * (8 if n <= 8192 else 4), 64)
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
    sample = args.cpu_sample or threads * 256
    x = np.stack([rng.integers(0, primes[i % nprimes], n, dtype=np.uint64) for i in range(sample)])
    secs1, _ = o.bench_ntt(x[: sample // threads], nprimes, threads=1)
    secs, _ = o.bench_ntt(x, nprimes, threads=threads)
    return {"value": round(2 * sample / secs, 2), "unit": "NTT/s", "cores": threads, "kind": "port",
            "sample": f"{sample} polynomials forward+inverse on {threads} threads",
            "single_thread_value": round(2 * (sample // threads) / secs1, 2), "host_cpus": cores}


if __name__ == "__main__":
    main()
