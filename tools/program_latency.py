#!/usr/bin/env python3
"""tools/program_latency.py [n] -- latency of ONE input set (the reference's call shape: FheRuntime::run takes one set of
arguments, sunscreen_runtime/src/run.rs:100-357) through the graph executor, scheduled vs node by node
(HIPBFV_PROGRAM_SERIAL=1), for the reference's example programs; plus small batches.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    from oracle import bfv_oracle as O
    from oracle.program_interp import run_program
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host
    from sunscreen_amd.workloads import chi_sq_optimized, dot_product

    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(5)
    lanes = n // 2
    elts = sorted({o.galois_elt_from_step(1 << i) for i in range(lanes.bit_length() - 1)} | {2 * n - 1})
    sk, pk, rk, gk = o.keygen(galois_elts=elts)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
    rng = np.random.default_rng(0)
    res = {"n": n, "unit": "microseconds per program run (all input sets of the batch)"}
    for name, prog, nin in (("chi_sq", chi_sq_optimized(), 3), ("dot_prod", dot_product(lanes), 2)):
        for batch in (1, 4, 16):
            vals = rng.integers(0, 7, (nin, batch, n)).astype(np.uint64)
            cts = [to_device(np.stack([o.encrypt(pk, o.batch_encode(v)) for v in vals[a]])) for a in range(nin)]
            out = {}
            for mode in ("scheduled", "node_by_node"):
                os.environ["HIPBFV_PROGRAM_SERIAL"] = "1" if mode == "node_by_node" else "0"
                bound = prog.prepare(ev, cts, rkd, gkd)
                for _ in range(5):
                    got = bound()
                torch.cuda.synchronize()
                reps = 30
                t0 = time.perf_counter()
                for _ in range(reps):
                    got = bound()
                torch.cuda.synchronize()
                out[mode] = round((time.perf_counter() - t0) / reps * 1e6, 1)
                out[mode + "_bits"] = [to_host(g) for g in got]
            for a_, b_ in zip(out.pop("scheduled_bits"), out.pop("node_by_node_bits")):
                assert (a_ == b_).all()
            if batch == 1:
                ref = run_program(o, prog.nodes, prog.edges, [to_host(c)[0] for c in cts], rk, gk)
                got = prog.run(ev, cts, rkd, gkd)
                for k in range(len(ref)):
                    assert (to_host(got[k])[0] == ref[k]).all()
            res[f"{name}_batch{batch}"] = out
    os.environ.pop("HIPBFV_PROGRAM_SERIAL", None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
