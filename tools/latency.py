#!/usr/bin/env python3
"""tools/latency.py -- drop-in (handle-level, one ciphertext per call) latency of the SEAL-named C ABI on the GPU box:
what a `seal_fhe::Evaluator` user sees per FFI call when nothing is batched (sunscreen_runtime/src/run.rs:237-282 issues
exactly these calls one node at a time).  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    reps = 200
    from oracle import bfv_oracle as O
    from sunscreen_amd import BFVEvaluator, Ciphertext, Context, GaloisKeys, RelinearizationKeys

    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(5)
    sk, pk, rk, gk = o.keygen(galois_elts=[3])
    ctx = Context.from_raw(n, primes, t)
    ev = BFVEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    gkd = GaloisKeys.from_arrays(ctx, gk)
    a = Ciphertext.from_array(ctx, o.encrypt(pk, o.batch_encode(np.arange(n, dtype=np.uint64) % 7)))
    b = Ciphertext.from_array(ctx, o.encrypt(pk, o.batch_encode(np.arange(n, dtype=np.uint64) % 5)))

    def timed(f):
        for _ in range(10):
            f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return (time.perf_counter() - t0) / reps * 1e6

    res = {
        "n": n,
        "multiply_us": timed(lambda: ev.multiply(a, b)),
        "multiply_relinearize_us": timed(lambda: ev.relinearize(ev.multiply(a, b), rkd)),
        "rotate_rows_1_us": timed(lambda: ev.rotate_rows(a, 1, gkd)),
        "add_us": timed(lambda: ev.add(a, b)),
    }
    # several host threads on ONE evaluator handle, each with its own operands (run.rs:415-469 dispatches ready nodes from a
    # rayon pool): every thread has its own stream inside the library, so the calls overlap on the device.  Python threads
    # release the GIL inside the ctypes calls.
    import threading

    def threaded(nthreads, seconds=1.5):
        cts = [(Ciphertext.from_array(ctx, a.to_array()), Ciphertext.from_array(ctx, b.to_array())) for _ in range(nthreads)]
        counts = [0] * nthreads
        stop = time.perf_counter() + seconds

        def work(i):
            x, y = cts[i]
            while time.perf_counter() < stop:
                ev.relinearize(ev.multiply(x, y), rkd)
                counts[i] += 1

        ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        return sum(counts) / (time.perf_counter() - t0)

    threaded(2, 0.3)
    res["multiply_relinearize_ops_per_s_by_threads"] = {str(k): round(threaded(k), 1) for k in (1, 2, 4, 8, 16, 32)}
    # the same on the CPU oracle (single thread, like one SEAL Evaluator call)
    ca, cb = a.to_array(), b.to_array()
    t0 = time.perf_counter()
    for _ in range(5):
        o.relinearize(o.multiply(ca, cb), rk)
    res["cpu_oracle_multiply_relinearize_us"] = (time.perf_counter() - t0) / 5 * 1e6
    got = ev.relinearize(ev.multiply(a, b), rkd).to_array()
    assert (got == o.relinearize(o.multiply(ca, cb), rk)).all()
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
