#!/usr/bin/env python3
"""tools/pcie_inclusive.py -- the headline operation with its operands in HOST memory: what a caller that hands the batched C ABI
host buffers gets when it stages them itself (bench.py's `value` has the inputs resident in HBM; this is the PCIe-inclusive rate
DESIGN.md section 9 quotes beside it, measured instead of estimated).

Per op at n = 8192, K = 4: two ciphertexts in (2 x 512 KiB), one out (512 KiB).  Three legs, all on pinned host memory:
  copies   -- H2D and D2H alone (GB/s of the link as this box gives it to hipMemcpyAsync)
  serial   -- copy in, hipbfv_batch_multiply_relin, copy out, one chunk after the other on one stream
  pipeline -- the same chunks on three streams (in / compute / out) with events, three chunks in flight
Only the library is used (its own key generator; operands are uniform canonical residues: valid ciphertext bit patterns).
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from sunscreen_amd import _lib
    from sunscreen_amd.batch import BatchEvaluator
    from sunscreen_amd.seal import CoefficientModulus, Context, KeyGenerator

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    reps = 5
    assert B % chunk == 0
    _lib.load().hipbfv_set_device(0)
    torch.cuda.set_device(0)
    primes = [int(m.value()) for m in CoefficientModulus.bfv_default(n)]
    ctx = Context.from_raw(n, primes, 114689 if n == 8192 else 786433)
    K = ctx.K
    ev = BatchEvaluator(ctx)
    rk = KeyGenerator(ctx, seed=7).create_relinearization_keys()

    rng = np.random.default_rng(3)
    one = np.stack([rng.integers(0, primes[i], size=(chunk, 2, n), dtype=np.int64) for i in range(K)], axis=2)  # [chunk, 2, K, n]
    ha = torch.empty((B, 2, K, n), dtype=torch.int64).pin_memory()
    hb = torch.empty((B, 2, K, n), dtype=torch.int64).pin_memory()
    ho = torch.empty((B, 2, K, n), dtype=torch.int64).pin_memory()
    for c in range(B // chunk):
        ha[c * chunk:(c + 1) * chunk] = torch.from_numpy(one)
        hb[c * chunk:(c + 1) * chunk] = torch.from_numpy(np.roll(one, c + 1, axis=0))
    ct_bytes = 2 * K * n * 8
    nbuf = 3
    da = [torch.empty((chunk, 2, K, n), dtype=torch.int64, device="cuda:0") for _ in range(nbuf)]
    db = [torch.empty_like(da[0]) for _ in range(nbuf)]
    do = [torch.empty_like(da[0]) for _ in range(nbuf)]

    def timed(f):
        f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    # ---- the link alone ----
    dbig = torch.empty((B, 2, K, n), dtype=torch.int64, device="cuda:0")
    t_h2d = timed(lambda: dbig.copy_(ha, non_blocking=True))
    t_d2h = timed(lambda: ho.copy_(dbig, non_blocking=True))
    s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s_in):
            dbig.copy_(ha, non_blocking=True)
        with torch.cuda.stream(s_out):
            ho.copy_(dbig, non_blocking=True)

    t_bidir = timed(both)
    del dbig

    # ---- resident (bench.py's condition) at this chunk size and at the whole batch ----
    def resident_chunks():
        for c in range(B // chunk):
            ev.multiply_relin(da[0], db[0], rk, out=do[0])

    da[0].copy_(ha[:chunk]); db[0].copy_(hb[:chunk])
    t_res_chunk = timed(resident_chunks)

    # ---- serial: one stream ----
    def serial():
        for c in range(B // chunk):
            sl = slice(c * chunk, (c + 1) * chunk)
            da[0].copy_(ha[sl], non_blocking=True)
            db[0].copy_(hb[sl], non_blocking=True)
            ev.multiply_relin(da[0], db[0], rk, out=do[0])
            ho[sl].copy_(do[0], non_blocking=True)

    t_serial = timed(serial)

    # ---- pipeline: in / compute / out streams, nbuf chunks in flight ----
    s_cmp = torch.cuda.Stream()

    def pipeline():
        free = [None] * nbuf  # event: the D2H of the chunk that last used buffer set i is done
        for c in range(B // chunk):
            i = c % nbuf
            sl = slice(c * chunk, (c + 1) * chunk)
            with torch.cuda.stream(s_in):
                if free[i] is not None:
                    s_in.wait_event(free[i])
                da[i].copy_(ha[sl], non_blocking=True)
                db[i].copy_(hb[sl], non_blocking=True)
                e_in = torch.cuda.Event()
                e_in.record(s_in)
            with torch.cuda.stream(s_cmp):
                s_cmp.wait_event(e_in)
                ev.multiply_relin(da[i], db[i], rk, out=do[i])
                e_c = torch.cuda.Event()
                e_c.record(s_cmp)
            with torch.cuda.stream(s_out):
                s_out.wait_event(e_c)
                ho[sl].copy_(do[i], non_blocking=True)
                free[i] = torch.cuda.Event()
                free[i].record(s_out)

    t_pipe = timed(pipeline)
    # the pipelined result equals the serial one (same library call on the same bits)
    ref = ho.clone()
    serial()
    torch.cuda.synchronize()
    same = bool(torch.equal(ref, ho))
    ev.check()

    res = {
        "n": n, "K": K, "batch": B, "chunk": chunk, "bytes_per_op_in": 2 * ct_bytes, "bytes_per_op_out": ct_bytes,
        "h2d_GBps": round(B * ct_bytes / t_h2d / 1e9, 2), "d2h_GBps": round(B * ct_bytes / t_d2h / 1e9, 2),
        "bidirectional_GBps_each_way": round(B * ct_bytes / t_bidir / 1e9, 2),
        "resident_ops_per_s_at_this_chunk": round(B / t_res_chunk, 1),
        "serial_host_fed_ops_per_s": round(B / t_serial, 1),
        "pipelined_host_fed_ops_per_s": round(B / t_pipe, 1),
        "link_bound_ops_per_s": round(1.0 / max(2 * ct_bytes / (B * ct_bytes / t_h2d), ct_bytes / (B * ct_bytes / t_d2h)), 1),
        "pipelined_equals_serial_bits": same,
    }
    print(json.dumps(res))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
