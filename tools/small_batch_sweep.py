#!/usr/bin/env python3
"""tools/small_batch_sweep.py <n> -- time per call of the batched multiply / relinearize / rotation for small batch sizes
(1 ... 256 ciphertexts), to place the switch between the whole-polynomial pipelines (more, smaller workgroups: lower latency
for a few ciphertexts) and the head / middle / tail pipelines (less HBM traffic: higher throughput).  Run once as is and once
with HIPBFV_NO_SPLIT_MUL=1 HIPBFV_NO_SPLIT_KS=1; prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bfv_oracle as O  # noqa: E402
from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys  # noqa: E402
from sunscreen_amd.batch import BatchEvaluator  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
primes, t = O.bfv_default(n), O.plain_batching(n, 17)
o = O.Oracle(n, primes, t)
O.seed(5)
elt = o.galois_elt_from_step(1)
sk, pk, rk, gk = o.keygen(galois_elts=[elt])
ctx = Context.from_raw(n, primes, t)
ev = BatchEvaluator(ctx)
ev.set_transparent_check(False)
rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
gen = torch.Generator(device="cuda:0")
gen.manual_seed(1)
B = 256
a = torch.empty((B, 2, ctx.K, n), dtype=torch.int64, device="cuda:0")
for i, q in enumerate(primes[: ctx.K]):
    a[:, :, i, :] = torch.randint(0, q, (B, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
b = a.flip(0).contiguous()
m3 = ev.multiply(a, b)
res = {"n": n, "env": {k: v for k, v in os.environ.items() if k.startswith("HIPBFV_")}}


def timed(f, reps=60):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
        torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / reps * 1e6, 1)


for c in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    res[str(c)] = {
        "multiply_us": timed(lambda: ev.multiply(a[:c], b[:c])),
        "relinearize_us": timed(lambda: ev.relinearize(m3[:c], rkd)),
        "multiply_relin_us": timed(lambda: ev.multiply_relin(a[:c], b[:c], rkd)),
        "rotate_us": timed(lambda: ev.apply_galois(a[:c], elt, gkd)),
    }
print(json.dumps(res))
