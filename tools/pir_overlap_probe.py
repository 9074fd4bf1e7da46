#!/usr/bin/env python3
"""tools/pir_overlap_probe.py [rows cols] -- does the device overlap examples/pir's HBM-bound plaintext-product stream with the VALU-bound
row-side multiply + relinearize when the two run on different streams?  (VERDICT r04 #8.)  n = 16384, database rows x cols in transform
form, split into row chunks: serial = product(chunk) then mul+relin(chunk) on one stream; pipelined = product(chunk i + 1) on stream A
while mul+relin(chunk i) runs on stream B.  Same primitives, same bits (checked)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import bfv_oracle as O
from sunscreen_amd import Context, RelinearizationKeys
from sunscreen_amd.batch import BatchEvaluator, to_device

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = 16384
primes = O.bfv_default(n)
t = O.plain_batching(n, 17)
o = O.Oracle(n, primes, t)
O.seed(5)
sk, pk, rk, _ = o.keygen()
ctx = Context.from_raw(n, primes, t)
ev = BatchEvaluator(ctx)
rkd = RelinearizationKeys.from_array(ctx, rk)
K = o.K
dev = torch.device("cuda:0")
zero = np.zeros(n, dtype=np.uint64)
one_ct = to_device(np.stack([o.encrypt(pk, zero)]))
cq = one_ct.repeat(cols, 1, 1, 1).contiguous()
rq = one_ct.repeat(rows, 1, 1, 1).contiguous()
db = torch.empty((rows, cols, K, n), dtype=torch.int64, device=dev)
rowbuf = torch.zeros((cols, n), dtype=torch.int64, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
for i in range(rows):
    rowbuf[:, 0] = torch.randint(1, 1000, (cols,), device=dev, generator=g)
    db[i] = ev.plain_to_ntt(rowbuf)
ctn = ev.ct_to_ntt(cq)
per = rows // chunks
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def serial():
    outs = []
    for c in range(chunks):
        col = ev.dot_plain_ntt(ctn, db[c * per:(c + 1) * per])
        outs.append(ev.multiply_relin(col, rq[c * per:(c + 1) * per], rkd))
    return outs


def pipelined():
    outs, cols_, evs = [], [], []
    with torch.cuda.stream(sa):
        for c in range(chunks):
            cols_.append(ev.dot_plain_ntt(ctn, db[c * per:(c + 1) * per]))
            e = torch.cuda.Event()
            e.record(sa)
            evs.append(e)
    with torch.cuda.stream(sb):
        for c in range(chunks):
            sb.wait_event(evs[c])
            outs.append(ev.multiply_relin(cols_[c], rq[c * per:(c + 1) * per], rkd))
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    return outs


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best.append(1e3 * (time.perf_counter() - t0))
    return sorted(best)[len(best) // 2], r


for rnd in range(2):
    ms_s, rs = timed(serial)
    ms_p, rp = timed(pipelined)
    same = all(torch.equal(a, b) for a, b in zip(rs, rp))
    print(f"rows {rows} cols {cols} chunks {chunks}: serial {ms_s:.2f} ms, two streams {ms_p:.2f} ms ({100 * (ms_s / ms_p - 1):+.1f} %), same bits {same}")
