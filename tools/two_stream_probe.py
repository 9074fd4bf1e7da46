#!/usr/bin/env python3
"""tools/two_stream_probe.py [n] [batch] -- does the device overlap two mul+relin pipelines issued on two streams (two
evaluators, two host threads, half the batch each) better than one pipeline over the whole batch?  (Round 3 probe for
co-residency of the middle kernels with head / tail kernels of another chunk.)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    from oracle import bfv_oracle as O
    from sunscreen_amd import Context, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator
    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t); O.seed(3)
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    evs = [BatchEvaluator(ctx) for _ in range(2)]
    rkd = RelinearizationKeys.from_array(ctx, rk)
    K = ctx.K
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    def rnd():
        x = torch.empty((B, 2, K, n), dtype=torch.int64, device="cuda:0")
        for i in range(K): x[:, :, i, :] = torch.randint(0, primes[i], (B, 2, n), generator=g, device="cuda:0", dtype=torch.int64)
        return x
    a, b = rnd(), rnd(); out = torch.empty_like(a)
    def one(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): evs[0].multiply_relin(a, b, rkd, out=out)
        torch.cuda.synchronize(); return B * steps / (time.perf_counter() - t0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    h = B // 2
    def two(steps, pieces=2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        def work(i):
            with torch.cuda.stream(streams[i]):
                for _ in range(steps):
                    evs[i].multiply_relin(a[i*h:(i+1)*h], b[i*h:(i+1)*h], rkd, out=out[i*h:(i+1)*h])
        th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [x.start() for x in th]; [x.join() for x in th]
        torch.cuda.synchronize(); return B * steps / (time.perf_counter() - t0)
    one(2); two(2)
    ref = out.clone(); two(1); assert torch.equal(ref, out)
    r = [(one(5), two(5)) for _ in range(3)]
    print("n", n, "batch", B, "one stream ops/s", [round(x[0]) for x in r], "two streams", [round(x[1]) for x in r])
main()
