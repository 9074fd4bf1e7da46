import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
from tests.bfv_helpers import params
from sunscreen_amd import Context
from sunscreen_amd.batch import BatchEvaluator
for name, B in (("default_8192_17", 4096), ("default_16384_17", 1024)):
    n, primes, t = params(name)
    ctx = Context.from_raw(n, primes, t); ev = BatchEvaluator(ctx)
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(1)
    a = torch.empty((B, 2, ctx.K, n), dtype=torch.int64, device="cuda:0")
    for i, q in enumerate(primes[:ctx.K]): a[:, :, i, :] = torch.randint(0, q, (B, 2, n), generator=gen, device="cuda:0", dtype=torch.int64)
    pl = torch.randint(0, t, (B, n), generator=gen, device="cuda:0", dtype=torch.int64)
    out = torch.empty_like(a)
    for shared in (False, True):
        p = pl[0] if shared else pl
        ev.multiply_plain(a, p, out); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5): ev.multiply_plain(a, p, out)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 5
        print(name, "shared" if shared else "per-op", f"{B/dt/1e3:.1f} K ct*plain/s", f"{dt*1e3:.2f} ms")
