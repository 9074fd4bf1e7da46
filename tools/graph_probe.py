#!/usr/bin/env python3
"""tools/graph_probe.py [n] -- single-ciphertext multiply + relinearize: launched kernel by kernel vs replayed from a captured
hipGraph (hipbfv_debug_graph_probe).  Prints one JSON object."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    from oracle import bfv_oracle as O
    from sunscreen_amd import Context, RelinearizationKeys, _lib
    from sunscreen_amd.batch import to_device, to_host
    import torch

    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(11)
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    a = o.encrypt(pk, o.batch_encode(np.arange(n, dtype=np.uint64) % 7))
    b = o.encrypt(pk, o.batch_encode(np.arange(n, dtype=np.uint64) % 5))
    da, db = to_device(a[None]), to_device(b[None])
    out = torch.empty_like(da)
    L = _lib.load()
    kp = C.POINTER(C.c_uint64)()
    assert L.hipbfv_KSwitchKeys_DevicePtr(rkd.get_handle(), 0, C.byref(kp)) == 0
    kp = C.cast(kp, C.c_void_p)
    res = {"n": n}
    for sel in ("0", "1"):
        os.environ["HIPBFV_NO_SMALL_BATCH"] = sel
        d, g = C.c_double(), C.c_double()
        hr = L.hipbfv_debug_graph_probe(ctx.get_handle(), C.c_void_p(da.data_ptr()), C.c_void_p(db.data_ptr()), kp, C.c_void_p(out.data_ptr()), 300, C.byref(d), C.byref(g))
        assert hr == 0, hex(hr & 0xFFFFFFFF)
        torch.cuda.synchronize()
        assert (to_host(out)[0] == o.relinearize(o.multiply(a, b), rk)).all()
        res["whole_polynomial_pipelines" if sel == "0" else "split_pipelines"] = {"kernel_by_kernel_us": round(d.value, 1), "hipgraph_replay_us": round(g.value, 1)}
    print(json.dumps(res))


main()
