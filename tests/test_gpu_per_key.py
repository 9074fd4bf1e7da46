"""Per-key batches on the GPU (SURVEY 8d config 3 "per-ciphertext-key worst case").

The reference passes the keys with every call (sunscreen_runtime/src/run.rs:100-105: `relin_keys`, `galois_keys`;
runtime.rs:310-327), so a server that batches the calls of many clients holds one key set per client.  Every test
calls libhipbfv.so through the C ABI (hipbfv_batch_*_keys, hipbfv_Program_RunKeys) with SEVERAL key sets in one call
and checks every item bit for bit against the CPU oracle run with that item's own keys.
"""
import os

import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import oracle_for, params
from tests.oracle_program import run_program

pytestmark = pytest.mark.gpu


def _tenants(name, ntenants, galois=None, seed=100, relin=True):
    """`ntenants` independent key sets (secret, public, relinearisation, Galois) of one parameter set."""
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator

    n, primes, t = params(name) if isinstance(name, str) else name
    o = oracle_for(name) if isinstance(name, str) else O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    ten = []
    for k in range(ntenants):
        O.seed(seed + k)
        sk, pk, rk, gk = o.keygen(galois_elts=galois)
        ten.append(
            {
                "sk": sk,
                "pk": pk,
                "rk": rk,
                "gk": gk,
                "rkd": RelinearizationKeys.from_array(ctx, rk) if relin and rk is not None else None,
                "gkd": GaloisKeys.from_arrays(ctx, gk) if gk else None,
            }
        )
    return o, ctx, ev, ten


def _enc(o, ten, key_index, rng, hi=50):
    vals = rng.integers(0, hi, (len(key_index), o.n)).astype(np.uint64)
    try:
        cts = np.stack([o.encrypt(ten[k]["pk"], o.batch_encode(v % o.t)) for v, k in zip(vals, key_index)])
    except ValueError:  # parameters without batching
        cts = np.stack([o.encrypt(ten[k]["pk"], v % o.t) for v, k in zip(vals, key_index)])
    return vals, cts


@pytest.mark.parametrize(
    "name,ntenants,count,chunk",
    [
        ("default_4096_16", 9, 26, 0),  # 9 key sets (one of them unused), scrambled items, one launch
        ("default_4096_16", 8, 21, 5),  # chunks of 5: every chunk sorts its own items
        ("default_8192_17", 8, 10, 0),  # the headline degree (fused five-kernel pipeline at count > 8)
        ("default_16384_17", 3, 5, 0),  # K = 8
    ],
)
def test_multiply_relinearize_with_one_key_set_per_client(name, ntenants, count, chunk):
    """>= 8 distinct relinearisation keys in ONE call; unsorted key indices; every item equals the oracle run with its key."""
    from sunscreen_amd.batch import to_device, to_host

    o, ctx, ev, ten = _tenants(name, ntenants)
    rng = np.random.default_rng(ntenants * 1000 + count)
    used = ntenants - 1 if ntenants == 9 else ntenants
    key_index = rng.integers(0, used, count).astype(np.uint32)
    key_index[:used] = rng.permutation(used)  # every used key set appears
    if chunk:
        ev.set_chunk_ops(chunk)
    va, a = _enc(o, ten, key_index, rng)
    vb, b = _enc(o, ten, key_index, rng)
    sets = [x["rkd"] for x in ten]
    da, db = to_device(a), to_device(b)
    fused = to_host(ev.multiply_relin_keys(da, db, sets, key_index))
    prod = ev.multiply(da, db)
    relin = to_host(ev.relinearize_keys(prod, sets, key_index))
    square = to_host(ev.multiply_relin_keys(da, da, sets, key_index))
    for i in range(count):
        k = int(key_index[i])
        ref = o.relinearize(o.multiply(a[i], b[i]), ten[k]["rk"])
        assert (fused[i] == ref).all(), (name, i, k)
        assert (relin[i] == ref).all(), (name, i, k)
        assert (square[i] == o.relinearize(o.multiply(a[i], a[i]), ten[k]["rk"])).all(), (name, i, k)
        if o.t % (2 * o.n) == 1:
            assert (o.batch_decode(o.decrypt(fused[i], ten[k]["sk"])) == (va[i] * vb[i]) % o.t).all()
    # another client's key gives other bits: the selection is not ignored
    other = (int(key_index[0]) + 1) % used
    assert not (fused[0] == o.relinearize(o.multiply(a[0], b[0]), ten[other]["rk"])).all()


def test_per_key_call_with_one_key_equals_the_single_key_call():
    """All items on key set 0 of several = hipbfv_batch_multiply_relin with that key, word for word (a large batch: the walk
    order of the key-switch kernel differs between the two calls, the bits must not)."""
    import torch
    from sunscreen_amd.batch import to_device

    o, ctx, ev, ten = _tenants("default_8192_17", 2)
    rng = np.random.default_rng(77)
    count = 67  # not a multiple of 8: the last XCD's run of the walk is short
    _, a = _enc(o, ten, [0] * 4, rng)
    _, b = _enc(o, ten, [0] * 4, rng)
    da = to_device(a).repeat(17, 1, 1, 1)[:count].contiguous()
    db = to_device(b).repeat(17, 1, 1, 1)[:count].contiguous()
    db[5:] = db[5:].roll(1, 0)  # (not 17 copies of the same four products)
    single = ev.multiply_relin(da, db, ten[0]["rkd"])
    sets = [ten[1]["rkd"], ten[0]["rkd"]]
    perkey = ev.multiply_relin_keys(da, db, sets, np.ones(count, dtype=np.uint32))
    mixed = ev.multiply_relin_keys(da, db, sets, (np.arange(count) % 2).astype(np.uint32))
    single1 = ev.multiply_relin(da, db, ten[1]["rkd"])
    torch.cuda.synchronize()
    assert torch.equal(single, perkey)
    odd = torch.arange(count, device=mixed.device) % 2 == 1
    assert torch.equal(mixed[odd], single[odd]) and torch.equal(mixed[~odd], single1[~odd])


def _shape_trials():
    # HIPBFV_FUZZ_PERKEY_TRIALS="lo:hi": extended campaigns (profiles/r06_s38_*); the suite's own are 0:6
    lo, hi = (int(x) for x in os.environ.get("HIPBFV_FUZZ_PERKEY_TRIALS", "0:6").split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("trial", _shape_trials())
def test_per_key_random_shapes(trial):
    """Random tenant counts, batch sizes, chunk sizes and key assignments (runs of one key, every eighth item, unused key sets,
    counts that are no multiple of 8: the per-XCD walk of the key-switch kernels): every item of the per-key calls equals the
    single-key call with that item's key word for word, and two items per trial equal the oracle."""
    import torch
    from sunscreen_amd.batch import to_device, to_host

    rng = np.random.default_rng(9000 + trial)
    name = ("default_4096_16", "default_8192_17", "default_4096_16", "default_16384_17")[trial % 4]
    ntenants = int(rng.integers(1, 4 if "16384" in name else 10))
    count = int(rng.integers(1, 24 if "16384" in name else 90))
    o, ctx, ev, ten = _tenants(name, ntenants, seed=300 + trial)
    chunk = int(rng.choice([0, 0, 3, 8, 17]))
    if chunk:
        ev.set_chunk_ops(chunk)
    kind = trial % 3
    if kind == 0:
        key_index = rng.integers(0, ntenants, count)
    elif kind == 1:
        key_index = np.sort(rng.integers(0, ntenants, count))  # runs of one key
    else:
        key_index = np.arange(count) % ntenants  # neighbours never share a key
    key_index = key_index.astype(np.uint32)
    n, primes = o.n, params(name)[1]
    K = o.K
    a = np.stack([rng.integers(0, pr, (count, 2, n), dtype=np.uint64) for pr in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, pr, (count, 2, n), dtype=np.uint64) for pr in primes[:K]], axis=2)
    da, db = to_device(a), to_device(b)
    sets = [x["rkd"] for x in ten]
    fused = ev.multiply_relin_keys(da, db, sets, key_index)
    relin = ev.relinearize_keys(ev.multiply(da, db), sets, key_index)
    torch.cuda.synchronize()
    assert torch.equal(fused, relin), (trial, name, ntenants, count, chunk)
    ki = torch.from_numpy(key_index.astype(np.int64)).to(fused.device)
    for k in range(ntenants):
        sel = ki == k
        if not bool(sel.any()):
            continue
        single = ev.multiply_relin(da[sel].contiguous(), db[sel].contiguous(), sets[k])
        assert torch.equal(fused[sel], single), (trial, name, ntenants, count, chunk, k)
    got = to_host(fused)
    for i in {0, count - 1}:
        assert (got[i] == o.relinearize(o.multiply(a[i], b[i]), ten[int(key_index[i])]["rk"])).all(), (trial, name, i)


def test_per_key_small_batches_take_the_whole_polynomial_pipeline():
    """A few items (<= 8 at n = 8192 under the product default) run through ks_mac instead of the split kernels: same contract."""
    from sunscreen_amd.batch import to_device, to_host

    o, ctx, ev, ten = _tenants("default_8192_17", 3)
    rng = np.random.default_rng(8)
    key_index = np.array([2, 0, 1, 2], dtype=np.uint32)
    _, a = _enc(o, ten, key_index, rng)
    _, b = _enc(o, ten, key_index, rng)
    got = to_host(ev.multiply_relin_keys(to_device(a), to_device(b), [x["rkd"] for x in ten], key_index))
    for i, k in enumerate(key_index):
        assert (got[i] == o.relinearize(o.multiply(a[i], b[i]), ten[int(k)]["rk"])).all(), i


def test_per_key_integer_policy_primes_3x54():
    """The north star's literal 3 x 54-bit set: integer-policy key primes (ks_mid_int_kernel) with per-item keys."""
    from sunscreen_amd.batch import to_device, to_host

    n = 8192
    primes = O.coeff_modulus_create(n, [54, 54, 54, 56])
    o, ctx, ev, ten = _tenants((n, primes, O.plain_batching(n, 17)), 4)
    rng = np.random.default_rng(54)
    key_index = np.array([3, 1, 0, 2, 1, 3, 0, 2, 2, 1], dtype=np.uint32)
    _, a = _enc(o, ten, key_index, rng)
    _, b = _enc(o, ten, key_index, rng)
    got = to_host(ev.multiply_relin_keys(to_device(a), to_device(b), [x["rkd"] for x in ten], key_index))
    for i, k in enumerate(key_index):
        assert (got[i] == o.relinearize(o.multiply(a[i], b[i]), ten[int(k)]["rk"])).all(), i


@pytest.mark.parametrize("name,count", [("default_4096_16", 11), ("default_8192_17", 6)])
def test_rotations_with_one_galois_key_set_per_client(name, count):
    """>= 4 Galois key sets in one call: direct keys, SEAL's NAF chain over power-of-two keys, the column swap."""
    from sunscreen_amd.batch import to_device, to_host

    o, ctx, ev, ten = _tenants(name, 5, galois="all", relin=False)
    rng = np.random.default_rng(count)
    key_index = rng.integers(0, 5, count).astype(np.uint32)
    key_index[:5] = rng.permutation(5)
    _, a = _enc(o, ten, key_index, rng)
    da = to_device(a)
    sets = [x["gkd"] for x in ten]
    for steps in (1, -3, 7):  # 1: a key every set holds; -3, 7: NAF chains (4 - 1, 8 - 1) over the power-of-two keys
        got = to_host(ev.rotate_rows_keys(da, steps, sets, key_index))
        for i, k in enumerate(key_index):
            assert (got[i] == o.rotate_rows(a[i], steps, ten[int(k)]["gk"])).all(), (steps, i)
    got = to_host(ev.rotate_columns_keys(da, sets, key_index))
    for i, k in enumerate(key_index):
        assert (got[i] == o.rotate_columns(a[i], ten[int(k)]["gk"])).all(), i
    elt = 9  # 3^2 mod 2n: the Galois element of a rotation by 2
    got = to_host(ev.apply_galois_keys(da, elt, sets, key_index))
    for i, k in enumerate(key_index):
        assert (got[i] == o.apply_galois(a[i], elt, ten[int(k)]["gk"])).all(), i


def test_program_run_with_one_key_set_per_input_set():
    """hipbfv_Program_RunKeys: the dot-product graph (multiply + relinearize, rotations, adds: examples/dot_prod/src/main.rs:38-75)
    over a batch whose input sets belong to 8 clients; merged launches number their items member-major and must still pick
    each input set's keys.  Both executors."""
    import os

    from sunscreen_amd.batch import to_device, to_host
    from sunscreen_amd.workloads import chi_sq_optimized, dot_product

    o, ctx, ev, ten = _tenants("default_4096_16", 8, galois="all")
    rng = np.random.default_rng(21)
    batch = 11
    key_index = rng.integers(0, 8, batch).astype(np.uint32)
    key_index[:8] = rng.permutation(8)
    prog = dot_product(o.n // 2)
    va, ca = _enc(o, ten, key_index, rng, hi=4)
    vb, cb = _enc(o, ten, key_index, rng, hi=4)
    rks, gks = [x["rkd"] for x in ten], [x["gkd"] for x in ten]
    for serial in ("0", "1"):
        os.environ["HIPBFV_PROGRAM_SERIAL"] = serial
        try:
            (out,) = prog.run(ev, [to_device(ca), to_device(cb)], rks, gks, key_index=key_index)
        finally:
            os.environ.pop("HIPBFV_PROGRAM_SERIAL", None)
        out = to_host(out)
        for i, k in enumerate(key_index):
            (ref,) = run_program(o, prog.nodes, prog.edges, [ca[i], cb[i]], ten[int(k)]["rk"], ten[int(k)]["gk"])
            assert (out[i] == ref).all(), (serial, i)
            dot = int((va[i].astype(np.int64) * vb[i].astype(np.int64)).sum()) % o.t
            assert (o.batch_decode(o.decrypt(out[i], ten[int(k)]["sk"])) == dot).all()
    # chi_sq (examples/chi_sq/src/main.rs:59-88): several multiply+relinearize members per round -> merged launches
    # (r06 s27: merged at every batch size -- 40 input sets: 3 x 40 and 2 x 40 items per launch, the key of item m * 40 + j is input set j's)
    prog = chi_sq_optimized()
    for key_index in (np.array([4, 7, 0, 4, 2], dtype=np.uint32), rng.integers(0, 8, 40).astype(np.uint32)):
        cts = [_enc(o, ten, key_index, rng, hi=7)[1] for _ in range(3)]
        outs = [to_host(t) for t in prog.run(ev, [to_device(c) for c in cts], rks, None, key_index=key_index)]
        for i, k in enumerate(key_index):  # every input set against the oracle under ITS client's keys
            ref = run_program(o, prog.nodes, prog.edges, [c[i] for c in cts], ten[int(k)]["rk"])
            for j in range(4):
                assert (outs[j][i] == ref[j]).all(), (len(key_index), i, j)


def test_per_key_argument_errors():
    """key_index out of range: E_INVALIDARG before anything is launched; a key set without the key: the missing-key error of the
    single-key call; a key object of another context counts as missing."""
    from sunscreen_amd import Context, RelinearizationKeys
    from sunscreen_amd.batch import to_device
    from sunscreen_amd.seal import HipBfvError

    o, ctx, ev, ten = _tenants("default_4096_16", 2)
    rng = np.random.default_rng(1)
    _, a = _enc(o, ten, [0, 1, 0], rng)
    da = to_device(a)
    sets = [x["rkd"] for x in ten]
    with pytest.raises(HipBfvError):
        ev.multiply_relin_keys(da, da, sets, np.array([0, 2, 1], dtype=np.uint32))
    n, primes, t = params("default_4096_16")
    other_ctx = Context.from_raw(n, primes, t)
    foreign = RelinearizationKeys.from_array(other_ctx, ten[0]["rk"])
    with pytest.raises(HipBfvError):
        ev.multiply_relin_keys(da, da, [sets[0], foreign], np.array([0, 1, 0], dtype=np.uint32))
    # and the call still works afterwards
    ev.multiply_relin_keys(da, da, sets, np.array([0, 1, 0], dtype=np.uint32))


def test_concurrent_clients_with_their_own_keys_on_one_evaluator():
    """The drop-in shape of a multi-tenant server: host threads (sunscreen_runtime's rayon pool, run.rs:415-469) call ONE
    evaluator's handle-level relinearize / rotate, each thread with its own client's keys.  Calls that meet in flight are
    combined into one per-key batch (capi.cpp: requests of one kind combine across keys); every result must be the oracle's for
    that client's key, whoever it was combined with."""
    import threading

    from sunscreen_amd import BFVEvaluator, Ciphertext

    o, ctx, bev, ten = _tenants("default_4096_16", 6, galois=[3])
    ev = BFVEvaluator(ctx)
    rng = np.random.default_rng(66)
    nthreads, iters = 6, 8
    vals = rng.integers(0, 30, (nthreads, 2, o.n)).astype(np.uint64)
    cts = [[o.encrypt(ten[i]["pk"], o.batch_encode(vals[i, j])) for j in range(2)] for i in range(nthreads)]
    exp_relin = [o.relinearize(o.multiply(c[0], c[1]), ten[i]["rk"]) for i, c in enumerate(cts)]
    exp_rot = [o.rotate_rows(c[0], 1, ten[i]["gk"]) for i, c in enumerate(cts)]
    errors = []
    start = threading.Barrier(nthreads)

    def worker(i):
        try:
            a, b = Ciphertext.from_array(ctx, cts[i][0]), Ciphertext.from_array(ctx, cts[i][1])
            start.wait()
            for _ in range(iters):
                m = ev.multiply(a, b)
                ev.relinearize_inplace(m, ten[i]["rkd"])
                if not (m.to_array() == exp_relin[i]).all():
                    errors.append((i, "relinearize mismatch"))
                r = ev.rotate_rows(a, 1, ten[i]["gkd"])
                if not (r.to_array() == exp_rot[i]).all():
                    errors.append((i, "rotation mismatch"))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    assert not errors, errors
