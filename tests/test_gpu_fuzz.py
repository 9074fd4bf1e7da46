"""Deterministic parameter fuzzing: a sweep of (degree, prime count, prime sizes, plain modulus) far outside the
default sets, every evaluator operation compared bit for bit with the oracle.  Exercises the code-path selection
(integer vs FP64 policy per modulus, own vs SEAL auxiliary base, split vs whole-polynomial pipelines, 4- / 8- / 16-prime
kernel instantiations, fast vs generic plain lift)."""
import os

import numpy as np
import pytest

from oracle import bfv_oracle as O

pytestmark = pytest.mark.gpu


def _configs():
    # HIPBFV_FUZZ_SEED / HIPBFV_FUZZ_COUNT: extended campaigns (profiles/r02_v3_fuzz_campaign.txt); the defaults are the suite's
    rng = np.random.default_rng(int(os.environ.get("HIPBFV_FUZZ_SEED", "20260925")))
    out = []
    for _ in range(int(os.environ.get("HIPBFV_FUZZ_COUNT", "40"))):
        n = int(rng.choice([1024, 2048, 4096, 8192, 16384, 32768]))
        kk = int(rng.integers(1, 10)) if n < 32768 else int(rng.integers(2, 5))
        style = rng.integers(0, 3)
        if style == 0:  # all FP64-capable
            bits = [int(b) for b in rng.integers(30, 50, kk)]
        elif style == 1:  # all wide
            bits = [int(b) for b in rng.integers(51, 61, kk)]
        else:  # mixed
            bits = [int(b) for b in rng.integers(30, 61, kk)]
        # stay inside SEAL's 128-bit bound is not required for arithmetic parity; just keep n large enough for the prime count
        tb = int(rng.integers(17, 41)) if n == 32768 else int(rng.integers(14, 41))
        out.append((n, bits, tb))
    return out


@pytest.mark.parametrize("n,bits,tbits", _configs())
def test_every_operation_on_random_parameter_sets(n, bits, tbits):
    from sunscreen_amd import Context, GaloisKeys, HipBfvError, RelinearizationKeys, SecretKey
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    try:
        primes = O.coeff_modulus_create(n, bits)
    except Exception:
        pytest.skip("not enough primes of the requested sizes")
    t = O.plain_batching(n, tbits)
    if not t:
        t = (1 << tbits) - 1
    if any(t % p == 0 for p in primes):
        pytest.skip("plain modulus collides with a coefficient prime")
    o = O.Oracle(n, primes, t)
    O.seed(n + sum(bits))
    batching = O.is_prime(t) and (t - 1) % (2 * n) == 0
    elt = o.galois_elt_from_step(1) if batching and o.KK > 1 else None
    sk, pk, rk, gk = o.keygen(galois_elts=[elt, 2 * n - 1] if elt else None)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    K = o.K
    rng = np.random.default_rng(n * 31 + len(bits))
    a = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, q, (2, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    da, db = to_device(a), to_device(b)
    m = to_host(ev.multiply(da, db))
    s = to_host(ev.add(da, db))
    d = to_host(ev.sub(da, db))
    for i in range(2):
        assert (m[i] == o.multiply(a[i], b[i])).all(), ("multiply", n, bits, tbits)
        assert (s[i] == o.add(a[i], b[i])).all() and (d[i] == o.sub(a[i], b[i])).all()
    pl = rng.integers(0, t, (2, n), dtype=np.uint64)
    pl[1, 1:] = 0
    pl[1, 0] = max(1, pl[1, 0])  # a monomial: SEAL's multiply_plain shortcut
    mp = to_host(ev.multiply_plain(da, to_device(pl)))
    ap = to_host(ev.add_plain(da, to_device(pl)))
    for i in range(2):
        assert (mp[i] == o.multiply_plain(a[i], pl[i])).all(), ("multiply_plain", n, bits, tbits)
        assert (ap[i] == o.add_plain(a[i], pl[i])).all()
    if o.KK > 1:
        rkd = RelinearizationKeys.from_array(ctx, rk)
        r = to_host(ev.multiply_relin(da, db, rkd))
        r2 = to_host(ev.relinearize(to_device(m), rkd))
        for i in range(2):
            ref = o.relinearize(o.multiply(a[i], b[i]), rk)
            assert (r[i] == ref).all() and (r2[i] == ref).all(), ("relinearize", n, bits, tbits)
        if elt:
            gkd = GaloisKeys.from_arrays(ctx, gk)
            rot = to_host(ev.rotate_rows(da, 1, gkd))
            col = to_host(ev.rotate_columns(da, gkd))
            for i in range(2):
                assert (rot[i] == o.rotate_rows(a[i], 1, gk)).all(), ("rotate_rows", n, bits, tbits)
                assert (col[i] == o.rotate_columns(a[i], gk)).all()
    else:
        with pytest.raises(HipBfvError):
            RelinearizationKeys.from_array(ctx, np.zeros((1, 2, 1, n), dtype=np.uint64))
    # decryption of arbitrary residues is deterministic: Decryptor parity on the same inputs
    dec = to_host(ev.decrypt(da, SecretKey.from_array(ctx, sk)))
    for i in range(2):
        assert (dec[i] == o.decrypt(a[i], sk)).all(), ("decrypt", n, bits, tbits)
    if batching:
        vals = rng.integers(0, t, (2, n), dtype=np.uint64)
        enc = to_host(ev.encode(to_device(vals)))
        for i in range(2):
            assert (enc[i] == o.batch_encode(vals[i])).all()


def _graph_seeds():
    # HIPBFV_FUZZ_GRAPH_SEEDS="lo:hi": extended campaigns of random graphs (profiles/r06_s36_*); the default is the suite's own 16
    lo, hi = (int(x) for x in os.environ.get("HIPBFV_FUZZ_GRAPH_SEEDS", "0:16").split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("seed", _graph_seeds())
def test_random_program_graphs(seed, monkeypatch):
    """Random FheProgram DAGs (every ciphertext node kind the compiler emits, run.rs:160-341) through the batch graph
    executor vs the oracle interpreter, bit for bit: exercises operand lifetime / buffer recycling, the fused
    Multiply->Relinearize, NAF rotation chains with a power-of-two key set and shared / per-item plaintexts.
    Batches of 1 / 2 take the scheduled executor's merged launches (ready nodes of one kind in one launch sequence, staged
    operands), batches of 40 its member-by-member path with Adds folded into key-switch tails; every graph also runs through
    the node-by-node executor (HIPBFV_PROGRAM_SERIAL=1) and the two must agree word for word."""
    from oracle.program_interp import run_program
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host
    from sunscreen_amd.program import FheProgram

    n = int(os.environ.get("HIPBFV_FUZZ_GRAPH_N", "4096"))  # campaigns at the larger split geometries: 8192, 16384 (per-row packed rows)
    primes, t = O.bfv_default(n), O.plain_batching(n, 16 if n == 4096 else 20)
    o = O.Oracle(n, primes, t)
    O.seed(1000 + seed)
    nb = (n // 2).bit_length() - 1  # steps below n/2 from power-of-two keys: 11 at n = 4096
    elts = sorted({o.galois_elt_from_step(1 << i) for i in range(nb)} | {o.galois_elt_from_step(-(1 << i)) for i in range(nb)} | {2 * n - 1})
    sk, pk, rk, gk = o.keygen(galois_elts=elts)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
    rng = np.random.default_rng(seed)
    p = FheProgram()
    cts = [p.append_input_ciphertext(i) for i in range(3)]
    pls = [p.append_input_plaintext(3), p.append_input_plaintext(4)]
    depth = {c: 0 for c in cts}  # multiplicative depth: keep products decryptable is NOT required, only determinism
    for _ in range(int(rng.integers(8, 24))):
        kind = rng.choice(["add", "sub", "neg", "mul", "rotl", "rotr", "swap", "addp", "subp", "mulp", "add", "mulp", "mul"])
        a = int(rng.choice(cts))
        b = int(rng.choice(cts))
        if kind == "add":
            c = p.append_add(a, b)
        elif kind == "sub":
            c = p.append_sub(a, b)  # a == b gives a transparent ciphertext: the whole run must fail, as the reference's does
        elif kind == "neg":
            c = p.append_negate(a)
        elif kind == "mul":
            c = p.append_relinearize(p.append_multiply(a, b))
        elif kind in ("rotl", "rotr"):
            k = p.append_input_literal(int(rng.integers(1, n // 2)))
            c = p.append_rotate_left(a, k) if kind == "rotl" else p.append_rotate_right(a, k)
        elif kind == "swap":
            c = p.append_swap_rows(a)
        elif kind == "addp":
            c = p.append_add_plaintext(a, int(rng.choice(pls)))
        elif kind == "subp":
            c = p.append_sub_plaintext(a, int(rng.choice(pls)))
        else:
            c = p.append_multiply_plaintext(a, int(rng.choice(pls)))
        cts.append(c)
        depth[c] = 0
    outs = [int(x) for x in rng.choice(cts[3:], size=min(3, len(cts) - 3), replace=False)]
    for c in outs:
        p.append_output_ciphertext(c)
    q = FheProgram.from_json(p.to_json())
    batch = (1, 2, 40)[seed % 3]
    ncheck = min(batch, 3)
    K = o.K
    ins = [np.stack([rng.integers(0, pr, (batch, 2, n), dtype=np.uint64) for pr in primes[:K]], axis=2) for _ in range(3)]
    shared = rng.integers(1, t, n, dtype=np.uint64)          # one plaintext for the whole batch
    per_item = rng.integers(1, t, (batch, n), dtype=np.uint64)
    from sunscreen_amd import HipBfvError

    refs, transparent = [], False
    for i in range(ncheck):
        try:
            refs.append(run_program(o, q.nodes, q.edges, [x[i] for x in ins] + [shared, per_item[i]], rk, gk))
        except RuntimeError as e:  # the oracle mirrors SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
            assert "transparent" in str(e)
            transparent = True
    if transparent:  # x - x somewhere in the graph: runtime.run fails (sunscreen/tests/features.rs:8-34), so must the batch executor
        with pytest.raises(HipBfvError, match="transparent"):
            q.run(ev, [to_device(x) for x in ins] + [to_device(shared), to_device(per_item)], rkd, gkd)
        return
    got = q.run(ev, [to_device(x) for x in ins] + [to_device(shared), to_device(per_item)], rkd, gkd)
    got = [to_host(g) for g in got]
    for i in range(ncheck):
        ref = refs[i]
        assert len(ref) == len(got)
        for k in range(len(ref)):
            assert (got[k][i] == ref[k]).all(), (seed, i, k, q.describe())
    monkeypatch.setenv("HIPBFV_PROGRAM_SERIAL", "1")
    serial = [to_host(g) for g in q.run(ev, [to_device(x) for x in ins] + [to_device(shared), to_device(per_item)], rkd, gkd)]
    for k in range(len(got)):
        assert (serial[k] == got[k]).all(), (seed, k, q.describe())


def _rotation_trials():
    # HIPBFV_FUZZ_ROT_TRIALS="lo:hi": extended campaigns (profiles/r06_s40_*); the suite's own are 0:6
    lo, hi = (int(x) for x in os.environ.get("HIPBFV_FUZZ_ROT_TRIALS", "0:6").split(":"))
    return range(lo, hi)


@pytest.mark.parametrize("trial", _rotation_trials())
def test_rotations_at_random_batch_sizes(trial):
    """The rotation head / tail order their workgroups row by row per XCD when the row count is a multiple of 8 and in the plain
    order otherwise (kernels_split.hip: KS_XCD_ROWS): batches of random size -- each item of the batched rotate_rows / rotate_columns
    equals the same call on that item alone (one item: the plain order) word for word, and two items equal the oracle."""
    import torch
    from sunscreen_amd import Context, GaloisKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    rng = np.random.default_rng(40000 + trial)
    n = (4096, 8192, 16384)[trial % 3]
    primes, t = O.bfv_default(n), O.plain_batching(n, 20)
    o = O.Oracle(n, primes, t)
    O.seed(700 + trial)
    step_pow2 = (1 << int(rng.integers(0, (n // 2).bit_length() - 1)))
    elts = sorted({o.galois_elt_from_step(step_pow2), 2 * n - 1})
    sk, pk, rk, gk = o.keygen(galois_elts=elts)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    gkd = GaloisKeys.from_arrays(ctx, gk)
    count = int(rng.integers(1, 30 if n == 16384 else 70))
    K = o.K
    a = np.stack([rng.integers(0, pr, (count, 2, n), dtype=np.uint64) for pr in primes[:K]], axis=2)
    da = to_device(a)
    rows = ev.rotate_rows(da, step_pow2, gkd)
    cols = ev.rotate_columns(da, gkd)
    torch.cuda.synchronize()
    for i in sorted({0, count - 1, int(rng.integers(0, count)), int(rng.integers(0, count))}):
        one = da[i:i + 1].contiguous()
        assert torch.equal(ev.rotate_rows(one, step_pow2, gkd)[0], rows[i]), (trial, n, count, i)
        assert torch.equal(ev.rotate_columns(one, gkd)[0], cols[i]), (trial, n, count, i)
    hr, hc = to_host(rows), to_host(cols)
    for i in {0, count - 1}:
        assert (hr[i] == o.rotate_rows(a[i], step_pow2, gk)).all(), (trial, n, count, i)
        assert (hc[i] == o.rotate_columns(a[i], gk)).all(), (trial, n, count, i)
