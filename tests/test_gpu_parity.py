"""GPU parity tests: the HIP path (through the C ABI) must equal the CPU oracle bit for bit.

Every test calls libhipbfv.so (sunscreen_amd/lib) -- never the oracle -- for the product result and
uses the oracle only as the checker.  Sizes are kept where the oracle finishes in seconds; full
BASELINE sizes are covered by size-independent properties in test_gpu_properties.py.
"""
import os

import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import oracle_for, params

pytestmark = pytest.mark.gpu


def _setup(name, galois=None, seed=11):
    import torch  # noqa: F401
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator

    n, primes, t = params(name)
    o = oracle_for(name)
    O.seed(seed)
    sk, pk, rk, gk = o.keygen(galois_elts=galois)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk) if rk is not None else None
    gkd = GaloisKeys.from_arrays(ctx, gk) if gk else None
    return o, sk, pk, rk, gk, ctx, ev, rkd, gkd


def _rand_cts(o, pk, count, rng, lo=0, hi=100):
    vals = rng.integers(lo, hi, (count, o.n)).astype(np.uint64)
    if o.t > 1 and hasattr(o, "batch_encode"):
        try:
            cts = np.stack([o.encrypt(pk, o.batch_encode(v % o.t)) for v in vals])
        except ValueError:
            cts = np.stack([o.encrypt(pk, v % o.t) for v in vals])
    return vals, cts


@pytest.mark.parametrize("name", ["default_1024_14", "default_2048_14", "default_4096_16", "default_8192_17", "default_16384_17"])
def test_ntt_forward_inverse_bit_exact(name):
    import torch
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    n, primes, t = params(name)
    o = oracle_for(name)
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rng = np.random.default_rng(n)
    KK = len(primes)
    polys = 3 * KK + 1
    x = np.stack([rng.integers(0, primes[p % KK], n, dtype=np.uint64) for p in range(polys)])
    # edge rows: all zeros, all q-1, single one
    x[0] = 0
    x[1 % polys] = primes[1 % KK] - 1
    d = to_device(x)
    ev.ntt(d, KK, inverse=False)
    got = to_host(d)
    for p in range(polys):
        assert (got[p] == o.ntt(p % KK, x[p])).all(), (name, p)
    ev.ntt(d, KK, inverse=True)
    torch.cuda.synchronize()
    assert (to_host(d) == x).all()


def test_ntt_reproduces_seal_secret_key_fixture():
    """The HIP forward NTT of the fixture's ternary secret key equals the bits SEAL serialised
    (seal_fhe/tests/data/secret_key.bin via tests/golden/seal_key_fixture.npz)."""
    import hashlib
    import os

    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seal_key_fixture.npz"))
    primes = [int(p) for p in g["primes"]]
    ctx = Context.from_raw(8192, primes, O.plain_batching(8192, 32))
    ev = BatchEvaluator(ctx)
    s = g["sk_ternary"].astype(np.int64)
    x = np.stack([np.where(s < 0, q + s, s).astype(np.uint64) for q in primes])
    d = to_device(x)
    ev.ntt(d, 5)
    y = to_host(d)
    for j in range(5):
        assert hashlib.sha256(y[j].astype("<u8").tobytes()).hexdigest() == str(g["sk_sha256"][j])


@pytest.mark.parametrize("name", ["default_4096_16", "default_8192_17", "seal_fhe_unit", "simple_multiply", "default_16384_17"])
def test_multiply_relinearize_bit_exact(name):
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, gk, ctx, ev, rkd, gkd = _setup(name)
    rng = np.random.default_rng(5)
    count = 3
    va, a = _rand_cts(o, pk, count, rng)
    vb, b = _rand_cts(o, pk, count, rng)
    da, db = to_device(a), to_device(b)
    m = to_host(ev.multiply(da, db))
    r = to_host(ev.relinearize(to_device(m), rkd))
    fused = to_host(ev.multiply_relin(da, db, rkd))
    for i in range(count):
        om = o.multiply(a[i], b[i])
        assert (m[i] == om).all(), (name, i)
        orl = o.relinearize(om, rk)
        assert (r[i] == orl).all(), (name, i)
        assert (fused[i] == orl).all(), (name, i)
        if o.t % (2 * o.n) == 1:
            assert (o.batch_decode(o.decrypt(fused[i], sk)) == (va[i] * vb[i]) % o.t).all()
    # x * x: one operand twice selects the squaring specialisation (two forward transforms, ext polys 0 and 1 only);
    # SEAL's Evaluator::square computes the same sums (seal_fhe/src/evaluator.rs square / square_inplace)
    sq = to_host(ev.multiply(da, da))
    sqr = to_host(ev.multiply_relin(db, db, rkd))
    for i in range(count):
        assert (sq[i] == o.multiply(a[i], a[i])).all(), (name, i)
        assert (sqr[i] == o.relinearize(o.multiply(b[i], b[i]), rk)).all(), (name, i)


def test_multiply_general_sizes_and_chunking():
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, gk, ctx, ev, rkd, gkd = _setup("default_4096_16")
    rng = np.random.default_rng(6)
    va, a = _rand_cts(o, pk, 5, rng, 0, 8)
    c3 = np.stack([o.multiply(x, x) for x in a])
    ev.set_chunk_ops(2)  # 5 items -> chunks of 2,2,1
    got = to_host(ev.multiply(to_device(c3), to_device(a)))
    assert got.shape[1] == 4
    for i in range(5):
        assert (got[i] == o.multiply(c3[i], a[i])).all()
    got2 = to_host(ev.multiply_relin(to_device(a), to_device(a), rkd))
    for i in range(5):
        assert (got2[i] == o.relinearize(c3[i], rk)).all()


@pytest.mark.parametrize("name", ["default_4096_16", "seal_fhe_unit"])
def test_rotations_bit_exact(name):
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, gk, ctx, ev, rkd, gkd = _setup(name, galois="all")
    rng = np.random.default_rng(8)
    va, a = _rand_cts(o, pk, 2, rng)
    da = to_device(a)
    for steps in (1, -1, 4, 3, -5):
        got = to_host(ev.rotate_rows(da, steps, gkd))
        for i in range(2):
            assert (got[i] == o.rotate_rows(a[i], steps, gk)).all(), (name, steps)
    got = to_host(ev.rotate_columns(da, gkd))
    for i in range(2):
        assert (got[i] == o.rotate_columns(a[i], gk)).all()
    # in place
    ev.rotate_rows(da, 7, gkd, out=da)
    got = to_host(da)
    for i in range(2):
        assert (got[i] == o.rotate_rows(a[i], 7, gk)).all()


@pytest.mark.parametrize("name", ["default_4096_16", "seal_fhe_unit", "simple_multiply"])
def test_eltwise_and_plain_ops_bit_exact(name):
    from sunscreen_amd.batch import to_device, to_host

    o, sk, pk, rk, gk, ctx, ev, rkd, gkd = _setup(name)
    rng = np.random.default_rng(9)
    count = 3
    va, a = _rand_cts(o, pk, count, rng)
    vb, b = _rand_cts(o, pk, count, rng)
    da, db = to_device(a), to_device(b)
    plain = rng.integers(0, o.t, (count, o.n)).astype(np.uint64)
    plain[0, ::2] = 0
    plain[1, :] = o.t - 1
    dp = to_device(plain)
    res = {
        "add": to_host(ev.add(da, db)),
        "sub": to_host(ev.sub(da, db)),
        "neg": to_host(ev.negate(da)),
        "addp": to_host(ev.add_plain(da, dp)),
        "subp": to_host(ev.sub_plain(da, dp)),
        "mulp": to_host(ev.multiply_plain(da, dp)),
        "mulp_shared": to_host(ev.multiply_plain(da, dp[2])),
    }
    for i in range(count):
        assert (res["add"][i] == o.add(a[i], b[i])).all()
        assert (res["sub"][i] == o.sub(a[i], b[i])).all()
        assert (res["neg"][i] == o.negate(a[i])).all()
        assert (res["addp"][i] == o.add_plain(a[i], plain[i])).all()
        assert (res["subp"][i] == o.sub_plain(a[i], plain[i])).all()
        assert (res["mulp"][i] == o.multiply_plain(a[i], plain[i])).all()
        assert (res["mulp_shared"][i] == o.multiply_plain(a[i], plain[2])).all()


def test_handle_level_api_mirrors_seal_fhe():
    """The SEAL-named C entry points, driven through the seal_fhe mirror (sunscreen_amd/seal.py), on the
    reference's unit-test parameters (seal_fhe/src/bfv_evaluator.rs:255-282)."""
    from sunscreen_amd import (
        BFVEvaluator,
        BfvEncryptionParametersBuilder,
        Ciphertext,
        CoefficientModulus,
        Context,
        GaloisKeys,
        HipBfvError,
        PlainModulus,
        Plaintext,
        RelinearizationKeys,
        SecurityLevel,
    )

    params_ = (
        BfvEncryptionParametersBuilder()
        .set_poly_modulus_degree(8192)
        .set_coefficient_modulus(CoefficientModulus.create(8192, [50, 30, 30, 50, 50]))
        .set_plain_modulus(PlainModulus.batching(8192, 32))
        .build()
    )
    ctx = Context(params_, False, SecurityLevel.TC128)
    assert ctx.key_primes == [1125899905744897, 1073643521, 1073692673, 1125899906629633, 1125899906826241]
    o = oracle_for("seal_fhe_unit")
    assert ctx.plain_modulus == o.t
    O.seed(21)
    sk, pk, rk, gk = o.keygen(galois_elts="all")
    ev = BFVEvaluator(ctx)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    gkd = GaloisKeys.from_arrays(ctx, gk)
    n = o.n
    va = np.array([n // 2 - i for i in range(n)], dtype=np.int64)
    vb = np.array([16 - i % 32 for i in range(n)], dtype=np.int64)
    a_np = o.encrypt(pk, o.batch_encode((va % o.t).astype(np.uint64)))
    b_np = o.encrypt(pk, o.batch_encode((vb % o.t).astype(np.uint64)))
    a, b = Ciphertext.from_array(ctx, a_np), Ciphertext.from_array(ctx, b_np)
    assert a.num_polynomials() == 2 and a.coeff_modulus_size() == 4 and not a.is_ntt_form()
    assert a.get_data(5) == int(a_np.reshape(-1)[5])
    assert a.get_coefficient(1, 7) == [int(a_np[1, i, 7]) for i in range(4)]

    def dec(ct):
        v = o.batch_decode(o.decrypt(ct.to_array(), sk)).astype(np.int64)
        return np.where(v > o.t // 2, v - o.t, v)

    assert (ev.negate(a).to_array() == o.negate(a_np)).all()
    assert (ev.add(a, b).to_array() == o.add(a_np, b_np)).all()
    assert (ev.sub(a, b).to_array() == o.sub(a_np, b_np)).all()
    m = ev.multiply(a, b)
    assert m.num_polynomials() == 3
    assert (m.to_array() == o.multiply(a_np, b_np)).all()
    r = ev.relinearize(m, rkd)
    assert r.num_polynomials() == 2
    assert (r.to_array() == o.relinearize(o.multiply(a_np, b_np), rk)).all()
    assert (dec(r) == va * vb).all()
    # in-place variants alias the destination with the operand
    c = a.clone()
    ev.multiply_inplace(c, b)
    ev.relinearize_inplace(c, rkd)
    assert (c.to_array() == r.to_array()).all()
    sq = ev.square(b)
    assert (sq.to_array() == o.multiply(b_np, b_np)).all()
    # add of size 3 and size 2
    mixed = ev.add(m, a)
    assert (mixed.to_array() == o.add(o.multiply(a_np, b_np), a_np)).all()
    mixed = ev.sub(a, m)
    assert (mixed.to_array() == o.sub(a_np, o.multiply(a_np, b_np))).all()
    # add_many / multiply_many / exponentiate
    s3 = ev.add_many([a, b, a])
    assert (s3.to_array() == o.add(o.add(a_np, b_np), a_np)).all()
    small = Ciphertext.from_array(ctx, b_np)
    e3 = ev.exponentiate(small, 3, rkd)
    assert (dec(e3) == vb**3).all()
    assert (e3.to_array() == o.exponentiate(b_np, 3, rk)).all()  # bits: SEAL's work-list order (oracle.multiply_many)
    mm = ev.multiply_many([b, b, b, b], rkd)
    assert (dec(mm) == vb**4).all()
    assert (mm.to_array() == o.multiply_many([b_np] * 4, rk)).all()
    # five distinct operands: the order matters -- ((c0 c1)(c2 c3)) c4 with c4 entering at the LAST product, not a left fold
    five = [a_np, b_np, o.add(a_np, b_np), o.negate(b_np), o.sub(a_np, b_np)]
    m5 = ev.multiply_many([Ciphertext.from_array(ctx, x) for x in five], rkd)
    assert (m5.to_array() == o.multiply_many(five, rk)).all()
    fold = five[0]
    for x in five[1:]:
        fold = o.relinearize(o.multiply(fold, x), rk)
    assert not (m5.to_array() == fold).all()
    e1 = ev.exponentiate(small, 1, rkd)
    assert (e1.to_array() == b_np).all()
    # plaintext operands
    pb = o.batch_encode((vb % o.t).astype(np.uint64))
    p = Plaintext.from_coefficients([int(x) for x in pb])
    assert p.len() == n and p.get_coefficient(3) == int(pb[3])
    assert (ev.add_plain(a, p).to_array() == o.add_plain(a_np, pb)).all()
    assert (ev.sub_plain(a, p).to_array() == o.sub_plain(a_np, pb)).all()
    assert (ev.multiply_plain(a, p).to_array() == o.multiply_plain(a_np, pb)).all()
    mono = Plaintext.from_coefficients([0, 0, 0, o.t - 2])
    assert (ev.multiply_plain(a, mono).to_array() == o.multiply_plain(a_np, np.array([0, 0, 0, o.t - 2], dtype=np.uint64))).all()
    zero = Plaintext.from_coefficients([0])
    with pytest.raises(HipBfvError) as ei:
        ev.multiply_plain(a, zero)  # transparent result (sunscreen/tests/features.rs:8-34)
    assert ei.value.kind == "InternalError"
    # rotations (bfv_evaluator.rs:880-970)
    rot = ev.rotate_rows(a, -1, gkd)
    assert (rot.to_array() == o.rotate_rows(a_np, -1, gk)).all()
    d = dec(rot)
    assert va[0] == d[1] and va[1] == d[2] and va[4096] == d[4097]
    rot3 = ev.rotate_rows(a, 3, gkd)  # NAF chain
    assert (rot3.to_array() == o.rotate_rows(a_np, 3, gk)).all()
    cols = ev.rotate_columns(a, gkd)
    assert (cols.to_array() == o.rotate_columns(a_np, gk)).all()
    c = a.clone()
    ev.rotate_columns_inplace(c, gkd)
    assert (c.to_array() == cols.to_array()).all()
    with pytest.raises(HipBfvError) as ei:
        ev.rotate_rows(a, 1, GaloisKeys())
    assert ei.value.kind == "InvalidArgument"
    with pytest.raises(HipBfvError):
        ev.relinearize(ev.multiply(m, a), rkd)  # size 4: "not enough relinearization keys"


def test_wire_format_roundtrip_on_handles():
    """Ciphertext / RelinKeys / GaloisKeys / Plaintext through the SEAL 4.0 wire format (zstd and uncompressed):
    as_bytes -> from_bytes reproduces the object and the evaluator accepts it (plaintext_ciphertext.rs:451-497,
    key_generator.rs:493-573)."""
    from sunscreen_amd import BFVEvaluator, Ciphertext, Context, GaloisKeys, HipBfvError, Plaintext, RelinearizationKeys

    n, primes, t = params("default_4096_16")
    o = oracle_for("default_4096_16")
    O.seed(41)
    sk, pk, rk, gk = o.keygen(galois_elts=[3, 2 * n - 1])
    ctx = Context.from_raw(n, primes, t)
    ev = BFVEvaluator(ctx)
    rng = np.random.default_rng(1)
    va = rng.integers(0, 50, n).astype(np.uint64)
    a_np = o.encrypt(pk, o.batch_encode(va))
    a = Ciphertext.from_array(ctx, a_np)
    for compr in (0, 2):
        blob = a.as_bytes(compr)
        assert blob[:5] == bytes([0x5E, 0xA1, 16, 4, 0]) and blob[5] == compr
        assert int.from_bytes(blob[8:16], "little") == len(blob)
        b = Ciphertext.from_bytes(ctx, blob)
        assert (b.to_array() == a_np).all() and b.num_polynomials() == 2
    rkd = RelinearizationKeys.from_bytes(ctx, RelinearizationKeys.from_array(ctx, rk).as_bytes())
    gkd = GaloisKeys.from_bytes(ctx, GaloisKeys.from_arrays(ctx, gk).as_bytes(0))
    m = ev.relinearize(ev.multiply(a, a), rkd)
    assert (m.to_array() == o.relinearize(o.multiply(a_np, a_np), rk)).all()
    r = ev.rotate_rows(m, 1, gkd)
    assert (r.to_array() == o.rotate_rows(o.relinearize(o.multiply(a_np, a_np), rk), 1, gk)).all()
    c = ev.rotate_columns(a, gkd)
    assert (c.to_array() == o.rotate_columns(a_np, gk)).all()
    p = Plaintext.from_coefficients([1, 2, 3, 0, 5])
    q = Plaintext.from_bytes(ctx, p.as_bytes())
    assert q.len() == 5 and [q.get_coefficient(i) for i in range(5)] == [1, 2, 3, 0, 5]
    # a ciphertext of other parameters is rejected (parms_id mismatch)
    other = Context.from_raw(n, primes, O.plain_batching(n, 17))
    with pytest.raises(HipBfvError) as ei:
        Ciphertext.from_bytes(other, a.as_bytes())
    assert ei.value.kind == "InvalidArgument"
    with pytest.raises(HipBfvError):
        Ciphertext.from_bytes(ctx, a.as_bytes()[:-5])


@pytest.mark.parametrize(
    "n,bits,tbits",
    [
        (2048, [54], 14),            # single prime, no special prime: multiply works, key switching does not
        (1024, [27], 0),             # smallest supported degree, non-batching plain modulus
        (4096, [58, 59, 60], 16),    # wide user primes: integer path everywhere (no FP64), generic Barrett
        (8192, [36, 36, 37, 38], 20),  # K = 3
        (16384, [40, 40, 40, 40, 40, 41], 17),  # K = 5 > 4: whole-polynomial multiply + split key switch
    ],
)
def test_multiply_edge_parameter_sets(n, bits, tbits):
    from sunscreen_amd import Context, HipBfvError, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, tbits) if tbits else 64
    o = O.Oracle(n, primes, t)
    O.seed(n + len(bits))
    sk, pk, rk, _ = o.keygen()
    ctx = Context.from_raw(n, primes, t)
    ev = BatchEvaluator(ctx)
    rng = np.random.default_rng(n)
    msgs = rng.integers(0, min(t, 16), (2, 2, n)).astype(np.uint64)
    a = np.stack([o.encrypt(pk, msgs[0, i]) for i in range(2)])
    b = np.stack([o.encrypt(pk, msgs[1, i]) for i in range(2)])
    m = to_host(ev.multiply(to_device(a), to_device(b)))
    for i in range(2):
        assert (m[i] == o.multiply(a[i], b[i])).all()
    if rk is not None:
        rkd = RelinearizationKeys.from_array(ctx, rk)
        r = to_host(ev.multiply_relin(to_device(a), to_device(b), rkd))
        for i in range(2):
            assert (r[i] == o.relinearize(o.multiply(a[i], b[i]), rk)).all()
    else:
        with pytest.raises(HipBfvError):
            RelinearizationKeys.from_array(ctx, np.zeros((1, 2, 1, n), dtype=np.uint64))


# These two tests assert the DEFAULT base choice; the switches that turn it off make them moot (the rest of the suite is
# meaningful -- and green -- under every switch, which is how the variants are exercised wholesale)
_own_base_off = os.environ.get("HIPBFV_SEAL_AUX") == "1" or os.environ.get("HIPBFV_NO_F64") == "1"


@pytest.mark.skipif(_own_base_off, reason="the own auxiliary base is switched off by the environment")
def test_auxiliary_base_choice_and_bound():
    """The BEHZ auxiliary base is internal to multiply.  FP64-capable data primes -> the library's own base of
    primes below 2^48 whose product covers 2^(bits(t) + log2 n + bits(q) + 3) (context.cpp derives it; SEAL reserves 32 bits
    where this has log2 n + 3; tests/test_behz_base_bound_cpu.py replays
    the steps the bound protects in exact integers);
    wide data primes where the split multiply runs (K <= 4, 4096 <= n <= 16384) -> the MIXED base: the same kind of
    auxiliary primes (FP64 rows in the middle kernel) beside integer data rows; wide data primes elsewhere -> SEAL's own
    61-bit base, identical to the oracle's."""
    from sunscreen_amd import Context

    def covers(ctx, primes, t, n):
        q = 1
        for p in primes[:-1]:
            q *= p
        prod = 1
        for p in ctx.aux_primes:
            assert p < 2**48 and p % (2 * n) == 1 and p not in primes and O.is_prime(p)
            prod *= p
        assert len(set(ctx.aux_primes)) == len(ctx.aux_primes)
        reserve = n.bit_length() - 1 + 3
        assert prod.bit_length() > reserve + t.bit_length() + q.bit_length()

    n, primes, t = params("default_8192_17")
    ctx = Context.from_raw(n, primes, t)
    assert ctx.aux_fp64 and not ctx.aux_mixed and len(ctx.aux_primes) >= 2
    covers(ctx, primes, t, n)
    wide = O.coeff_modulus_create(4096, [58, 59, 60])
    t2 = O.plain_batching(4096, 16)
    ctx2 = Context.from_raw(4096, wide, t2)
    assert ctx2.aux_mixed and not ctx2.aux_fp64
    covers(ctx2, wide, t2, 4096)
    wide3 = O.coeff_modulus_create(2048, [54, 55])  # n = 2048: no split pipelines, SEAL's base
    t3 = O.plain_batching(2048, 16)
    ctx3 = Context.from_raw(2048, wide3, t3)
    o3 = O.Oracle(2048, wide3, t3)
    assert not ctx3.aux_fp64 and not ctx3.aux_mixed and ctx3.aux_primes == [int(p) for p in o3.bsk]


@pytest.mark.skipif(_own_base_off, reason="the own auxiliary base is switched off by the environment")
@pytest.mark.parametrize("n,tbits", [(8192, 40), (8192, 50), (4096, 30), (16384, 45)])
def test_multiply_large_plain_modulus_own_base(n, tbits):
    """Large plain moduli push the integers that pass through the auxiliary base towards its size bound
    (floor(t*c/q) grows with t): the own-base product must still equal the oracle's (SEAL-base) product, on
    random operands, on operands with every residue at q_i - 1, and on the operands that drive the tensor to its largest
    magnitude: every coefficient of every polynomial = floor(q/2) (the Montgomery step leaves |x'| ~ q/2 whatever the sign,
    the negacyclic products then reach N * q^2/4 at coefficient N - 1 and -(N - 1) * q^2/4 at coefficient 0, and c1 is two of
    them), and the same with alternating signs."""
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.bfv_default(n)
    t = O.plain_batching(n, tbits)
    o = O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    assert ctx.aux_fp64
    ev = BatchEvaluator(ctx)
    K = len(primes) - 1
    rng = np.random.default_rng(tbits)
    cnt = 20  # beyond the few-operations threshold (evaluator.cpp few_for_split_mul): the split pipelines run these
    a = np.stack([rng.integers(0, q, (cnt, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, q, (cnt, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    Q = 1
    for q in primes[:K]:
        Q *= q
    half = Q // 2
    sign = np.where(np.arange(n) % 2 == 0, 1, -1)
    for i, q in enumerate(primes[:K]):
        a[2, :, i, :] = q - 1
        b[2, :, i, :] = q - 1
        a[3, :, i, :] = half % q
        b[3, :, i, :] = half % q
        # +half, -half, +half, ... against all +half: the other sign pattern of the extreme sums
        a[4, :, i, :] = np.where(sign > 0, half % q, (Q - half) % q).astype(np.uint64)
        b[4, :, i, :] = half % q
    m = to_host(ev.multiply(to_device(a), to_device(b)))
    for i in (0, 1, 2, 3, 4, cnt - 1):
        assert (m[i] == o.multiply(a[i], b[i])).all(), i
    # and the same five through the whole-polynomial kernels (a call of five operands stays below the threshold at n = 8192)
    if n == 8192:
        m5 = to_host(ev.multiply(to_device(a[:5]), to_device(b[:5])))
        assert (m5 == m[:5]).all()


@pytest.mark.skipif(_own_base_off, reason="the own auxiliary base is switched off by the environment")
@pytest.mark.parametrize("n,bits", [(8192, [46, 47, 47, 48]), (8192, [45, 45, 46, 46, 47]), (16384, [46, 47, 47, 47, 47, 48])])
def test_plan_flags_on_primes_that_keep_their_store_side_reductions(n, bits):
    """r05's range plan skips the reduction in front of a packed store and after the tail's scaling where a prime is small enough
    (every prime of the default sets at n = 8192).  These sets sit on the OTHER side of those decisions -- 45 ... 48-bit data and
    key primes, packed rows (all below 2^48), kPlanStoreReduce set for the head's and / or the middle kernel's outputs, the one
    reduction of the inverse not in the last pass -- and must give the oracle's bits too: multiply and the fused multiply +
    relinearize (K <= 4: five launches; K = 5: the 8-prime instantiations), on random operands and on the ones that drive
    every intermediate to its extreme (all q_i - 1, all floor(q/2), alternating signs)."""
    import ctypes as C

    from sunscreen_amd import Context, RelinearizationKeys, _lib
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False
    O.seed(4647)
    sk, pk, rk, _ = o.keygen()
    logn = n.bit_length() - 1
    flags = 0
    for q in primes:  # the plans really are on the far side of at least one skip for some prime of the set
        out = (C.c_uint32 * 6)()
        assert _lib.load().hipbfv_debug_f64_plan(C.c_uint64(int(q)), C.c_uint32(logn), out) == 0 and out[0] == 1 and out[3] == 1
        flags |= (out[4] | out[5]) & (1 << 30)
    assert flags, "no prime of this set keeps a store-side reduction: the test would not test what it says"
    ctx = Context.from_raw(n, primes, t)
    # (under the HIPBFV_NO_PACK switch suite the rows are 8-byte doubles and the store-side flags are moot: the bits are still checked)
    assert ctx.aux_fp64 and (ctx.packed_mul or os.environ.get("HIPBFV_NO_PACK"))
    ev = BatchEvaluator(ctx)
    ev.set_transparent_check(False)
    rkd = RelinearizationKeys.from_array(ctx, rk)
    K = len(primes) - 1
    rng = np.random.default_rng(sum(bits))
    cnt = 20  # beyond the few-operations threshold: the split pipelines run these
    a = np.stack([rng.integers(0, q, (cnt, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, q, (cnt, 2, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    Q = 1
    for q in primes[:K]:
        Q *= q
    half = Q // 2
    sign = np.where(np.arange(n) % 2 == 0, 1, -1)
    for i, q in enumerate(primes[:K]):
        a[2, :, i, :] = q - 1
        b[2, :, i, :] = q - 1
        a[3, :, i, :] = half % q
        b[3, :, i, :] = half % q
        a[4, :, i, :] = np.where(sign > 0, half % q, (Q - half) % q).astype(np.uint64)
        b[4, :, i, :] = half % q
    da, db = to_device(a), to_device(b)
    m = to_host(ev.multiply(da, db))
    r = to_host(ev.multiply_relin(da, db, rkd))
    for i in (0, 1, 2, 3, 4, cnt - 1):
        om = o.multiply(a[i], b[i])
        assert (m[i] == om).all(), ("multiply", i)
        assert (r[i] == o.relinearize(om, rk)).all(), ("multiply_relin", i)


@pytest.mark.skipif(_own_base_off, reason="the own auxiliary base is switched off by the environment")
@pytest.mark.parametrize("n,tbits,size", [(4096, 30, 8), (8192, 50, 4), (8192, 17, 3)])
def test_multiply_extreme_operands_of_larger_sizes_own_base(n, tbits, size):
    """Evaluator::multiply accepts size_a + size_b <= 16: a coefficient of the tensor is then a sum of up to 8 negacyclic
    products, the factor the base bound reserves its 3 bits for (context.cpp).  All-floor(q/2) operands of size `size` x `size`
    reach that sum; the product equals the oracle's (SEAL's base, sized with 32 bits of reserve)."""
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.bfv_default(n)
    t = O.plain_batching(n, tbits)
    o = O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    assert ctx.aux_fp64
    ev = BatchEvaluator(ctx)
    K = len(primes) - 1
    Q = 1
    for q in primes[:K]:
        Q *= q
    half = Q // 2
    rng = np.random.default_rng(size)
    a = np.stack([rng.integers(0, q, (2, size, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, q, (2, size, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    for i, q in enumerate(primes[:K]):
        a[1, :, i, :] = half % q
        b[1, :, i, :] = half % q
    m = to_host(ev.multiply(to_device(a), to_device(b)))
    assert m.shape[1] == 2 * size - 1
    for i in range(2):
        assert (m[i] == o.multiply(a[i], b[i])).all(), i


@pytest.mark.skipif(_own_base_off, reason="the own auxiliary bases are switched off by the environment")
@pytest.mark.parametrize("n,bits,tbits,size", [(8192, [54, 54, 54, 56], 17, 2), (16384, [52, 54], 26, 2), (4096, [57, 51], 20, 2),
                                               (16384, [58, 52, 59, 52, 53], 25, 2), (8192, [54, 54, 54, 56], 40, 3)])
def test_multiply_extreme_operands_mixed_base(n, bits, tbits, size):
    """The same extreme operands through the MIXED base (integer data rows, FP64 auxiliary rows).  Under the derived base bound
    the Shenoy-Kumaresan correction alpha_sk = e - floor(F/B) is as large as m_sk/4 -- the 32-bit cast the mixed tail used to
    apply to it (adequate under SEAL's sizing, |F/B| < 2^25) gave wrong products on three parameter sets of the fuzz suite the
    day the bound changed; this test holds the largest |F| there is: all-floor(q/2) operands, both sign patterns, and random
    ones, against the oracle (SEAL's base)."""
    from sunscreen_amd import Context
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    primes = O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, tbits)
    o = O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    assert ctx.aux_mixed
    ev = BatchEvaluator(ctx)
    K = len(primes) - 1
    Q = 1
    for q in primes[:K]:
        Q *= q
    half = Q // 2
    rng = np.random.default_rng(n + size)
    cnt = 20  # beyond the few-operations threshold: the split pipelines (mul_head / mul_mid / mul_tail) when size == 2
    a = np.stack([rng.integers(0, q, (cnt, size, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    b = np.stack([rng.integers(0, q, (cnt, size, n), dtype=np.uint64) for q in primes[:K]], axis=2)
    sign = np.where(np.arange(n) % 2 == 0, 1, -1)
    for i, q in enumerate(primes[:K]):
        a[1, :, i, :] = half % q
        b[1, :, i, :] = half % q
        a[2, :, i, :] = np.where(sign > 0, half % q, (Q - half) % q).astype(np.uint64)
        b[2, :, i, :] = half % q
    m = to_host(ev.multiply(to_device(a), to_device(b)))
    for i in (0, 1, 2, cnt - 1):
        assert (m[i] == o.multiply(a[i], b[i])).all(), i


def test_largest_degree_n32768_two_kernel_ntt():
    """n = 32768 (SEAL default: 16 primes, K = 15): residue polynomials (256 KB) exceed one CU's LDS, so the
    transforms run as two kernels each (head + block-local / block-local + tail).  sunscreen/src/params.rs:37
    lists 32768 as the largest lattice dimension the compiler tries."""
    from sunscreen_amd import Context, GaloisKeys, RelinearizationKeys
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    n = 32768
    primes = O.bfv_default(n)
    t = O.plain_batching(n, 17)
    o = O.Oracle(n, primes, t)
    O.seed(15)
    sk, pk, rk, gk = o.keygen(galois_elts=[3])
    ctx = Context.from_raw(n, primes, t)
    assert ctx.K == 15 and ctx.KK == 16
    ev = BatchEvaluator(ctx)
    rng = np.random.default_rng(15)
    # NTT round trip and parity for every key-level prime
    x = np.stack([rng.integers(0, primes[p % 16], n, dtype=np.uint64) for p in range(17)])
    d = to_device(x)
    ev.ntt(d, 16)
    got = to_host(d)
    for p in range(17):
        assert (got[p] == o.ntt(p % 16, x[p])).all(), p
    ev.ntt(d, 16, inverse=True)
    assert (to_host(d) == x).all()
    va = rng.integers(0, 100, n).astype(np.uint64)
    vb = rng.integers(0, 100, n).astype(np.uint64)
    a = np.stack([o.encrypt(pk, o.batch_encode(va))])
    b = np.stack([o.encrypt(pk, o.batch_encode(vb))])
    r = to_host(ev.multiply_relin(to_device(a), to_device(b), RelinearizationKeys.from_array(ctx, rk)))
    ref = o.relinearize(o.multiply(a[0], b[0]), rk)
    assert (r[0] == ref).all()
    assert (o.batch_decode(o.decrypt(r[0], sk)) == (va * vb) % t).all()
    g = to_host(ev.rotate_rows(to_device(a), 1, GaloisKeys.from_arrays(ctx, gk)))
    assert (g[0] == o.rotate_rows(a[0], 1, gk)).all()
    # r06: the key switch runs through the head / middle / tail kernels at this degree too (integer-policy key primes, K = 15 digits
    # of 16 rows each); a few distinct items in one call, the stand-alone relinearisation, and the same bits with the whole-polynomial
    # key switch (HIPBFV_NO_SPLIT_KS=1 in a fresh evaluator)
    a3 = np.stack([a[0], b[0], o.encrypt(pk, o.batch_encode((va + vb) % t))])
    b3 = np.stack([b[0], b[0], a[0]])
    rkd = RelinearizationKeys.from_array(ctx, rk)
    prod = ev.multiply(to_device(a3), to_device(b3))
    r3 = to_host(ev.relinearize(prod, rkd))
    for i in range(3):
        assert (r3[i] == o.relinearize(o.multiply(a3[i], b3[i]), rk)).all(), i
    g3 = to_host(ev.rotate_rows(to_device(a3), 1, GaloisKeys.from_arrays(ctx, gk)))
    os.environ["HIPBFV_NO_SPLIT_KS"] = "1"
    try:
        ev_whole = BatchEvaluator(ctx)
        assert torch_equal(ev_whole.relinearize(prod, rkd), r3)
        assert torch_equal(ev_whole.rotate_rows(to_device(a3), 1, GaloisKeys.from_arrays(ctx, gk)), g3)
    finally:
        os.environ.pop("HIPBFV_NO_SPLIT_KS", None)


def torch_equal(t, host):
    from sunscreen_amd.batch import to_host

    return bool((to_host(t) == host).all())


def test_modulus_switching_chain():
    """Evaluator_ModSwitchToNext (seal_fhe/src/evaluator.rs:84-157): ciphertexts move down the chain, and every
    operation, the Decryptor and the wire format accept them at their level.  Bit-exact against the oracle level by
    level (the oracle's lower level is simply an Oracle over the shortened prime list)."""
    from sunscreen_amd import (BFVEvaluator, Ciphertext, Context, Decryptor, GaloisKeys, HipBfvError, Plaintext, RelinearizationKeys,
                               SecretKey)
    from sunscreen_amd.batch import BatchEvaluator, to_device, to_host

    n = 8192
    primes, t = O.bfv_default(n), O.plain_batching(n, 17)
    o0 = O.Oracle(n, primes, t)
    O.seed(61)
    elt = o0.galois_elt_from_step(3)
    sk, pk, rk, gk = o0.keygen(galois_elts=[elt, 2 * n - 1])
    ctx = Context.from_raw(n, primes, t)
    be = BFVEvaluator(ctx)
    rkd, gkd = RelinearizationKeys.from_array(ctx, rk), GaloisKeys.from_arrays(ctx, gk)
    dec = Decryptor(ctx, SecretKey.from_array(ctx, sk))
    rng = np.random.default_rng(2)
    va, vb = rng.integers(0, 40, n).astype(np.uint64), rng.integers(0, 40, n).astype(np.uint64)
    a0, b0 = o0.encrypt(pk, o0.batch_encode(va)), o0.encrypt(pk, o0.batch_encode(vb))
    ha, hb = Ciphertext.from_array(ctx, a0), Ciphertext.from_array(ctx, b0)

    def lower(o, keys, galois):  # oracle-side view of the next level: drop the last data prime everywhere
        o2 = o.next_level()
        rows = list(range(o2.K)) + [o.KK - 1]
        k2 = np.ascontiguousarray(keys[: o2.K][:, :, rows, :])
        g2 = {e: np.ascontiguousarray(g[: o2.K][:, :, rows, :]) for e, g in galois.items()}
        return o2, k2, g2

    o, cur_rk, cur_gk, cur_sk = o0, rk, gk, sk
    ra, rb = a0, b0
    for level in range(1, 4):  # K = 4 -> 3 -> 2 -> 1
        ha, hb = be.mod_switch_to_next(ha), be.mod_switch_to_next(hb)
        ra, rb = o.mod_switch_to_next(ra), o.mod_switch_to_next(rb)
        o2, cur_rk, cur_gk = lower(o, cur_rk, cur_gk)
        cur_sk = np.concatenate([cur_sk[: o2.K], cur_sk[o.K :]])
        o = o2
        assert ha.coeff_modulus_size() == o.K and (ha.to_array() == ra).all() and (hb.to_array() == rb).all(), level
        # every operation at this level, against the oracle of this level
        prod = be.multiply(ha, hb)
        assert (prod.to_array() == o.multiply(ra, rb)).all(), level
        rel = be.relinearize(prod, rkd)
        assert (rel.to_array() == o.relinearize(o.multiply(ra, rb), cur_rk)).all(), level
        assert (be.add(ha, hb).to_array() == o.add(ra, rb)).all()
        assert (be.rotate_rows(ha, 3, gkd).to_array() == o.rotate_rows(ra, 3, cur_gk)).all(), level
        assert (be.rotate_columns(ha, gkd).to_array() == o.rotate_columns(ra, cur_gk)).all(), level
        pl = o.batch_encode(vb)
        assert (be.multiply_plain(ha, Plaintext.from_coefficients([int(x) for x in pl])).to_array() == o.multiply_plain(ra, pl)).all()
        # decryption and the noise budget at this level
        got = dec.decrypt(rel)
        coeffs = np.array([got.get_coefficient(k) for k in range(got.len())] + [0] * (n - got.len()), dtype=np.uint64)
        ref_rel = o.relinearize(o.multiply(ra, rb), cur_rk)
        assert (coeffs == o.decrypt(ref_rel, cur_sk)).all(), level
        if o.noise_budget(ref_rel, cur_sk) > 0:  # the single 43-bit prime of the last level cannot hold a product
            assert (o.batch_decode(coeffs) == (va * vb) % t).all(), level
        assert dec.invariant_noise_budget(ha) == o.noise_budget(ra, cur_sk), level
        # wire format: the parms_id names the level
        back = Ciphertext.from_bytes(ctx, ha.as_bytes())
        assert back.coeff_modulus_size() == o.K and (back.to_array() == ra).all()
        with pytest.raises(HipBfvError):  # operands of different levels do not mix
            be.add(ha, Ciphertext.from_array(ctx, a0))
    with pytest.raises(HipBfvError) as ei:  # end of the chain
        be.mod_switch_to_next(ha)
    assert ei.value.kind == "InvalidArgument"
    with pytest.raises(HipBfvError):  # BFV plaintexts are not in NTT form
        be.mod_switch_to_next_plaintext(Plaintext.from_coefficients([1]))
    # batched form: level contexts for the device-pointer API
    ctx1 = ctx.next_level()
    assert ctx1.K == 3 and ctx1.key_primes == primes[:3] + primes[-1:]
    ev0, ev1 = BatchEvaluator(ctx), BatchEvaluator(ctx1)
    batch = np.stack([a0, b0])
    sw = ev0.mod_switch(to_device(batch))
    assert (to_host(sw)[0] == o0.mod_switch_to_next(a0)).all()
    o1, rk1, _ = lower(o0, rk, gk)
    r1 = to_host(ev1.multiply_relin(sw[:1].contiguous(), sw[1:].contiguous(), RelinearizationKeys.from_array(ctx1, rk1)))
    assert (r1[0] == o1.relinearize(o1.multiply(o0.mod_switch_to_next(a0), o0.mod_switch_to_next(b0)), rk1)).all()
    # a context created without the chain refuses
    from sunscreen_amd import BfvEncryptionParametersBuilder, CoefficientModulus, PlainModulus

    params = (BfvEncryptionParametersBuilder().set_poly_modulus_degree(n).set_coefficient_modulus(CoefficientModulus.bfv_default(n))
              .set_plain_modulus(PlainModulus.batching(n, 17)).build())
    flat = Context(params, expand_mod_chain=False)
    with pytest.raises(HipBfvError):
        BFVEvaluator(flat).mod_switch_to_next(Ciphertext.from_array(flat, a0))
