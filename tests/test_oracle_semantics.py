"""Decrypt-and-compare matrix for the CPU oracle, mirroring the reference's evaluator tests:
seal_fhe/src/bfv_evaluator.rs:322-970, seal_fhe/tests/assumptions.rs, sunscreen_runtime/src/run.rs:595-881.
These tests establish that the oracle is a correct BFV evaluator with SEAL's observable behaviour
(slot semantics, ciphertext sizes, noise-budget invariants) before it is used to judge the HIP path.
"""
import numpy as np
import pytest

from oracle import bfv_oracle as O
from tests.bfv_helpers import decode_signed, encode_signed, make_small_vec, make_vec, oracle_for


@pytest.fixture(scope="module")
def unit():
    o = oracle_for("seal_fhe_unit")
    O.seed(0xBF5)
    sk, pk, rk, gk = o.keygen(galois_elts="all")
    return o, sk, pk, rk, gk


def enc(o, pk, vals):
    return o.encrypt(pk, encode_signed(o, vals))


def dec(o, sk, ct):
    return decode_signed(o, o.decrypt(ct, sk))


def test_negate(unit):
    o, sk, pk, rk, gk = unit
    a = make_vec(o.n)
    assert (dec(o, sk, o.negate(enc(o, pk, a))) == -a).all()


def test_add_sub(unit):
    o, sk, pk, rk, gk = unit
    a, b = make_vec(o.n), make_vec(o.n)[::-1].copy()
    ca, cb = enc(o, pk, a), enc(o, pk, b)
    assert (dec(o, sk, o.add(ca, cb)) == a + b).all()
    assert (dec(o, sk, o.sub(ca, cb)) == a - b).all()


def test_multiply_square_relinearize(unit):
    o, sk, pk, rk, gk = unit
    a, b = make_vec(o.n), make_vec(o.n)[::-1].copy()
    ca, cb = enc(o, pk, a), enc(o, pk, b)
    cm = o.multiply(ca, cb)
    assert cm.shape[0] == 3  # assumptions.rs:36-82: size 2 -> 3
    assert (dec(o, sk, cm) == a * b).all()
    cr = o.relinearize(cm, rk)
    assert cr.shape[0] == 2
    assert (dec(o, sk, cr) == a * b).all()
    sq = o.multiply(ca, ca)
    assert (dec(o, sk, sq) == a * a).all()


def test_multiply_size3_by_size2(unit):
    o, sk, pk, rk, gk = unit
    a = make_small_vec(o.n)
    ca = enc(o, pk, a)
    c3 = o.multiply(ca, ca)
    c4 = o.multiply(c3, ca)
    assert c4.shape[0] == 4
    assert (dec(o, sk, c4) == a * a * a).all()


def test_relin_reduces_noise_growth_over_two_squarings(unit):
    # seal_fhe/src/bfv_evaluator.rs:618-680
    o, sk, pk, rk, gk = unit
    a = make_small_vec(o.n)
    ca = enc(o, pk, a)
    with_relin = o.relinearize(o.multiply(ca, ca), rk)
    with_relin = o.relinearize(o.multiply(with_relin, with_relin), rk)
    no_relin = o.multiply(ca, ca)
    no_relin = o.multiply(no_relin, no_relin)
    assert (dec(o, sk, with_relin) == a**4).all()
    assert o.noise_budget(with_relin, sk) >= o.noise_budget(no_relin, sk)


def test_plain_ops(unit):
    o, sk, pk, rk, gk = unit
    a, b = make_vec(o.n), make_small_vec(o.n)
    ca, pb = enc(o, pk, a), encode_signed(o, b)
    assert (dec(o, sk, o.add_plain(ca, pb)) == a + b).all()
    assert (dec(o, sk, o.sub_plain(ca, pb)) == a - b).all()
    mp = o.multiply_plain(ca, pb)
    assert mp.shape[0] == 2  # assumptions.rs:36-82: multiply_plain keeps the size
    assert (dec(o, sk, mp) == a * b).all()


def test_multiply_plain_monomial_and_zero(unit):
    o, sk, pk, rk, gk = unit
    a = make_small_vec(o.n)
    ca = enc(o, pk, a)
    # plaintext "5 * x^3": polynomial product, checked in the coefficient domain
    plain = np.zeros(4, dtype=np.uint64)
    plain[3] = 5
    got = o.decrypt(o.multiply_plain(ca, plain), sk)
    pa = encode_signed(o, a).astype(object)
    exp = np.zeros(o.n, dtype=object)
    for k in range(o.n):
        idx = k + 3
        v = int(pa[k]) * 5
        if idx >= o.n:
            exp[idx - o.n] = (-v) % o.t
        else:
            exp[idx] = v % o.t
    assert [int(x) for x in got] == [int(x) for x in exp]
    # zero plaintext -> transparent ciphertext -> error (sunscreen/tests/features.rs:8-34)
    with pytest.raises(RuntimeError, match="transparent"):
        o.multiply_plain(ca, np.zeros(1, dtype=np.uint64))
    # every Evaluator operation carries the same check in a SEAL built with SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT
    # (seal_fhe/build.rs:46-66): x - x, and anything computed from a transparent operand alone
    with pytest.raises(RuntimeError, match="transparent"):
        o.sub(ca, ca)
    zero_ct = np.zeros_like(ca)
    zero_ct[0] = ca[0]  # (c0, 0): transparent
    for op in (lambda: o.negate(zero_ct), lambda: o.add(zero_ct, zero_ct), lambda: o.multiply(zero_ct, zero_ct),
               lambda: o.add_plain(zero_ct, np.ones(1, dtype=np.uint64))):
        with pytest.raises(RuntimeError, match="transparent"):
            op()
    assert (o.add(zero_ct, ca)[1] == ca[1]).all()  # a transparent OPERAND is fine as long as the result is not


def test_rotate_rows_and_columns(unit):
    # seal_fhe/src/bfv_evaluator.rs:880-970: negative steps rotate right
    o, sk, pk, rk, gk = unit
    n, h = o.n, o.n // 2
    a = np.arange(n, dtype=np.int64)
    ca = enc(o, pk, a)
    c = dec(o, sk, o.rotate_rows(ca, -1, gk))
    assert a[0] == c[1] and a[1] == c[2] and a[h] == c[h + 1] and a[h + 1] == c[h + 2]
    assert (c == np.concatenate([np.roll(a[:h], 1), np.roll(a[h:], 1)])).all()
    c = dec(o, sk, o.rotate_columns(ca, gk))
    assert (c == np.concatenate([a[h:], a[:h]])).all()
    # non power-of-two steps go through the NAF chain (run.rs:793-881 rotates by 3)
    for steps in (3, -3, 5, 7, -11):
        c = dec(o, sk, o.rotate_rows(ca, steps, gk))
        assert (c == np.concatenate([np.roll(a[:h], -steps), np.roll(a[h:], -steps)])).all(), steps


def test_rotate_missing_key_is_an_error(unit):
    o, sk, pk, rk, gk = unit
    ca = enc(o, pk, make_small_vec(o.n))
    with pytest.raises(RuntimeError, match="missing key"):
        o.rotate_rows(ca, 1, {})


def test_slot_modulus_wrap_and_relin_noise_invariance():
    # seal_fhe/tests/assumptions.rs:5-34,84-107,138-194  (n=8192, bfv_default, batching bits 17)
    o = oracle_for("default_8192_17")
    assert o.t == 114689
    O.seed(77)
    sk, pk, rk, _ = o.keygen()
    v = np.full(o.n, 10_000, dtype=np.uint64)
    c1, c2 = o.encrypt(pk, o.batch_encode(v)), o.encrypt(pk, o.batch_encode(v))
    m = o.multiply(c1, c2)
    assert (o.batch_decode(o.decrypt(m, sk)) == 105_881).all()
    pre = o.noise_budget(m, sk)
    r = o.relinearize(m, rk)
    post = o.noise_budget(r, sk)
    assert (o.batch_decode(o.decrypt(r, sk)) == 105_881).all()
    assert pre == post  # "relinearization_consumes_no_noise_budget"
    # add noise <= operands (assumptions.rs:196-247)
    s = o.add(c1, c2)
    assert o.noise_budget(s, sk) >= min(o.noise_budget(c1, sk), o.noise_budget(c2, sk)) - 1


@pytest.mark.parametrize("n,expected_bits", [(4096, 26), (8192, 28), (16384, 29)])
def test_mul_relin_noise_cost_matches_published_table(n, expected_bits):
    """sunscreen_docs Tables_of_things.md:7-14: one mul+relin costs ~26/28/29 bits at the minimum
    batching modulus.  The oracle must land within 2 bits of the published cost."""
    bits = {4096: 16, 8192: 17, 16384: 17}[n]  # smallest bit size that admits a batching prime == 1 mod 2n
    primes = O.bfv_default(n)
    t = O.plain_batching(n, bits)
    o = O.Oracle(n, primes, t)
    O.seed(n)
    sk, pk, rk, _ = o.keygen()
    rng = np.random.default_rng(n)
    va = rng.integers(0, t, o.n).astype(np.uint64)
    vb = rng.integers(0, t, o.n).astype(np.uint64)
    ca, cb = o.encrypt(pk, o.batch_encode(va)), o.encrypt(pk, o.batch_encode(vb))
    fresh = min(o.noise_budget(ca, sk), o.noise_budget(cb, sk))
    r = o.relinearize(o.multiply(ca, cb), rk)
    assert (o.batch_decode(o.decrypt(r, sk)) == (va.astype(object) * vb.astype(object)) % t).all()
    cost = fresh - o.noise_budget(r, sk)
    assert abs(cost - expected_bits) <= 2, cost


def test_simple_multiply_config0():
    """BASELINE.json configs[0]: examples/simple_multiply (15 * 5 = 75) at n=4096, t=262144.
    Signed encoding = binary digits as polynomial coefficients (sunscreen/src/types/bfv/signed.rs:83-115)."""
    o = oracle_for("simple_multiply")
    O.seed(4096)
    sk, pk, rk, _ = o.keygen()

    def enc_signed(v):
        p = np.zeros(o.n, dtype=np.uint64)
        for i in range(64):
            if (abs(v) >> i) & 1:
                p[i] = 1 if v >= 0 else o.t - 1
        return p

    ca, cb = o.encrypt(pk, enc_signed(15)), o.encrypt(pk, enc_signed(5))
    r = o.relinearize(o.multiply(ca, cb), rk)
    p = o.decrypt(r, sk).astype(np.int64)
    p = np.where(p > o.t // 2, p - o.t, p)
    assert sum(int(c) << i for i, c in enumerate(p[:128])) == 75


def test_symmetric_encryption_and_no_special_prime_context():
    # n=1024 default has a single prime: no key switching possible, but encrypt/add work
    n = 1024
    o = O.Oracle(n, O.bfv_default(n), 64)
    O.seed(5)
    sk, pk, rk, _ = o.keygen()
    assert rk is None and o.K == 1 and o.KK == 1
    v = np.arange(n, dtype=np.uint64) % 7
    c = o.encrypt_symmetric(sk, v)
    assert (o.decrypt(c, sk) == v).all()
    c2 = o.add(c, o.encrypt(pk, v))
    assert (o.decrypt(c2, sk) == (2 * v) % o.t).all()


def test_multiply_many_and_exponentiate(unit):
    """seal_fhe/src/bfv_evaluator.rs:445-480 can_multiply_many (a*b*c*d slot-wise) and can_exponentiate; plus the structure of
    SEAL's work list: an odd last operand enters at the LAST product."""
    o, sk, pk, rk, gk = unit
    a = make_small_vec(o.n)
    cts = [enc(o, pk, a) for _ in range(4)]
    out = o.multiply_many(cts, rk)
    assert out.shape[0] == 2
    assert (dec(o, sk, out) == a * a * a * a).all()
    e = o.exponentiate(cts[0], 2, rk)
    assert (dec(o, sk, e) == a * a).all()
    assert (e == o.relinearize(o.multiply(cts[0], cts[0]), rk)).all()
    assert (o.exponentiate(cts[0], 1, rk) == cts[0]).all()
    with pytest.raises(ValueError):
        o.exponentiate(cts[0], 0, rk)
    three = o.multiply_many(cts[:3], rk)
    assert (three == o.relinearize(o.multiply(o.relinearize(o.multiply(cts[0], cts[1]), rk), cts[2]), rk)).all()
    b = np.sign(a)
    five = [enc(o, pk, b) for _ in range(5)]
    p01 = o.relinearize(o.multiply(five[0], five[1]), rk)
    p23 = o.relinearize(o.multiply(five[2], five[3]), rk)
    # work list [p01, p23, c4] -> append p01*p23 -> [.., c4, p0123] -> append c4 * p0123
    want = o.relinearize(o.multiply(five[4], o.relinearize(o.multiply(p01, p23), rk)), rk)
    assert (o.multiply_many(five, rk) == want).all()
