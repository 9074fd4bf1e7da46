"""Known-answer tests that pin the CPU oracle to the reference's own golden values.

Sources (all paths relative to the reference tree):
  * seal_fhe/src/modulus.rs:279-313            PlainModulus::batching / CoeffModulus::create answers
  * seal_fhe/src/encryption_parameters.rs:340-365   bfv_default(1024, {128,192,256}) and create(8192,[50,30,30,50,50])
  * logproof/src/rings.rs:36-125               SEAL default moduli for n = 1024..8192
  * seal_fhe/tests/assumptions.rs:109-136      batching(8192,17) == 114689
  * seal_fhe/tests/data/{secret,public}_key.bin     NTT convention + layout (via tests/golden/seal_key_fixture.npz)
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import bfv_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "seal_key_fixture.npz")


def test_plain_modulus_batching_known_answers():
    assert O.plain_batching(1024, 20) == 1038337  # modulus.rs:283
    assert O.plain_batching(8192, 17) == 114689  # assumptions.rs:109-136
    assert O.plain_batching(8192, 20) == 1032193  # logproof/tests/seal.rs:44-54
    assert O.plain_batching(16384, 17) == 65537


def test_coeff_modulus_create_known_answer():
    # encryption_parameters.rs:340-365 / modulus.rs:300-313
    assert O.coeff_modulus_create(8192, [50, 30, 30, 50, 50]) == [
        1125899905744897,
        1073643521,
        1073692673,
        1125899906629633,
        1125899906826241,
    ]


def test_bfv_default_tables():
    assert O.bfv_default(1024, 128) == [132120577]  # modulus.rs:291-297
    assert O.bfv_default(1024, 192) == [520193]
    assert O.bfv_default(1024, 256) == [12289]
    # logproof/src/rings.rs:52-58,80,102,124
    assert O.bfv_default(2048) == [0x3FFFFFFF000001]
    assert O.bfv_default(4096) == [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001]
    assert O.bfv_default(8192) == [0x7FFFFFD8001, 0x7FFFFFC8001, 0xFFFFFFFC001, 0xFFFFFF6C001, 0xFFFFFEBC001]
    for n, count, bits in [(1024, 1, 27), (2048, 1, 54), (4096, 3, 109), (8192, 5, 218), (16384, 9, 438), (32768, 16, 881)]:
        ps = O.bfv_default(n)
        assert len(ps) == count
        assert sum(p.bit_length() for p in ps) == bits  # SEAL's max bit counts for 128-bit security
        assert all(O.is_prime(p) and (p - 1) % (2 * n) == 0 for p in ps)


def test_primality_against_sympy_free_reference():
    # Compare against a slow trial-division/pow based check on a window of candidates.
    def slow(v):
        if v < 2:
            return False
        i = 2
        while i * i <= v:
            if v % i == 0:
                return False
            i += 1
        return True

    for v in list(range(0, 2000)) + [65537, 65539, 1038337, 1038339, 2**31 - 1, 2**31 + 1]:
        assert O.is_prime(v) == slow(v), v
    # Carmichael numbers and strong pseudoprimes to small bases
    for v in [561, 1105, 1729, 2047, 3215031751, 3825123056546413051]:
        assert not O.is_prime(v)


def test_minimal_root_is_minimal_and_primitive():
    for n, q in [(8, 17), (16, 97), (1024, 132120577), (4096, 0xFFFFEE001)]:
        r = O.minimal_primitive_root(2 * n, q)
        assert pow(r, n, q) == q - 1
        if n <= 16:
            cands = [x for x in range(2, q) if pow(x, n, q) == q - 1]
            assert r == min(cands)


@pytest.fixture(scope="module")
def fixture_ctx():
    g = np.load(GOLDEN)
    primes = [int(p) for p in g["primes"]]
    o = O.Oracle(8192, primes, O.plain_batching(8192, 32))
    return g, o


def test_forward_ntt_reproduces_seal_secret_key_bits(fixture_ctx):
    """NTT_oracle(ternary sk) must equal, bit for bit, the NTT-form secret key SEAL serialised."""
    g, o = fixture_ctx
    s = g["sk_ternary"].astype(np.int64)
    assert set(np.unique(s)).issubset({-1, 0, 1})
    for j, q in enumerate(o.key_primes):
        x = np.where(s < 0, q + s, s).astype(np.uint64)
        y = o.ntt(j, x)
        assert [int(v) for v in y[:8]] == [int(v) for v in g["sk_head"][j]]
        assert hashlib.sha256(y.astype("<u8").tobytes()).hexdigest() == str(g["sk_sha256"][j])
        assert (o.ntt(j, y, inverse=True) == x).all()


@pytest.mark.skipif(not os.path.exists("/root/reference/seal_fhe/tests/data/public_key.bin"), reason="reference tree absent")
def test_public_key_fixture_direct(fixture_ctx):
    """With the reference present: INTT(pk0 + pk1 (.) sk) is the same small error for all primes."""
    import importlib.util

    spec = importlib.util.spec_from_file_location(
        "mk", os.path.join(os.path.dirname(__file__), "golden", "make_seal_fixture_vectors.py")
    )
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    g, o = fixture_ctx
    pk = mk.find_payload(mk.load_seal_object(os.path.join(mk.REF, "public_key.bin")), 2 * 5 * 8192).reshape(2, 5, 8192)
    sk = mk.find_payload(mk.load_seal_object(os.path.join(mk.REF, "secret_key.bin")), 5 * 8192).reshape(5, 8192)
    for j, q in enumerate(o.key_primes):
        assert hashlib.sha256(pk[0, j].astype("<u8").tobytes()).hexdigest() == str(g["pk_sha256"][0][j])
        d = np.array([(int(a) + int(b) * int(c)) % q for a, b, c in zip(pk[0, j], pk[1, j], sk[j])], dtype=np.uint64)
        e = o.ntt(j, d, inverse=True).astype(np.int64)
        e = np.where(e > q // 2, e - q, e)
        assert (e == g["pk_err"].astype(np.int64)).all()


def test_ntt_matches_naive_evaluation():
    """Forward output slot i holds x(psi^(2*bitrev(i)+1)) -- checked against a naive O(n^2) evaluation."""
    n = 64
    primes = O.coeff_modulus_create(n, [30, 30])
    o = O.Oracle(n, primes, 257)
    rng = np.random.default_rng(7)
    for j, q in enumerate(primes):
        psi = O.minimal_primitive_root(2 * n, q)
        x = rng.integers(0, q, n).astype(np.uint64)
        y = o.ntt(j, x)
        for i in range(n):
            e = 2 * int(format(i, "06b")[::-1], 2) + 1
            w = pow(psi, e, q)
            acc = 0
            for k in range(n - 1, -1, -1):
                acc = (acc * w + int(x[k])) % q
            assert acc == int(y[i])


def test_ntt_is_negacyclic_convolution():
    n = 128
    primes = O.coeff_modulus_create(n, [40, 40])
    o = O.Oracle(n, primes, 257)
    rng = np.random.default_rng(8)
    q = primes[0]
    a = rng.integers(0, q, n).astype(np.uint64)
    b = rng.integers(0, q, n).astype(np.uint64)
    fa, fb = o.ntt(0, a), o.ntt(0, b)
    prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    c = o.ntt(0, prod, inverse=True)
    ref = [0] * n
    for i in range(n):
        for k in range(n):
            v = int(a[i]) * int(b[k])
            if i + k >= n:
                ref[i + k - n] -= v
            else:
                ref[i + k] += v
    assert [r % q for r in ref] == [int(v) for v in c]


def test_aux_base_matches_seal_rule():
    """RNSTool: aux primes are the first |B|+2 61-bit primes == 1 mod 2n, as [m_sk, gamma, B...]."""
    for n in (4096, 8192, 16384):
        primes = O.bfv_default(n)
        o = O.Oracle(n, primes, O.plain_batching(n, 17))
        aux = O.get_primes(2 * n, 61, o.K + 3)
        nB = len(o.bsk) - 1
        assert nB in (o.K, o.K + 1)
        assert o.bsk[-1] == aux[0] and o.gamma == aux[1] and o.bsk[:-1] == aux[2 : 2 + nB]
        grow = 32 + o.t.bit_length() + o.total_coeff_bits >= 61 * o.K + 61
        assert nB == o.K + (1 if grow else 0)
