"""The oracle's plaintext operations against their integer meaning (SEAL 4.0 Evaluator::add_plain / sub_plain / multiply_plain,
bound by seal_fhe/src/evaluator_base.rs:242-298; scaling in util/scalingvariant.cpp multiply_add_plain_with_scaling_variant):

  add_plain / sub_plain:  c0 +/- (floor(q/t) * m + floor((m * (q mod t) + floor((t+1)/2)) / t))      = round(q*m/t), SEAL's rounding
  multiply_plain:         c_j * m~ in Z[X]/(X^n+1), m~_k = m_k - t for m_k >= floor((t+1)/2), else m_k (the centred lift) ...
  ... except a MONOMIAL plaintext (one non-zero coefficient), which SEAL multiplies through a negacyclic shift by the
      coefficient AS IT IS, without the centred lift (when t < every q_i): a different ciphertext, the same plaintext underneath.

Big-integer products by Kronecker substitution; the oracle must give the same bits.  Test infrastructure (imports oracle/).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import bfv_oracle as O  # noqa: E402
from test_oracle_behz_exact import _negacyclic, _prod  # noqa: E402
from test_oracle_keyswitch_exact import _crt_compose  # noqa: E402

CASES = [("n1024_2x30", 1024, [30, 30, 31], 16), ("n4096_default", 4096, None, 17), ("n8192_default", 8192, None, 20), ("n8192_3x54", 8192, [54, 54, 54, 56], 20)]


def _setup(n, bits, tbits):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    o = O.Oracle(n, primes, O.plain_batching(n, tbits))
    o.throw_on_transparent = False
    return o, o.key_primes[: o.K]


@pytest.mark.parametrize("name,n,bits,tbits", CASES, ids=[c[0] for c in CASES])
def test_add_and_sub_plain_add_seals_rounding_of_q_m_over_t(name, n, bits, tbits):
    o, q = _setup(n, bits, tbits)
    Q, t = _prod(q), o.t
    rng = np.random.default_rng(n)
    ct = np.stack([rng.integers(0, p, (2, n), dtype=np.uint64) for p in q], axis=1)
    plain = rng.integers(0, t, n, dtype=np.uint64)
    plain[:4] = [0, 1, t - 1, (t + 1) // 2]
    scaled = [(Q // t) * int(m) + (int(m) * (Q % t) + ((t + 1) >> 1)) // t for m in plain]
    for sub, fn in ((False, o.add_plain), (True, o.sub_plain)):
        want = ct.copy()
        for j, qj in enumerate(q):
            want[0, j] = [(int(c) + (-y if sub else y)) % qj for c, y in zip(ct[0, j], scaled)]
        assert (fn(ct, plain) == want).all(), (name, sub)


@pytest.mark.parametrize("name,n,bits,tbits", CASES, ids=[c[0] for c in CASES])
def test_multiply_plain_is_the_product_with_the_centred_lift_and_the_monomial_is_not(name, n, bits, tbits):
    o, q = _setup(n, bits, tbits)
    Q, t = _prod(q), o.t
    rng = np.random.default_rng(n + 3)
    ct = np.stack([rng.integers(0, p, (2, n), dtype=np.uint64) for p in q], axis=1)
    ct[1] = np.stack([np.full(n, p - 1, dtype=np.uint64) for p in q])  # one polynomial at the top of every residue
    xs = [_crt_compose([ct[c, j] for j in range(o.K)], q) for c in range(2)]
    bits_needed = Q.bit_length() + t.bit_length() + n.bit_length() + 4

    def product(plain_lifted):
        want = np.zeros_like(ct)
        for c in range(2):
            d = _negacyclic(xs[c], plain_lifted, bits_needed)
            for j, qj in enumerate(q):
                want[c, j] = [v % qj for v in d]
        return want

    half = (t + 1) >> 1
    dense = rng.integers(0, t, n, dtype=np.uint64)
    dense[:3] = [t - 1, half, half - 1]
    lifted = [int(m) - t if int(m) >= half else int(m) for m in dense]
    assert (o.multiply_plain(ct, dense) == product(lifted)).all(), (name, "dense")
    two = np.zeros(n, dtype=np.uint64)
    two[0], two[n - 1] = t - 1, 5
    assert (o.multiply_plain(ct, two) == product([-1] + [0] * (n - 2) + [5])).all(), (name, "two terms")
    # monomials: (t - 2) * X^e is multiplied in as the integer t - 2, NOT as -2 (every q_i exceeds t here)
    for e, coeff in ((0, t - 2), (7, half), (n - 1, 3)):
        mono = np.zeros(n, dtype=np.uint64)
        mono[e] = coeff
        plain_int = [0] * n
        plain_int[e] = int(coeff)
        got = o.multiply_plain(ct, mono)
        assert (got == product(plain_int)).all(), (name, "monomial", e)
        if coeff >= half:
            centred = [0] * n
            centred[e] = int(coeff) - t
            assert not (got == product(centred)).all()  # ... and the two really are different ciphertexts
