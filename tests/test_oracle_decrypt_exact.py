"""The oracle's decryption (the step after the hot path that the reference's decrypt-and-compare tests go through) against its
integer meaning.  SEAL 4.0 Decryptor::bfv_decrypt + RNSTool::decrypt_scale_and_round (seal_fhe/src/encryptor_decryptor.rs:560-640):

  x   = c0 + c1*s (+ c2*s^2)  in Z[X]/(X^n+1) mod q                                 dot product with the secret key
  D   = sum_i [t*gamma*x_i * (q/q_i)^-1]_{q_i} * (q/q_i)                            fast conversion of t*gamma*x to {t, gamma}
  W   = (t*gamma*x - D) / q                                                          = floor(t*gamma*x / q) - a: what "-D/q" is mod t, mod gamma
  m   = (W - centred(W mod gamma)) / gamma  mod t                                    the gamma correction

and, for a ciphertext whose noise is inside its budget, m = round(t * centred(x) / q) mod t -- BFV decryption proper.
Python integers and Kronecker products; valid ciphertexts after 0, 1 and 2 multiplications and arbitrary (invalid) residues.
Test infrastructure (imports oracle/).
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
from test_oracle_behz_exact import _crt_terms, _fast_conv_integer, _negacyclic, _prod  # noqa: E402
from test_oracle_keyswitch_exact import _crt_compose  # noqa: E402


def _secret_coefficients(o, sk) -> list[int]:
    """sk: uint64[KK][n] in NTT form -> the ternary polynomial as integers in {-1, 0, 1}."""
    s0 = o.ntt(0, sk[0], inverse=True)
    q0 = o.key_primes[0]
    out = [1 if int(v) == 1 else (-1 if int(v) == q0 - 1 else 0) for v in s0]
    assert all(int(v) in (0, 1, q0 - 1) for v in s0)
    return out


def _dot_over_the_integers(o, ct, s) -> list[int]:
    q = o.key_primes[: o.K]
    Q = _prod(q)
    n = o.n
    bits = Q.bit_length() + n.bit_length() * 3 + 8
    acc = _crt_compose([ct[0, j] for j in range(o.K)], q)
    spow = list(s)
    for p in range(1, ct.shape[0]):
        c = _crt_compose([ct[p, j] for j in range(o.K)], q)
        acc = [u + v for u, v in zip(acc, _negacyclic(c, spow, bits))]
        if p + 1 < ct.shape[0]:
            spow = _negacyclic(spow, s, 4 * n.bit_length() + 8)
    return [v % Q for v in acc]


def _decrypt_over_the_integers(o, xs) -> tuple[np.ndarray, list[int]]:
    q = o.key_primes[: o.K]
    Q, terms = _crt_terms(q)
    t, gamma = o.t, o.gamma
    out = np.zeros(o.n, dtype=np.uint64)
    slack = []
    for k, x in enumerate(xs):
        d = [(t * gamma * x) % p for p in q]
        D = _fast_conv_integer(d, q, terms)
        assert (t * gamma * x - D) % Q == 0
        W = (t * gamma * x - D) // Q
        g = W % gamma
        if g > gamma >> 1:
            g -= gamma
        assert (W - g) % gamma == 0
        out[k] = ((W - g) // gamma) % t
        slack.append(g)
    return out, slack


@pytest.mark.parametrize("n,bits,tbits", [(1024, [30, 30, 31], 14), (4096, None, 17), (8192, None, 20)])
def test_decrypt_is_the_integer_algorithm_and_rounds_valid_ciphertexts(n, bits, tbits):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    o = O.Oracle(n, primes, O.plain_batching(n, tbits))
    o.throw_on_transparent = False
    O.seed(n)
    sk, pk, rk, gk = o.keygen()
    s = _secret_coefficients(o, sk)
    q = o.key_primes[: o.K]
    Q, t = _prod(q), o.t
    rng = np.random.default_rng(n)
    plain = rng.integers(0, t, n, dtype=np.uint64)
    ct = o.encrypt(pk, plain)
    cases = [("fresh", ct, True)]
    sq = o.multiply(ct, ct)
    cases.append(("squared, size 3", sq, o.noise_budget(sq, sk) > 0))
    if n >= 4096:
        r = o.relinearize(sq, rk)
        cases.append(("relinearized", r, o.noise_budget(r, sk) > 0))
    junk = np.stack([rng.integers(0, p, (2, n), dtype=np.uint64) for p in q], axis=1)
    cases.append(("arbitrary residues", junk, False))
    for label, c, valid in cases:
        xs = _dot_over_the_integers(o, c, s)
        for j, p in enumerate(q):  # the dot product itself, residue for residue
            assert (o.dot_with_secret(c, sk)[j] == np.array([x % p for x in xs], dtype=np.uint64)).all(), (label, j)
        want, slack = _decrypt_over_the_integers(o, xs)
        got = o.decrypt(c, sk)
        assert (got == want).all(), label
        if valid:
            # BFV decryption proper: round(t * centred(x) / q) mod t
            rounded = []
            for x in xs:
                xc = x - Q if x > Q // 2 else x
                rounded.append(((2 * t * xc + Q) // (2 * Q)) % t)
            assert (got == np.array(rounded, dtype=np.uint64)).all(), label
    assert (o.decrypt(ct, sk) == plain).all()
