"""The oracle's hybrid key switch (relinearize, apply_galois, mod_switch_to_next) against the same steps over the INTEGERS.

SEAL 4.0 Evaluator::switch_key_inplace (native/src/seal/evaluator.cpp; bound by seal_fhe/src/evaluator_base.rs:214-240 relinearize
and :300-407 rotations; one special prime p, digits = the RNS residues of the polynomial being switched) is, in integers:

  d_i   = the polynomial's residues mod q_i, read as integers in [0, q_i)                     (RNS decomposition)
  X     = sum_i d_i * ksk_i            in Z[X]/(X^n + 1), coefficients mod Q*p                  (the key product, per key component)
  out   = floor((X + floor(p/2)) / p)  mod q_j                                                  (mod-down by the special prime: SEAL adds
                                                                                                 p >> 1 before it subtracts the residue mod p)
  relinearize:  (c0 + out_0, c1 + out_1) from c2;   apply_galois: (sigma(c0) + out_0, out_1) from sigma(c1)
  mod_switch_to_next (SEAL divide_and_round_q_last_inplace): c -> floor((c + floor(q_last/2)) / q_last) per coefficient, same form

Here the products are Kronecker-substitution products of Python integers and the keys are CRT-composed from the oracle's
key arrays (brought to coefficient form with the oracle's inverse NTT, which tests/test_oracle_kat.py pins on the SEAL key
fixture and on a naive evaluation).  Random operands and every-residue-maximal operands; the oracle must give the same bits.

Test infrastructure: imports oracle/ as the thing under test.
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


def _crt_compose(residues: list[np.ndarray], primes: list[int]) -> list[int]:
    """residues[j][k] = x_k mod primes[j] -> the integers x_k in [0, prod)."""
    P = _prod(primes)
    coef = [(P // p) * pow(P // p, -1, p) for p in primes]
    n = len(residues[0])
    return [sum(int(residues[j][k]) * coef[j] for j in range(len(primes))) % P for k in range(n)]


def _key_coefficients(o, key: np.ndarray) -> list[list[list[int]]]:
    """key: uint64[K][2][KK][n] in NTT form -> [digit][component] -> n integers mod Q*p (coefficient form)."""
    out = []
    for i in range(o.K):
        comps = []
        for c in range(2):
            res = [o.ntt(j, key[i][c][j], inverse=True) for j in range(o.KK)]
            comps.append(_crt_compose(res, o.key_primes))
        out.append(comps)
    return out


def _switch_over_the_integers(o, poly: np.ndarray, keyc) -> list[np.ndarray]:
    """poly: uint64[K][n] (coefficient form) -> the two polynomials the key switch adds, uint64[K][n] each."""
    n, K = o.n, o.K
    q = o.key_primes[:K]
    p = o.key_primes[-1]
    QP = _prod(o.key_primes)
    half = p >> 1
    bits = max(q).bit_length() + QP.bit_length() + n.bit_length() + 4
    outs = []
    for c in range(2):
        acc = [0] * n
        for i in range(K):
            d = [int(v) for v in poly[i]]
            prod = _negacyclic(d, keyc[i][c], bits)
            acc = [u + v for u, v in zip(acc, prod)]
        res = np.zeros((K, n), dtype=np.uint64)
        for k in range(n):
            X = acc[k] % QP
            y = (X + half) // p
            for j in range(K):
                res[j, k] = y % q[j]
        outs.append(res)
    return outs


def _galois_poly(poly_row: np.ndarray, elt: int, qj: int) -> np.ndarray:
    """x(X) -> x(X^elt) mod (X^n + 1), coefficients mod qj."""
    n = len(poly_row)
    out = np.zeros(n, dtype=np.uint64)
    for k in range(n):
        e = (k * elt) % (2 * n)
        v = int(poly_row[k])
        if e >= n:
            out[e - n] = (qj - v) % qj
        else:
            out[e] = v
    return out


CASES = [
    ("n1024_2x30", 1024, [30, 30, 31], 16),
    ("n2048_3x36", 2048, [36, 36, 36, 37], 16),
    ("n4096_default", 4096, None, 17),
    ("n8192_default", 8192, None, 20),
    ("n8192_3x54", 8192, [54, 54, 54, 56], 20),
]


def _operands(o, rng, size):
    q = o.key_primes[: o.K]
    rand = np.stack([rng.integers(0, p, (size, o.n), dtype=np.uint64) for p in q], axis=1)
    top = np.stack([np.full((size, o.n), p - 1, dtype=np.uint64) for p in q], axis=1)
    return [rand, top]


@pytest.mark.parametrize("name,n,bits,tbits", CASES, ids=[c[0] for c in CASES])
def test_relinearize_is_the_integer_key_switch(name, n, bits, tbits):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    o = O.Oracle(n, primes, O.plain_batching(n, tbits))
    o.throw_on_transparent = False
    O.seed(n + tbits)
    sk, pk, rk, gk = o.keygen()
    keyc = _key_coefficients(o, rk)
    q = o.key_primes[: o.K]
    rng = np.random.default_rng(n)
    for idx, ct3 in enumerate(_operands(o, rng, 3)):
        add0, add1 = _switch_over_the_integers(o, ct3[2], keyc)
        want = np.zeros((2, o.K, n), dtype=np.uint64)
        for j, qj in enumerate(q):
            want[0, j] = (ct3[0, j].astype(object) + add0[j].astype(object)) % qj
            want[1, j] = (ct3[1, j].astype(object) + add1[j].astype(object)) % qj
        got = o.relinearize(ct3, rk)
        assert (got == want).all(), (name, idx)


@pytest.mark.parametrize("name,n,bits,tbits", CASES[:4], ids=[c[0] for c in CASES[:4]])
def test_apply_galois_is_permutation_plus_integer_key_switch(name, n, bits, tbits):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    o = O.Oracle(n, primes, O.plain_batching(n, tbits))
    o.throw_on_transparent = False
    O.seed(n + 1)
    elts = [3, 2 * n - 1]
    sk, pk, rk, gk = o.keygen(relin=False, galois_elts=elts)
    q = o.key_primes[: o.K]
    rng = np.random.default_rng(n + 7)
    for elt in elts:
        keyc = _key_coefficients(o, gk[elt])
        for idx, ct in enumerate(_operands(o, rng, 2)):
            s0 = np.stack([_galois_poly(ct[0, j], elt, qj) for j, qj in enumerate(q)])
            s1 = np.stack([_galois_poly(ct[1, j], elt, qj) for j, qj in enumerate(q)])
            add0, add1 = _switch_over_the_integers(o, s1, keyc)
            want = np.zeros((2, o.K, n), dtype=np.uint64)
            for j, qj in enumerate(q):
                want[0, j] = (s0[j].astype(object) + add0[j].astype(object)) % qj
                want[1, j] = add1[j]
            got = o.apply_galois(ct, elt, gk)
            assert (got == want).all(), (name, elt, idx)


@pytest.mark.parametrize("name,n,bits,tbits", [CASES[1], CASES[3], CASES[4]], ids=[CASES[1][0], CASES[3][0], CASES[4][0]])
def test_mod_switch_to_next_is_the_integer_rounding(name, n, bits, tbits):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    o = O.Oracle(n, primes, O.plain_batching(n, tbits))
    o.throw_on_transparent = False
    q = o.key_primes[: o.K]
    last = q[-1]
    rng = np.random.default_rng(n + 11)
    for idx, ct in enumerate(_operands(o, rng, 2)):
        want = np.zeros((2, o.K - 1, n), dtype=np.uint64)
        for c in range(2):
            xs = _crt_compose([ct[c, j] for j in range(o.K)], q)
            for k in range(n):
                y = (xs[k] + (last >> 1)) // last
                for j in range(o.K - 1):
                    want[c, j, k] = y % q[j]
        got = o.mod_switch_to_next(ct)
        assert got.shape == want.shape
        assert (got == want).all(), (name, idx)
