"""The oracle's BEHZ multiply against a second, structurally different restatement: the same algorithm over the INTEGERS.

Every RNS step of SEAL's bfv_multiply (native/src/seal/evaluator.cpp bfv_multiply + util/rnstool.cpp fastbconv_m_tilde, sm_mrq,
fast_floor, fastbconv_sk; bound by seal_fhe/src/evaluator_base.rs:198-212; the algorithm is Bajard-Eynard-Hasan-Zucca, SAC 2016)
has an integer meaning that does not mention the auxiliary base at all:

  v   = sum_i [x_i * m~ * (q/q_i)^-1]_{q_i} * (q/q_i)                    fastbconv_m_tilde: x*m~ + (multiple of q), 0 <= v < K*q
  x'  = (v + q * r) / m~,  r = [-v / q]_{m~} centred                     sm_mrq: x' == x (mod q), |x'| <= q/2 * (1 + 2K/m~)
  c'  = sum over cross terms of the negacyclic products x'_a * x'_b      the tensor, over Z
  T   = t * c'
  F   = (T - sum_i [T_i * (q/q_i)^-1]_{q_i} * (q/q_i)) / q               fast_floor: floor(T/q) - a', a' in [0, K)
  out = F mod q_i                                                        fastbconv_sk: exact while the base covers |F|

The oracle (oracle/ora_eval.c) and the library do this in residues with an auxiliary base (SEAL's 61-bit one in the oracle, the
library's own in libhipbfv); here it is done with Python integers and Kronecker substitution for the products, for random
operands and for the operands that drive every intermediate to its largest magnitude.  Agreement pins the oracle's multiply on
something that is not a transcription of it -- and is the direct form of DESIGN.md section 4's claim that the product does not
depend on WHICH auxiliary primes are used.

Test infrastructure: imports oracle/ as the thing under test.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bfv_oracle as O  # noqa: E402

M_TILDE = 1 << 32


def _prod(xs):
    r = 1
    for x in xs:
        r *= x
    return r


def _crt_terms(q):
    Q = _prod(q)
    return Q, [(Q // p, pow(Q // p, -1, p)) for p in q]


def _fast_conv_integer(res, q, terms):
    """The integer a fast base conversion stands for: sum_i [x_i * (Q/q_i)^-1]_{q_i} * (Q/q_i)  (= [x]_Q + a*Q, 0 <= a < K)."""
    return sum(((int(x) * inv) % p) * punct for x, p, (punct, inv) in zip(res, q, terms))


def _negacyclic(a: list[int], b: list[int], bound_bits: int) -> list[int]:
    """a * b mod (X^n + 1) over Z by Kronecker substitution: signed coefficients packed at a stride wide enough for the sums."""
    n = len(a)
    w = (bound_bits + 7) // 8 * 8  # slot width in bits, a whole number of bytes
    step = w // 8
    half = 1 << (w - 1)
    # pack through bytes (linear time): every slot offset by 2^(w-1) to be non-negative, the offsets subtracted as one integer
    offs = int.from_bytes(half.to_bytes(step, "little") * n, "little")
    A = int.from_bytes(b"".join((int(v) + half).to_bytes(step, "little") for v in a), "little") - offs
    B = int.from_bytes(b"".join((int(v) + half).to_bytes(step, "little") for v in b), "little") - offs
    R = A * B
    # make every digit of the product non-negative the same way, peel bytes, subtract again
    R += int.from_bytes(half.to_bytes(step, "little") * (2 * n), "little")
    raw = R.to_bytes(2 * n * step + 8, "little")
    assert not any(raw[2 * n * step :])
    d = [int.from_bytes(raw[k * step : (k + 1) * step], "little") - half for k in range(2 * n)]
    return [d[k] - d[k + n] for k in range(n)]


def behz_multiply_over_the_integers(a: np.ndarray, b: np.ndarray, q: list[int], t: int) -> np.ndarray:
    """a, b: uint64[size][K][n] (coefficient form) -> uint64[size_a + size_b - 1][K][n], by the integer steps above."""
    n = a.shape[2]
    K = len(q)
    Q, terms = _crt_terms(q)
    neg_inv_q_mod_mt = (-pow(Q, -1, M_TILDE)) % M_TILDE

    def lift(ct):
        polys = []
        for poly in ct:
            xs = []
            for k in range(n):
                res = [(int(poly[i][k]) * M_TILDE) % q[i] for i in range(K)]
                v = _fast_conv_integer(res, q, terms)
                r = (v * neg_inv_q_mod_mt) % M_TILDE
                if r >= M_TILDE // 2:
                    r -= M_TILDE
                num = v + Q * r
                assert num % M_TILDE == 0
                x = num // M_TILDE
                assert all(x % q[i] == int(poly[i][k]) for i in range(K))
                assert 2 * abs(x) <= Q + (2 * K * Q) // M_TILDE + 2
                xs.append(x)
            polys.append(xs)
        return polys

    A, B = lift(a), lift(b)
    bits = 2 * (Q.bit_length() + 1) + n.bit_length() + 8
    size = len(A) + len(B) - 1
    out = np.zeros((size, K, n), dtype=np.uint64)
    for j in range(size):
        c = [0] * n
        for ia in range(len(A)):
            ib = j - ia
            if 0 <= ib < len(B):
                d = _negacyclic(A[ia], B[ib], bits)
                c = [u + v for u, v in zip(c, d)]
        for k in range(n):
            T = t * c[k]
            conv = _fast_conv_integer([T % p for p in q], q, terms)
            assert (T - conv) % Q == 0
            F = (T - conv) // Q
            for i in range(K):
                out[j, i, k] = F % q[i]
    return out


CASES = [
    ("n1024_2x30", 1024, [30, 30, 31], 16, 2, 2),
    ("n1024_1x27", 1024, [27, 28], 14, 2, 2),
    ("n2048_54_55", 2048, [54, 55], 16, 2, 2),
    ("n1024_3x40_size3x2", 1024, [40, 41, 40, 42], 20, 3, 2),
    ("n4096_default", 4096, None, 17, 2, 2),
    ("n8192_default_t20", 8192, None, 20, 2, 2),
    ("n8192_3x54", 8192, [54, 54, 54, 56], 20, 2, 2),
    ("n16384_default_t20", 16384, None, 20, 2, 2),  # K = 8: the configuration whose auxiliary base the library shortened
]


@pytest.mark.parametrize("name,n,bits,tbits,sa,sb", CASES, ids=[c[0] for c in CASES])
def test_oracle_multiply_is_the_integer_algorithm(name, n, bits, tbits, sa, sb):
    primes = O.bfv_default(n) if bits is None else O.coeff_modulus_create(n, bits)
    t = O.plain_batching(n, tbits)
    o = O.Oracle(n, primes, t)
    o.throw_on_transparent = False
    q = [int(p) for p in primes[: o.K]]
    Q = _prod(q)
    rng = np.random.default_rng(n + tbits)

    def rand(size):
        return np.stack([rng.integers(0, p, (size, n), dtype=np.uint64) for p in q], axis=1)

    half = Q // 2
    sign = np.where(np.arange(n) % 2 == 0, 1, -1)

    def const(size, value_even, value_odd=None):
        ct = np.zeros((size, len(q), n), dtype=np.uint64)
        for i, p in enumerate(q):
            ct[:, i, :] = np.where(sign > 0, value_even % p, (value_even if value_odd is None else value_odd) % p).astype(np.uint64)
        return ct

    operands = [
        (rand(sa), rand(sb)),
        (const(sa, half), const(sb, half)),                    # every coefficient floor(q/2): the largest sums
        (const(sa, half, Q - half), const(sb, half)),          # ... with alternating signs
        (const(sa, Q - 1), const(sb, Q - 1)),                  # -1 everywhere
        (const(sa, half + 1), rand(sb)),
    ]
    if n >= 16384:
        operands = operands[:2]  # random + the extreme (9 s each in Python)
    for idx, (a, b) in enumerate(operands):
        want = behz_multiply_over_the_integers(a, b, q, t)
        got = o.multiply(a, b)
        assert got.shape == want.shape
        assert (got == want).all(), (name, idx)
