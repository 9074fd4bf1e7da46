"""The size bound of the library's own BEHZ auxiliary base, replayed in exact integers (no GPU, no oracle).

sunscreen_amd/csrc/context.cpp picks the auxiliary base {B_1..B_nB, m_sk} of a context as the fewest / smallest FP64-eligible
primes with  prod(B) * m_sk >= 2^(bits(t) + log2 N + bits(q) + 3),  where SEAL reserves 32 bits instead of log2 N + 3
(SEAL native/src/seal/util/rnstool.cpp, RNSTool::initialize: "we require K * n * t * q^2 < q * prod(B) * m_sk ... We reserve
32 bits for K * n"; not vendored under /root/reference -- seal_fhe binds it through seal_fhe/src/evaluator_base.rs:198-212).
The residues every row carries are exact whatever the size of the integers behind them; the base size matters in ONE step,
the Shenoy-Kumaresan conversion Bsk -> q at the end of the multiply (rnstool.cpp fastbconv_sk), which recovers
F = floor(t*c/q) - a' only while |F| < B * (m_sk/2 - nB - 1).

This file takes the base the LIBRARY picks (hipbfv_debug_aux_base: Context::create run host-only) and runs fast_floor and
fastbconv_sk on Python integers, for integers t*c at and around the largest magnitude a tensor coefficient can have
(8 cross terms: size_a + size_b <= 16), and checks that the residues mod q that come out are those of F -- and that they are
NOT once the integer outgrows the base (so the check can fail).
"""
from __future__ import annotations

import ctypes as C
import os
import random

import pytest

from sunscreen_amd import _lib, seal

M_TILDE = 1 << 32


def _aux_base(n: int, primes: list[int], t: int) -> tuple[list[int], int]:
    L = _lib.load()
    arr = (C.c_uint64 * len(primes))(*primes)
    out = (C.c_uint64 * 32)()
    cnt = C.c_uint64()
    flags = C.c_int()
    rc = L.hipbfv_debug_aux_base(n, arr, len(primes), t, C.byref(cnt), out, 32, C.byref(flags))
    assert rc == 0, rc
    return [int(v) for v in out[: cnt.value]], flags.value


def _default_primes(n: int) -> list[int]:
    return [int(m.value()) for m in seal.CoefficientModulus.bfv_default(n)]


def _create_primes(n: int, bits: list[int]) -> list[int]:
    return [int(m.value()) for m in seal.CoefficientModulus.create(n, bits)]


def _plain_batching(n: int, bits: int) -> int:
    return int(seal.PlainModulus.batching(n, bits).value())


def _prod(xs):
    r = 1
    for x in xs:
        r *= x
    return r


def _fastbconv(res: list[int], base: list[int], targets: list[int]) -> list[int]:
    """rnstool.cpp BaseConverter::fast_convert: sum_i [x_i * (P/p_i)^-1]_{p_i} * (P/p_i) reduced by every target -- the integer
    behind it is [x]_P + a*P with a in [0, len(base))."""
    P = _prod(base)
    ys = [(x * pow(P // p, -1, p)) % p for x, p in zip(res, base)]
    return [sum(y * ((P // p) % m) for y, p in zip(ys, base)) % m for m in targets]


def floor_then_sk(T: int, q: list[int], B: list[int], m_sk: int) -> tuple[list[int], int]:
    """fast_floor (Bsk residues of floor(T/q) - a') followed by fastbconv_sk (back to q), on the exact residues of T.
    Returns (residues mod q_i of the result, the integer F = floor(T/q) - a' they should be residues of)."""
    Q = _prod(q)
    bsk = B + [m_sk]
    t_q = [T % p for p in q]
    t_bsk = [T % p for p in bsk]
    conv = _fastbconv(t_q, q, bsk)  # ([T]_q + a' * Q) mod Bsk
    f_bsk = [((tv - cv) * pow(Q, -1, p)) % p for tv, cv, p in zip(t_bsk, conv, bsk)]
    # the integer the residues stand for: a' from the exact sum
    ys = [(x * pow(Q // p, -1, p)) % p for x, p in zip(t_q, q)]
    a_prime = (sum(y * (Q // p) for y, p in zip(ys, q)) - (T % Q)) // Q
    assert 0 <= a_prime < len(q)
    F = (T - (T % Q)) // Q - a_prime
    assert all(F % p == fv for p, fv in zip(bsk, f_bsk))  # fast_floor itself never depends on the base size
    # Shenoy-Kumaresan: B -> q and m_sk, alpha from the m_sk residue, centred
    Bp = _prod(B)
    conv2 = _fastbconv(f_bsk[:-1], B, q + [m_sk])
    alpha = ((conv2[-1] - f_bsk[-1]) * pow(Bp, -1, m_sk)) % m_sk
    if alpha > m_sk // 2:
        alpha -= m_sk
    return [(cv - alpha * Bp) % p for cv, p in zip(conv2[:-1], q)], F


CONFIGS = [
    ("default_4096", 4096, None, 17),
    ("default_8192", 8192, None, 17),
    ("default_8192_t50", 8192, None, 50),
    ("default_16384", 16384, None, 17),
    ("default_16384_t45", 16384, None, 45),
    ("bits54x3_8192", 8192, [54, 54, 54, 56], 17),
    ("wide_4096", 4096, [58, 59, 60], 16),
]


@pytest.mark.parametrize("name,n,bits,tbits", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_own_base_holds_the_largest_floor_the_multiply_can_produce(name, n, bits, tbits):
    key = _default_primes(n) if bits is None else _create_primes(n, bits)
    t = _plain_batching(n, tbits)
    q = key[:-1]
    aux, flags = _aux_base(n, key, t)
    if not flags & (1 | 16):
        pytest.skip("SEAL's base for this parameter set (own base switched off or not eligible)")
    B, m_sk = aux[:-1], aux[-1]
    Q, K = _prod(q), len(q)
    assert len(set(aux)) == len(aux) and not set(aux) & set(key)
    assert all(p % (2 * n) == 1 and p < (1 << 48) for p in aux)
    reserve = n.bit_length() - 1 + 3
    assert (_prod(B) * m_sk).bit_length() > t.bit_length() + Q.bit_length() + reserve
    # the largest |t * c'|: operands after the Montgomery step |x'| <= q/2 + q*K/m~, 8 cross terms of N products each
    xmax = Q // 2 + (Q * K) // M_TILDE + 1
    Tmax = t * 8 * n * xmax * xmax
    # analytic form of the condition, with the slack the comment in context.cpp claims (a factor of two)
    Fmax = Tmax // Q + K + 1
    assert 2 * Fmax < _prod(B) * (m_sk // 2 - len(B) - 1)
    rng = random.Random(n * 131 + tbits)
    cases = [Tmax, -Tmax, Tmax - 1, -Tmax + 1, 0, 1, -1, Q, -Q, Q - 1, -Q - 1, Tmax // 4, -(Tmax // 4)]
    cases += [rng.randrange(-Tmax, Tmax + 1) for _ in range(200)]
    # integers whose floor sits right at a multiple of B (floor(F/B) changes there) and at multiples of q
    Bp = _prod(B)
    for k in (1, 2, -1, -2, rng.randrange(1, max(2, Fmax // Bp)), -rng.randrange(1, max(2, Fmax // Bp))):
        for d in (-1, 0, 1):
            cases.append((k * Bp + d) * Q + rng.randrange(Q))
    for T in cases:
        if abs(T) > Tmax:
            continue
        got, F = floor_then_sk(T, q, B, m_sk)
        assert got == [F % p for p in q], (name, T)


def test_the_replay_fails_once_the_integer_outgrows_the_base():
    """Sensitivity: with |F| beyond B * m_sk / 2 the Shenoy-Kumaresan correction wraps and the residues are wrong."""
    n = 8192
    key = _default_primes(n)
    t = _plain_batching(n, 17)
    q = key[:-1]
    aux, flags = _aux_base(n, key, t)
    B, m_sk = aux[:-1], aux[-1]
    Q = _prod(q)
    T = (_prod(B) * m_sk) * Q  # F ~ B * m_sk: alpha is off by m_sk
    got, F = floor_then_sk(T, q, B, m_sk)
    assert got != [F % p for p in q]
    T = (_prod(B) * (m_sk // 2 - len(B) - 2)) * Q  # the last F the analysis guarantees
    got, F = floor_then_sk(T, q, B, m_sk)
    assert got == [F % p for p in q]


def test_the_new_bound_saves_a_row_where_it_is_claimed_to():
    """n = 16384, SEAL default primes: nine auxiliary primes instead of ten (17 rows in the multiply instead of 18)."""
    if os.environ.get("HIPBFV_SEAL_AUX") == "1" or os.environ.get("HIPBFV_NO_F64") == "1":
        pytest.skip("base sizing switched by the environment")
    key = _default_primes(16384)
    aux, flags = _aux_base(16384, key, _plain_batching(16384, 17))
    assert flags & 1 and len(aux) == 9 and all(p < (1 << 48) for p in aux)


def test_a_plain_modulus_too_large_for_the_own_base_falls_back_to_seals():
    """A 60-bit plain modulus at n = 16384: the bound asks for more 48-bit primes than the 8-prime kernels carry; the context
    keeps SEAL's base (flags 0), it does not shrink the bound."""
    key = _default_primes(16384)
    aux, flags = _aux_base(16384, key, _plain_batching(16384, 60))
    if flags & 1:
        B, m_sk = aux[:-1], aux[-1]
        t = _plain_batching(16384, 60)
        assert (_prod(aux)).bit_length() > t.bit_length() + 14 + _prod(key[:-1]).bit_length() + 3
    else:
        assert all(p.bit_length() == 61 for p in aux)
