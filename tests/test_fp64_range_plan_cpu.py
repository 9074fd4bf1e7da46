"""The FP64 policy keeps residues as exact integers in doubles (sunscreen_amd/csrc/nttcore.hpp, ArithD).  Two things make that
safe and neither needs a GPU to check:

 (1) the twiddle product `mul_tw` -- T = Y*W - rint(fl(Y*W) * fl(1/q)) * q computed with an error-free split -- is EXACT and
     |T| <= q * (0.5 + |Y| * 1.5 * 2^-52): replayed here in exact integer arithmetic plus IEEE double roundings (Python floats),
     on random and on adversarial operands, for every prime the FP64 policy is used with;
 (2) the range plan (context.cpp: at which passes everything is reduced) keeps every intermediate below 2^53 under that
     bound: the plans come out of the library (hipbfv_debug_f64_plan, host only) and are replayed against a worst-case model
     written independently here, pass structure (nttshape.hpp) included.
"""
import ctypes as C
import math
import random

import pytest

from oracle import bfv_oracle as O

TWO52 = float(1 << 52)
LIMIT = 1 << 53


def _mul_tw(y: int, w: int, q: int):
    """ArithD::mul_tw with every rounding made explicit; returns (T, exact) where exact says no rounding lost anything."""
    qinv = 1.0 / float(q)                  # fl(1/q)
    xh = float(y) * float(w)               # fl(y*w): Python floats are IEEE doubles, round to nearest even
    xl_exact = y * w - int(xh)             # fma(y, w, -xh) computes this exactly when it is representable
    exact = float(xl_exact) == xl_exact and abs(xl_exact) < LIMIT
    qf = round(xh * qinv)                  # v_rndne_f64: round half to even, as Python's round() on a float
    t_exact = int(xh) - qf * q             # fma(-qf, q, xh): exact when the result is representable
    exact = exact and abs(t_exact) < LIMIT
    total = t_exact + xl_exact
    exact = exact and abs(total) < LIMIT   # the final add of two integers in doubles
    return total, exact


def _fp64_primes():
    primes = set()
    for n in (4096, 8192, 16384):
        primes.update(O.bfv_default(n))
    # the auxiliary bases are chosen below 2^48; the policy's hard limit is 2^50
    primes.update(O.get_primes(2 * 16384, 47, 3))
    primes.update(O.get_primes(2 * 16384, 50, 2))
    return sorted(p for p in primes if p < (1 << 50))


@pytest.mark.parametrize("q", _fp64_primes())
def test_twiddle_product_is_exact_and_within_the_priced_bound(q):
    rng = random.Random(q)
    ymax = int(0.98 * LIMIT)  # the plan keeps |Y| below 0.98 * 2^53
    ys = [ymax, -ymax, ymax - 1, q - 1, -(q - 1), 1, -1, 0, (q - 1) // 2, q // 2 + 1]
    ws = [q - 1, 1, (q - 1) // 2, (q + 1) // 2, 2, q - 2]
    cases = [(y, w) for y in ys for w in ws]
    cases += [(rng.randrange(-ymax, ymax + 1), rng.randrange(q)) for _ in range(4000)]
    # operands whose exact quotient sits next to a half-integer: the rounding of the estimate decides the result's sign
    for _ in range(2000):
        w = rng.randrange(1, q)
        k = rng.randrange(1, ymax // q)
        y = ((2 * k + 1) * q // 2 + rng.randrange(-2, 3)) * pow(w, -1, q) % q + q * rng.randrange(0, ymax // q)
        cases.append((y if rng.random() < 0.5 else -y, w))
    worst = 0.0
    for y, w in cases:
        t, exact = _mul_tw(y, w, q)
        assert exact, (y, w)
        assert (t - y * w) % q == 0, (y, w)
        bound = q * (0.5 + abs(y) * 1.5 / TWO52)
        assert abs(t) <= bound * (1 + 1e-12), (y, w, t, bound)
        worst = max(worst, abs(t) / bound)
    assert worst > 0.3  # the cases do reach into the bound


# ---- pass structure, restated from nttshape.hpp ----
def _whole_radices(logn, ept_log=4):
    npass = (logn + ept_log - 1) // ept_log
    return [logn // npass + (1 if p < logn % npass else 0) for p in range(npass)]


def _head_log(logn):
    return 2 if logn == 14 else 3


def _split_fwd_radices(logn):
    rem = logn - _head_log(logn)
    return {9: [3, 3, 3], 10: [3, 3, 2, 2], 11: [3, 3, 3, 2], 12: [3, 3, 3, 3]}[rem]


def _split_inv_radices(logn):
    rem = logn - 2
    return {10: [3, 3, 2, 2], 11: [2, 3, 3, 3], 12: [3, 3, 3, 3], 13: [3, 3, 3, 2, 2]}[rem]


class _Model:
    """Worst-case magnitudes in units of q.  A twiddle product of a value of magnitude m is within 0.5 + m*q*1.5*2^-52
    (test above); a reduction v - rint(v * fl(1/q)) * q estimates v/q with two roundings: within 0.5 + m * 2^-51 (loose)."""

    def __init__(self, q):
        self.q = q
        self.peak = 0.0

    def see(self, m):
        self.peak = max(self.peak, m)
        return m

    def tw(self, m):
        return 0.5 + m * self.q * 1.5 / TWO52

    def red(self, m):
        return 0.5 + m / float(1 << 51)

    def reduce_by_mask(self, m, mask, p, second=16):
        if (mask >> p) & 1:
            m = self.red(m)
        if (mask >> (p + second)) & 1:
            m = self.red(m)
        return m

    def fwd_stage(self, m):  # X' = X + T, Y' = X - T
        self.see(m)  # the operand of the product
        return self.see(m + self.tw(m))

    def inv_stage(self, m):  # X' = X + Y, Y' = (X - Y) * W
        d = self.see(2.0 * m)
        return max(d, self.tw(d))


def _plan(q, logn):
    out = (C.c_uint32 * 6)()
    from sunscreen_amd import _lib

    rc = _lib.load().hipbfv_debug_f64_plan(C.c_uint64(q), C.c_uint32(logn), out)
    assert rc == 0
    return list(out)


@pytest.mark.parametrize("n", [4096, 8192, 16384])
def test_range_plans_of_the_default_primes_hold_under_an_independent_model(n):
    logn = n.bit_length() - 1
    for q in O.bfv_default(n):
        use, fmask, imask, split, sfmask, simask = _plan(q, logn)
        assert use == 1 and split == 1, (n, q)  # every SEAL-default prime of these degrees runs on the FP64 pipe, split pipelines included
        room = LIMIT / q
        # whole-polynomial transforms, 16 elements per thread
        mod = _Model(q)
        m = 1.0
        for p, r in enumerate(_whole_radices(logn)):
            m = mod.reduce_by_mask(m, fmask, p)
            for _ in range(r):
                m = mod.fwd_stage(m)
        mi = 1.0
        rad = _whole_radices(logn)
        for p in range(len(rad)):
            mi = mod.reduce_by_mask(mi, imask, p)
            for _ in range(rad[len(rad) - 1 - p]):
                mi = mod.inv_stage(mi)
        assert mod.peak < room, (n, q, mod.peak, room)
        # head / middle / tail transforms
        _replay_split_plan(q, logn, sfmask, simask)


PLAN_STORE_REDUCE = 1 << 30  # devctx.hpp kPlanStoreReduce / kPlanScaleReduce
PLAN_SCALE_REDUCE = 1 << 29
PACK_RANGE = float(1 << 47)  # a 48-bit packed row holds |v| < 2^47 (kernels_split.hip nat_store)


def _replay_split_plan(q, logn, sfmask, simask):
    """The split plan of one prime against the independent model: every intermediate below 2^53; rows that travel 48-bit packed
    WITHOUT a reduction in front of the store (flag clear) fit the packed range; the tail's scaling product is canonical after
    one conditional add unless the plan asks for a reduction first.  Rows are modelled as 8-byte doubles (no reduction at a
    store): a packed row that does reduce there only starts the next kernel smaller."""
    room = LIMIT / q
    mod = _Model(q)
    m = 1.0
    for _ in range(_head_log(logn)):
        m = mod.fwd_stage(m)
    packable = q < (1 << 48)  # rows of wider primes are 8-byte doubles in every configuration (context.cpp: pack_ks / pack_mul / per-row flags)
    if packable and not sfmask & PLAN_STORE_REDUCE:
        assert m * q < PACK_RANGE, (q, logn, m)
    if not packable:
        assert not (sfmask | simask) & PLAN_STORE_REDUCE, (q, logn)  # r06: no packed-range reductions planned for rows that never pack
    for p, r in enumerate(_split_fwd_radices(logn)):
        m = mod.reduce_by_mask(m, sfmask, p)
        for _ in range(r):
            m = mod.fwd_stage(m)
    assert (m * q) ** 2 < 2.0 ** 105  # the tensor product multiplies two such values
    # what enters the inverse middle passes: a product a * b of two forward outputs (ArithD::mul_var, bound checked by
    # test_variable_product_bound below) or a freshly reduced accumulator
    mi = max(mod.red(4 * m), 0.5 + m * m * q * 1.5 / TWO52)
    if logn == 15:
        mi = 1.0  # only the stand-alone two-kernel inverse exists at this degree: canonical residues in, no product in split form
    for p, r in enumerate(_split_inv_radices(logn)):
        mi = mod.reduce_by_mask(mi, simask, p)
        for _ in range(r):
            mi = mod.inv_stage(mi)
    if packable and not simask & PLAN_STORE_REDUCE:
        assert mi * q < PACK_RANGE, (q, logn, mi)
    mi = mod.reduce_by_mask(mi, simask, 8)  # bit 8 / 24: at the start of the tail stages
    for _ in range(2):
        mi = mod.inv_stage(mi)
    assert mod.peak < room, (q, logn, mod.peak, room)
    if not simask & PLAN_SCALE_REDUCE:
        # ArithD::mul_const(v, c), c < q: q * (0.5 + |v| * 2^-52); below q (here: 0.97 q) one conditional add makes it canonical
        assert 0.5 + mi * q / TWO52 < 0.97, (q, logn, mi)
    return m, mi


@pytest.mark.parametrize("logn", [12, 13, 14, 15])
def test_split_plans_hold_for_every_prime_size_the_policy_admits(logn):
    """Beyond the SEAL default sets: the library's own auxiliary primes (36 ... 48 bits) and user primes up to the policy's 2^50.
    Whatever the plan decides (skip the store reduction or keep it, where the one reduction of the inverse goes) must hold."""
    skipped_store, kept_store = 0, 0
    for bits in range(36, 51):
        for q in O.get_primes(2 << logn, bits, 2):
            use, _, _, split, sfmask, simask = _plan(q, logn)
            if not (use and split):
                assert bits >= 49, (q, bits)  # every prime below 2^48 keeps the FP64 pipe at every degree
                continue
            _replay_split_plan(q, logn, sfmask, simask)
            if simask & PLAN_STORE_REDUCE:
                kept_store += 1
            else:
                skipped_store += 1
    assert skipped_store > 0 and kept_store > 0  # both arms are exercised


def test_default_headline_primes_need_no_store_or_scale_reduction():
    """n = 8192, SEAL default set + the 42-bit auxiliary base: the reductions in front of packed stores and after the tail's scaling are
    all skipped, and the inverse reduces once, at the start of its last middle pass (the r05 instruction cut)."""
    for q in O.bfv_default(8192) + O.get_primes(2 * 8192, 42, 5):
        use, _, _, split, sfmask, simask = _plan(q, 13)
        assert use and split
        assert sfmask == 0, hex(sfmask)
        assert simask == 1 << 3, hex(simask)


@pytest.mark.parametrize("q", _fp64_primes())
def test_variable_product_bound(q):
    """ArithD::mul_var(a, b) with BOTH operands lazy (|a|, |b| up to a few q): exact, congruent, and within
    q * (0.5 + |a * b| / q * 1.5 * 2^-52) -- the entry bound of the inverse middle passes."""
    rng = random.Random(q ^ 0x5A5A)
    for scale in (1, 3, 8):
        top = min(scale * q, int(2.0 ** 52.4))
        if (top * top) >= (1 << 105):
            continue
        for _ in range(1500):
            a, b = rng.randrange(-top, top + 1), rng.randrange(-top, top + 1)
            t, exact = _mul_var(a, b, q)
            assert exact and (t - a * b) % q == 0
            assert abs(t) <= q * (0.5 + abs(a * b) / q * 1.5 / TWO52) * (1 + 1e-12)


def test_primes_beyond_the_policy_are_refused():
    for q in O.get_primes(2 * 8192, 54, 2) + O.get_primes(2 * 8192, 51, 1):
        assert _plan(q, 13)[0] == 0
    big = _plan(O.get_primes(2 * 16384, 50, 1)[0], 14)
    assert big[0] in (0, 1)  # just below 2^50: whichever the plan says, it says it without a device


def _mul_var(a: int, b: int, q: int):
    """ArithD::mul_var (two variable operands): the same error-free split with the quotient estimated from the rounded product."""
    return _mul_tw(a, b, q)


@pytest.mark.parametrize("data_bits", [54, 56, 61])
def test_mixed_base_sums_take_wide_data_residues_as_two_exact_halves(data_bits):
    """DevCtx::aux_mixed (behzcore.hpp, behz_extend_multi_mixed / behz_floor_sk_coeff_mixed): a data residue y < 2^62 enters a sum
    modulo an FP64-pipe auxiliary prime p < 2^48 as y = yh * 2^30 + yl, the high half against the constant pre-multiplied by
    2^30.  Replayed here with every rounding explicit: each product is exact, congruent, within p * (0.5 + tiny), and a K-term
    sum (2K products + the r_mtilde term) stays far below 2^53."""
    rng = random.Random(data_bits)
    for p in O.get_primes(2 * 8192, 44, 2) + O.get_primes(2 * 8192, 47, 2):
        for _ in range(500):
            ys = [rng.randrange(1 << data_bits) for _ in range(4)]
            cs = [rng.randrange(p) for _ in range(4)]
            acc, exact_sum = 0, 0
            for y, c in zip(ys, cs):
                yh, yl = y >> 30, y & ((1 << 30) - 1)
                assert yh < (1 << 32) and float(yh) == yh and float(yl) == yl
                c_hi = (c << 30) % p
                for a, b in ((yh, c_hi), (yl, c)):
                    t, exact = _mul_var(a, b, p)
                    assert exact and (t - a * b) % p == 0 and abs(t) <= p * 0.5000001
                    acc += t
                exact_sum += y * c
            rc = rng.randrange(-(1 << 31), 1 << 31)
            t, exact = _mul_var(rc, rng.randrange(p), p)
            assert exact
            assert abs(acc) + abs(t) <= 4.5 * p * 1.000001 and 4.5 * p < LIMIT / 8  # nine terms of at most p/2 each: nowhere near 2^53
            assert (acc - exact_sum) % p == 0


def test_lds_placement_is_conflict_free_under_both_lane_group_rules():
    """nttcore.hpp lds_pos (= kernels_split.hip blk_pos) against tools/lds_swizzle_search.py: the map in the header is the one the tool
    holds, and every pass window of every transform shape is conflict free both for 32-lane loads over 32 eight-byte banks and for
    16-lane stores / merged loads over 16 (the LDS model of MI355X_MICROARCH.md; lane-by-lane simulation + the linear criterion)."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import lds_swizzle_search as S

    src = open(os.path.join(root, "sunscreen_amd", "csrc", "nttcore.hpp")).read()
    body = src[src.index("u32 lds_pos(u32 e) {"):]
    body = body[:body.index("}")]
    cols = {int(b): int(c, 16) for b, c in re.findall(r"\(\(e >> (\d+)\) & 1u\) \* (0x[0-9A-Fa-f]+)u", body)}
    assert cols, "could not read the map out of nttcore.hpp"
    for b, c in cols.items():
        assert S.IN_TREE[b] == (c ^ ((1 << b) if b < 5 else 0)), (b, hex(c))
    assert set(cols) == {b for b in S.IN_TREE if S.IN_TREE[b] != ((1 << b) if b < 5 else 0)}
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "lds_swizzle_search.py"), "check"], capture_output=True).returncode == 0
    # the map of rounds 1-4 passes the 32-lane rule and fails the 16-lane one: the measurement that prompted the new search
    old = {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 0b00101, 6: 0b01010, 7: 0b10001}
    cons = S.constraints()
    split = [c for c in cons]
    assert not S.satisfies(old, split)
    got, floor = S.simulate(old, ("split", 13, 8))
    assert got > floor
