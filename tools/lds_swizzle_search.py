#!/usr/bin/env python3
"""Search a GF(2)-linear LDS bank swizzle for the block-local (split) NTT kernels.

Element e of a block is stored at e ^ m(e) where m is a linear function of bits 5.. of e taking values in
the five bank bits; an access pattern is conflict free for a 32-lane half-wave iff the five varying index
bits map to five independent bank vectors.  Patterns: a pass over the window [LOW, LOW+R) executed with
virtual thread vt = tid + g*TPB touches e = (hi << (LOW+R)) | (k << LOW) | lo, lo = vt & (2^LOW-1), hi = vt >> LOW.
"""
import itertools
import sys


def varying_bits(low, r):
    # the 5 low bits of vt map to: lo bits e0..e(low-1), then hi bits e(low+r)...
    bits = []
    for i in range(5):
        bits.append(i if i < low else i + r)
    return bits


def schedules(L):
    lf, li = L - 3, L - 2  # stages done in the middle kernel (forward after a radix-8 head, inverse before a radix-4 tail)
    fwd = {10: [3, 3, 2, 2], 9: [3, 3, 3], 11: [2, 3, 3, 3], 12: [3, 3, 3, 3], 8: [3, 3, 2], 7: [3, 2, 2]}[lf]
    inv = {11: [2, 3, 3, 3], 10: [3, 3, 2, 2], 12: [3, 3, 3, 3], 13: [3, 3, 3, 2, 2], 9: [2, 3, 2, 2], 8: [2, 2, 2, 2]}[li]
    pats = []
    s0 = 3
    for r in fwd:
        low = L - s0 - r
        pats.append((low, r))
        s0 += r
    assert s0 == L
    low = 0
    assert inv[0] == fwd[-1]
    for r in inv:
        pats.append((low, r))
        low += r
    assert low == li
    return fwd, inv, pats


def rank5(vecs):
    basis = []
    for v in vecs:
        for b in basis:
            v = min(v, v ^ b)
        if v:
            basis.append(v)
    return len(basis)


def ok(cols, pats):
    for low, r in pats:
        vecs = []
        for b in varying_bits(low, r):
            vecs.append((1 << b) if b < 5 else cols.get(b, 0))
        if rank5(vecs) != 5:
            return False
    return True


def main():
    Ls = [int(a) for a in sys.argv[1:]] or [12, 13, 14]
    pats = []
    for L in Ls:
        f, i, p = schedules(L)
        print("L", L, "fwd", f, "inv", i, "windows", p)
        pats += p
    pats.append((16, 0))  # linear copy: bits 0..4 vary
    hi_bits = sorted({b for low, r in pats for b in varying_bits(low, r) if b >= 5})
    print("high bits involved:", hi_bits)
    for choice in itertools.product(range(32), repeat=len(hi_bits)):
        cols = dict(zip(hi_bits, choice))
        if ok(cols, pats):
            print("solution:", {b: format(c, "05b") for b, c in cols.items()})
            return
    print("no linear solution")


if __name__ == "__main__":
    main()
