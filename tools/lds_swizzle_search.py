#!/usr/bin/env python3
"""tools/lds_swizzle_search.py [check] -- the LDS placement of the NTT kernels (nttcore.hpp lds_pos = kernels_split.hip blk_pos).

Element e of a polynomial (or of a middle-kernel block) is stored at position (e & ~31) | A(e), A a GF(2)-linear map of the index
bits into the five bank bits that is a bijection on every 32-word block.  A pass over the index window [LOW, LOW + R) executed by
virtual thread vt = tid + g * T touches e = (hi << (LOW + R)) | (k << LOW) | lo with lo = vt & (2^LOW - 1), hi = vt >> LOW: lane bit
i of a wavefront lands on index bit i (i < LOW) or i + R.

LDS model (MI355X_MICROARCH.md, "LDS [CDNA4]"): 8-byte accesses,
  ds_read_b64                        2 groups of 32 lanes, 32 eight-byte banks: lane bits 0..4 -> 5 independent vectors of GF(2)^5;
  ds_write_b64, ds_read2*, ds_write2* 4 groups of 16 CONTIGUOUS lanes, 16 eight-byte banks: lane bits 0..3 -> 4 independent vectors
                                     after dropping bank bit 4.
Rounds 1-4 searched under the first rule only; the maps they found cost 25-33 % extra LDS cycles on every store whose 16 lanes reach
index bit 4 (measured in round 5: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.27 in mul_mid<13>, 0.29 in ks_mid<13>).  Every window is
required to pass BOTH rules (a pass stores with its own window and the next pass loads with the next one; the compiler is free to merge
loads into the read2 forms).

No argument: search (columns of index bits 4 ... 8; bits 0..3 stay unit vectors), print the first map and its simulated cycles.
`check`: verify the map that is in nttcore.hpp (linear criterion + a lane-by-lane simulation); exit status 1 if it has a conflict.
"""
import itertools
import sys

IN_TREE = {0: 1, 1: 2, 2: 4, 3: 8, 4: 0b10101, 5: 0b01110, 6: 0b01001, 7: 0b11000, 8: 0b10000}  # nttcore.hpp lds_pos


def ilog2(v):
    return v.bit_length() - 1


def elem_index(low, r, vt, k):
    lo, hi = vt & ((1 << low) - 1), vt >> low
    return (hi << (low + r)) | (k << low) | lo


def split_windows(L):
    """pass windows of the middle kernels (nttshape.hpp): forward after the head's stages, inverse before the tail's two"""
    head = 2 if L == 14 else 3
    fwd = {9: [3, 3, 3], 10: [3, 3, 2, 2], 11: [3, 3, 3, 2], 12: [3, 3, 3, 3]}[L - head]
    inv = {10: [3, 3, 2, 2], 11: [2, 3, 3, 3], 12: [3, 3, 3, 3], 13: [3, 3, 3, 2, 2]}[L - 2]
    out, s = [], head
    for r in fwd:
        s += r
        out.append((L - s, r))
    low = 0
    for r in inv:
        out.append((low, r))
        low += r
    return out


def whole_windows(logn, ept):
    """pass windows of the whole-polynomial transforms (nttshape.hpp ntt_pass_radix) + the linear read-out"""
    npass = (logn + ilog2(ept) - 1) // ilog2(ept)
    out, s0 = [], 0
    for p in range(npass):
        r = logn // npass + (1 if p < logn % npass else 0)
        out.append((logn - s0 - r, r))
        s0 += r
    return out + [(24, 0)]


SHAPES = [("split", L, e) for L, e in ((12, 8), (13, 8), (14, 8), (14, 16), (15, 8))] + \
         [("whole", n, e) for n in (10, 11, 12, 13, 14) for e in (16, 8) if not (n == 14 and e == 8)]


def windows(shape):
    kind, L, ept = shape
    return split_windows(L) if kind == "split" else whole_windows(L, ept)


def constraints():
    cons = set()
    for sh in SHAPES:
        for low, r in windows(sh):
            cons.add((5, tuple(b if b < low else b + r for b in range(5))))  # 32 lanes, 32 banks
            cons.add((4, tuple(b if b < low else b + r for b in range(4))))  # 16 lanes, 16 banks
    return sorted(cons)


def rank(vecs):
    basis = []
    for v in vecs:
        for b in basis:
            v = min(v, v ^ b)
        if v:
            basis.append(v)
    return len(basis)


def satisfies(cols, cons):
    for nbits, bits in cons:
        vecs = [cols.get(b, 0) for b in bits]
        if nbits == 4:
            vecs = [v & 15 for v in vecs]
        if rank(vecs) != nbits:
            return False
    return True


def place(cols, e):
    m = 0
    for b, c in cols.items():
        if (e >> b) & 1:
            m ^= c
    return (e & ~31) | m


def simulate(cols, shape):
    """LDS-array cycles of every store (16-lane groups) and load (32-lane groups) of one transform, lane by lane: (cycles, minimum)"""
    kind, L, ept = shape
    threads = ((1 << (L - 2)) if kind == "split" else (1 << L)) // ept
    got = floor = 0
    for low, r in windows(shape):
        if r == 0:
            continue
        for g in range(ept >> r):
            for k in range(1 << r):
                for wave in range(max(threads // 64, 1)):
                    for group, banks in ((16, 16), (32, 32)):
                        for g0 in range(0, 64, group):
                            seen = {}
                            for lane in range(g0, g0 + group):
                                p = place(cols, elem_index(low, r, wave * 64 + lane + g * threads, k))
                                seen.setdefault(p % banks, set()).add(p)
                            got += max(len(v) for v in seen.values())
                            floor += 1
    return got, floor


def main():
    cons = constraints()
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        ok = satisfies(IN_TREE, cons) and rank([IN_TREE[b] for b in range(5)]) == 5
        for sh in SHAPES:
            got, floor = simulate(IN_TREE, sh)
            print(sh, "cycles", got, "minimum", floor)
            ok = ok and got == floor
        print("conflict free under both rules" if ok else "CONFLICTS")
        return 0 if ok else 1
    high = sorted({b for _, bits in cons for b in bits if b >= 5})
    print("index bits above 4 that reach a bank:", high, "--", len(cons), "constraints")
    for c4 in range(16):
        for choice in itertools.product(range(32), repeat=len(high)):
            cols = {0: 1, 1: 2, 2: 4, 3: 8, 4: 16 | c4}
            cols.update(zip(high, choice))
            if satisfies(cols, cons):
                print("map:", {b: format(c, "05b") for b, c in cols.items()})
                for sh in SHAPES:
                    print(sh, simulate(cols, sh))
                return 0
    print("no map with unit columns for bits 0..3")
    return 1


if __name__ == "__main__":
    sys.exit(main())
