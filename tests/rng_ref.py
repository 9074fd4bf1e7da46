"""Test-side restatement (numpy) of sunscreen_amd/csrc/rng.hpp: ChaCha20 block function (RFC 8439) evaluated at a position,
and the seed derivation.  Used to check that the DEVICE generator is the documented one (tests/test_rng.py)."""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def _rotl(v, c):
    return ((v << np.uint64(c)) | (v >> np.uint64(32 - c))) & M32


def chacha20_block(key8, i0, i1, i2, i3):
    """key8: 8 words; i0..i3: scalars or equal-shape arrays.  Returns 16 arrays of words (as uint64 holding 32-bit values)."""
    shape = np.broadcast(np.asarray(i0), np.asarray(i1), np.asarray(i2), np.asarray(i3)).shape
    const = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    init = [np.full(shape, c, dtype=np.uint64) for c in const] + [np.full(shape, int(k), dtype=np.uint64) for k in key8]
    init += [np.broadcast_to(np.asarray(v, dtype=np.uint64), shape).copy() for v in (i0, i1, i2, i3)]
    x = [v.copy() for v in init]

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[j] + init[j]) & M32 for j in range(16)]


def seed_from_512(seed64: bytes):
    w = np.frombuffer(seed64, dtype="<u4")
    lo, hi = w[:8], w[8:]
    keys = []
    for dom in (1, 2):
        a = chacha20_block(lo, dom, 0x68697062, 0x66762D6B, 0x64663031)
        b = chacha20_block(hi, dom, 0x68697062, 0x66762D6B, 0x64663031)
        keys.append([int(a[j]) ^ int(b[8 + j]) for j in range(8)])
    return {"secret": keys[0], "pub": keys[1]}


def seed_from_u64_for_tests(seed: int):
    k = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, 0x74657374, 0, 0, 0, 0, 0]
    w = chacha20_block(k, 0, 0, 0, 0)
    return seed_from_512(np.array([int(v) for v in w], dtype="<u4").tobytes())
