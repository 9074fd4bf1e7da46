#!/usr/bin/env python3
"""Derive compact golden vectors from the reference's binary SEAL key fixtures.

Input  (read-only, only present in the build container):
    /root/reference/seal_fhe/tests/data/secret_key.bin   (SEAL 4.0 SecretKey, zstd)
    /root/reference/seal_fhe/tests/data/public_key.bin   (SEAL 4.0 PublicKey, zstd)
  loaded by the reference's `deterministic` test seal_fhe/src/encryptor_decryptor.rs:886-933 with
  parameters n=8192, CoeffModulus::create(8192,[50,30,30,50,50]), PlainModulus::batching(8192,20)
  (the key material does not depend on the plain modulus; only the parms_id in the header does).

Output: tests/golden/seal_key_fixture.npz  (a few KB) holding
    primes        the five key-level primes
    sk_ternary    int8[8192]  INTT(sk_j), identical for all five primes (checked here)
    pk_err        int8[8192]  INTT(pk0 + pk1 (.) sk), identical for all five primes (checked here)
    sk_sha256     sha256 of each NTT-form residue polynomial u64[8192] (little endian) of the secret key
    pk_sha256     sha256 of each of the 2x5 residue polynomials of the public key
    sk_head/pk_head  first 8 words of every residue polynomial (for debugging a mismatch)

The fixture therefore pins, bit for bit, SEAL's choice of psi (minimal primitive 2n-th root),
the bit-reversed output order of the forward transform and the [poly][rns][coeff] layout:
tests/test_oracle_golden.py recomputes NTT(sk_ternary) with the oracle and compares hashes.

The NTT used to *decode* the fixture here is written in plain numpy/python ints independently of
oracle/ so the golden file does not depend on the code it is meant to check.
"""
import ctypes
import hashlib
import os
import struct
import sys

import numpy as np

REF = "/root/reference/seal_fhe/tests/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seal_key_fixture.npz")
N = 8192
PRIMES = [1125899905744897, 1073643521, 1073692673, 1125899906629633, 1125899906826241]


def zstd_decompress(buf: bytes) -> bytes:
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_getFrameContentSize.restype = ctypes.c_ulonglong
    z.ZSTD_getFrameContentSize.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    size = z.ZSTD_getFrameContentSize(buf, len(buf))
    if size in (2**64 - 1, 2**64 - 2):
        size = 64 << 20
    out = ctypes.create_string_buffer(size)
    got = z.ZSTD_decompress(out, size, buf, len(buf))
    return out.raw[:got]


def load_seal_object(path: str) -> bytes:
    raw = open(path, "rb").read()
    magic, hdr, major, minor, compr, _res, total = struct.unpack_from("<HBBBBHQ", raw, 0)
    assert magic == 0xA15E and hdr == 16 and (major, minor) == (4, 0), (hex(magic), hdr, major, minor)
    assert total == len(raw)
    body = raw[16:]
    if compr == 2:
        body = zstd_decompress(body)
    elif compr != 0:
        raise ValueError("unsupported compression %d" % compr)
    return body


def find_payload(body: bytes, words: int) -> np.ndarray:
    """The u64 payload is the trailing `words` 8-byte words of the object body."""
    arr = np.frombuffer(body[len(body) - 8 * words :], dtype="<u8")
    return arr.copy()


def min_root(q: int, two_n: int) -> int:
    e = (q - 1) // two_n
    g = 2
    while True:
        c = pow(g, e, q)
        if pow(c, two_n // 2, q) == q - 1:
            break
        g += 1
    sq = c * c % q
    best = cur = c
    for _ in range(two_n // 2):
        best = min(best, cur)
        cur = cur * sq % q
    return best


def bitrev(v: int, bits: int) -> int:
    return int(format(v, "0%db" % bits)[::-1], 2)


def intt(x: np.ndarray, q: int) -> list:
    """Inverse of: CT forward, natural in / bit-reversed out, twiddle psi^bitrev(m+i)."""
    n = len(x)
    logn = n.bit_length() - 1
    psi = min_root(q, 2 * n)
    ipsi = pow(psi, -1, q)
    ipow = [1] * n
    for i in range(1, n):
        ipow[i] = ipow[i - 1] * ipsi % q
    a = [int(v) for v in x]
    gap, m = 1, n // 2
    while m >= 1:
        for i in range(m):
            w = ipow[bitrev(m + i, logn)]
            base = 2 * i * gap
            for j in range(base, base + gap):
                u, v = a[j], a[j + gap]
                a[j] = (u + v) % q
                a[j + gap] = (u - v) * w % q
        m //= 2
        gap *= 2
    ninv = pow(n, -1, q)
    return [v * ninv % q for v in a]


def centred(v: list, q: int) -> np.ndarray:
    return np.array([x - q if x > q // 2 else x for x in v], dtype=np.int64)


def main() -> int:
    sk_body = load_seal_object(os.path.join(REF, "secret_key.bin"))
    pk_body = load_seal_object(os.path.join(REF, "public_key.bin"))
    sk = find_payload(sk_body, 5 * N).reshape(5, N)
    pk = find_payload(pk_body, 2 * 5 * N).reshape(2, 5, N)
    sk_coeff = []
    err = []
    for j, q in enumerate(PRIMES):
        assert int(sk[j].max()) < q and int(pk[:, j].max()) < q
        s = centred(intt(sk[j], q), q)
        assert np.abs(s).max() <= 1, "secret key is not ternary under this NTT convention"
        sk_coeff.append(s)
        d = [(int(a) + int(b) * int(c)) % q for a, b, c in zip(pk[0, j], pk[1, j], sk[j])]
        e = centred(intt(np.array(d, dtype=object), q), q)
        assert np.abs(e).max() <= 41, "public key error is not small"
        err.append(e)
        print("prime %d ok: |e|max=%d" % (j, np.abs(e).max()))
    for j in range(1, 5):
        assert (sk_coeff[j] == sk_coeff[0]).all() and (err[j] == err[0]).all()
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()
    np.savez_compressed(
        OUT,
        primes=np.array(PRIMES, dtype=np.uint64),
        sk_ternary=sk_coeff[0].astype(np.int8),
        pk_err=err[0].astype(np.int8),
        sk_sha256=np.array([sha(sk[j]) for j in range(5)]),
        pk_sha256=np.array([[sha(pk[p, j]) for j in range(5)] for p in range(2)]),
        sk_head=sk[:, :8].copy(),
        pk_head=pk[:, :, :8].copy(),
    )
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
