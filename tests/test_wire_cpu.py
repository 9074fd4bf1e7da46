"""SEAL 4.0 wire format (sunscreen_amd/csrc/wire.cpp) against the reference's own binary fixtures
seal_fhe/tests/data/{public,secret}_key.bin.  Host-only entry points: no GPU needed.

The fixtures cannot travel to the GPU box, so the fixture-dependent checks skip when /root/reference is
absent; the parms_id known answer (taken from the fixture header) is kept as a literal."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from oracle import bfv_oracle as O
from sunscreen_amd import _lib

REF = "/root/reference/seal_fhe/tests/data"
PRIMES = [1125899905744897, 1073643521, 1073692673, 1125899906629633, 1125899906826241]
FIXTURE_PARMS_ID = "128ba7b692add0263494d111647d7d667962562704e7ad1802f32a4406aae084"


def parms_id(n, primes, t):
    out = C.create_string_buffer(32)
    arr = (C.c_uint64 * len(primes))(*primes)
    assert _lib.load().hipbfv_wire_parms_id(n, arr, len(primes), t, out) == 0
    return out.raw


def decode_ct(blob):
    L = _lib.load()
    pid = C.create_string_buffer(32)
    ntt, size, n, k, used = C.c_bool(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int64()
    assert L.hipbfv_wire_decode_ciphertext(blob, len(blob), pid, C.byref(ntt), C.byref(size), C.byref(n), C.byref(k), None, 0, C.byref(used)) == 0
    data = np.zeros(size.value * n.value * k.value, dtype=np.uint64)
    assert L.hipbfv_wire_decode_ciphertext(blob, len(blob), pid, C.byref(ntt), C.byref(size), C.byref(n), C.byref(k),
                                           data.ctypes.data_as(_lib.u64p), data.size, C.byref(used)) == 0
    return pid.raw, ntt.value, size.value, n.value, k.value, data, used.value


def encode_ct(pid, ntt, size, n, k, data, compr):
    L = _lib.load()
    written = C.c_int64()
    assert L.hipbfv_wire_encode_ciphertext(pid, ntt, size, n, k, data.ctypes.data_as(_lib.u64p), compr, None, 0, C.byref(written)) == 0
    buf = C.create_string_buffer(written.value)
    assert L.hipbfv_wire_encode_ciphertext(pid, ntt, size, n, k, data.ctypes.data_as(_lib.u64p), compr, buf, written.value, C.byref(written)) == 0
    return buf.raw[: written.value]


def test_parms_id_known_answer_from_the_fixture():
    # key-level parameters of the `deterministic` test (seal_fhe/src/encryptor_decryptor.rs:696-709,889)
    t = O.plain_batching(8192, 20)
    assert parms_id(8192, PRIMES, t).hex() == FIXTURE_PARMS_ID
    # BLAKE2b-256 cross-check against hashlib on other parameter sets
    import hashlib

    for n, primes, tt in [(4096, O.bfv_default(4096), 262144), (16384, O.bfv_default(16384), 65537), (1024, [132120577], 12289)]:
        ref = hashlib.blake2b(struct.pack("<%dQ" % (3 + len(primes)), 1, n, *primes, tt), digest_size=32).digest()
        assert parms_id(n, primes, tt) == ref


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "public_key.bin")), reason="reference tree absent")
def test_decode_reference_public_key_fixture_and_reencode():
    blob = open(os.path.join(REF, "public_key.bin"), "rb").read()
    pid, ntt, size, n, k, data, used = decode_ct(blob)
    assert used == len(blob)
    assert pid.hex() == FIXTURE_PARMS_ID and ntt and (size, n, k) == (2, 8192, 5)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seal_key_fixture.npz"))
    assert (data.reshape(2, 5, 8192)[:, :, :8] == g["pk_head"]).all()
    # uncompressed re-encoding == header + zstd-decompressed body of the fixture
    import importlib.util

    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(__file__), "golden", "make_seal_fixture_vectors.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    body = mk.load_seal_object(os.path.join(REF, "public_key.bin"))
    plain = encode_ct(pid, ntt, size, n, k, data, 0)
    assert plain[16:] == body
    assert plain[:8] == bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) and struct.unpack("<Q", plain[8:16])[0] == len(plain)
    # zstd re-encoding decodes to the same object
    z = encode_ct(pid, ntt, size, n, k, data, 2)
    assert z[5] == 2 and len(z) < len(plain)
    pid2, ntt2, size2, n2, k2, data2, _ = decode_ct(z)
    assert (pid2, ntt2, size2, n2, k2) == (pid, ntt, size, n, k) and (data2 == data).all()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "secret_key.bin")), reason="reference tree absent")
def test_decode_reference_secret_key_fixture():
    """SecretKey is serialised as a Plaintext (parms_id, coeff_count, scale, DynArray)."""
    blob = open(os.path.join(REF, "secret_key.bin"), "rb").read()
    L = _lib.load()
    pid = C.create_string_buffer(32)
    count, used = C.c_uint64(), C.c_int64()
    assert L.hipbfv_wire_decode_plaintext(blob, len(blob), pid, C.byref(count), None, 0, C.byref(used)) == 0
    assert used.value == len(blob) and count.value == 5 * 8192 and pid.raw.hex() == FIXTURE_PARMS_ID
    data = np.zeros(count.value, dtype=np.uint64)
    assert L.hipbfv_wire_decode_plaintext(blob, len(blob), pid, C.byref(count), data.ctypes.data_as(_lib.u64p), data.size, C.byref(used)) == 0
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seal_key_fixture.npz"))
    assert (data.reshape(5, 8192)[:, :8] == g["sk_head"]).all()


def test_malformed_objects_are_rejected():
    L = _lib.load()
    used = C.c_int64()
    for blob in (b"", b"\x00" * 16, bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) + struct.pack("<Q", 9999), bytes([0x5E, 0xA1, 16, 4, 0, 7, 0, 0]) + struct.pack("<Q", 16)):
        hr = L.hipbfv_wire_decode_ciphertext(blob, len(blob), None, None, None, None, None, None, 0, C.byref(used))
        assert hr != 0
    # truncated valid object
    data = np.arange(2 * 3 * 1024, dtype=np.uint64)
    good = encode_ct(b"\x01" * 32, False, 2, 1024, 3, data, 0)
    assert L.hipbfv_wire_decode_ciphertext(good[:-8], len(good) - 8, None, None, None, None, None, None, 0, C.byref(used)) != 0
    assert L.hipbfv_wire_decode_ciphertext(good, len(good), None, None, None, None, None, None, 0, C.byref(used)) == 0


def test_untrusted_bytes_cannot_drive_allocations_or_unwind_through_the_c_abi():
    """a tiny zstd frame may declare gigabytes of content; a huge element count may sit in a header.  The
    decoders cap what they allocate by what the caller can hold and every exported function is an exception barrier:
    such inputs come back as an HRESULT, never as a C++ exception crossing extern "C" (which would abort the process)."""
    import ctypes.util

    L = _lib.load()
    used = C.c_int64()
    z = C.CDLL(ctypes.util.find_library("zstd") or "libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    raw = bytes(1 << 26)  # 64 MiB of zeros -> a frame of a few KB
    dst = C.create_string_buffer(z.ZSTD_compressBound(len(raw)))
    got = z.ZSTD_compress(dst, len(dst), raw, len(raw), 3)
    frame = dst.raw[:got]
    assert len(frame) < 1 << 16
    bomb = bytes([0x5E, 0xA1, 16, 4, 0, 2, 0, 0]) + struct.pack("<Q", 16 + len(frame)) + frame
    small = np.zeros(1024, dtype=np.uint64)
    pid = C.create_string_buffer(32)
    ntt, size, n, k = C.c_bool(), C.c_uint64(), C.c_uint64(), C.c_uint64()
    # with a destination buffer the decoder refuses to inflate beyond what that buffer could hold
    hr = L.hipbfv_wire_decode_ciphertext(bomb, len(bomb), pid, C.byref(ntt), C.byref(size), C.byref(n), C.byref(k),
                                         small.ctypes.data_as(_lib.u64p), small.size, C.byref(used))
    assert hr == 0x80131620 - (1 << 32) or hr == 0x80131620  # COR_E_IO
    # an element count far beyond the bytes present
    body = b"\x01" * 32 + b"\x00" + struct.pack("<QQQdQ", 2, 1 << 19, 60, 1.0, 1)
    body += bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) + struct.pack("<QQ", 24, 1 << 40)
    obj = bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) + struct.pack("<Q", 16 + len(body)) + body
    assert L.hipbfv_wire_decode_ciphertext(obj, len(obj), None, None, None, None, None, None, 0, C.byref(used)) != 0
    # JSON: nesting beyond the parser's depth limit, and a number that ends exactly at the end of an unterminated buffer
    prog = C.c_void_p()
    assert L.hipbfv_Program_Create(C.byref(prog)) == 0
    deep = b"[" * 200000
    assert L.hipbfv_Program_LoadJson(prog, deep, len(deep)) != 0
    tail = b'{"graph": {"nodes": [], "edges": [], "x": 12345'
    buf = (C.c_char * len(tail)).from_buffer_copy(tail)  # no NUL after the digits
    assert L.hipbfv_Program_LoadJson(prog, C.cast(buf, C.c_char_p), len(tail)) != 0
    assert L.hipbfv_Program_Destroy(prog) == 0
    # hex polynomial with an absurd exponent: refused before anything is sized by it
    out = C.c_void_p()
    assert L.Plaintext_Create4(b"1x^99999999999", None, C.byref(out)) != 0


def test_seed_compressed_objects_are_refused_with_their_own_message():
    """A SEAL client may save keys / symmetric ciphertexts with save_seed: Ciphertext::save_members then writes the first
    polynomial only, followed by a UniformRandomGeneratorInfo object (header, prng type, 64-byte seed)
    (/root/reference/seal_fhe/src/key_generator.rs:89-158 create_compact_*).  Expanding that seed needs SEAL's own PRNG stream,
    which nothing here can be pinned against: the decoder says so instead of failing as "malformed"."""
    L = _lib.load()
    used = C.c_int64()
    n, k = 1024, 2
    hdr = lambda total: bytes([0x5E, 0xA1, 16, 4, 0, 0, 0, 0]) + struct.pack("<Q", total)
    half = np.arange(n * k, dtype=np.uint64).tobytes()
    dyn = hdr(16 + 8 + len(half)) + struct.pack("<Q", n * k) + half
    info = hdr(16 + 1 + 64) + bytes([1]) + bytes(64)  # blake2xb, all-zero seed
    body = b"\x07" * 32 + b"\x01" + struct.pack("<QQQdQ", 2, n, k, 1.0, 1) + dyn + info
    obj = hdr(16 + len(body)) + body
    hr = L.hipbfv_wire_decode_ciphertext(obj, len(obj), None, None, None, None, None, None, 0, C.byref(used))
    assert hr & 0xFFFFFFFF == 0x80131620
    msg = _lib.last_error()
    assert "seed-compressed" in msg
    # the same bytes without the trailing seed object are just a truncated ciphertext
    obj2 = hdr(16 + len(body) - len(info)) + body[: -len(info)]
    assert L.hipbfv_wire_decode_ciphertext(obj2, len(obj2), None, None, None, None, None, None, 0, C.byref(used)) != 0
    assert "seed-compressed" not in _lib.last_error()
