"""Mutation fuzzing of everything that parses untrusted bytes on the host: the SEAL wire decoders (ciphertext, plaintext, both
compression modes) and the serde-JSON program loader.  Thousands of mutated inputs per target -- bit flips, byte splices,
truncations, length-field edits, valid fragments glued together -- go through the C ABI in a child process: every call must come
back with an HRESULT (accept or reject), the child must exit normally.  A crash, an abort (an exception crossing extern "C") or
a hang fails the test.  (untrusted bytes drove allocations and unwound through the C ABI.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, json, struct, sys
import numpy as np
sys.path.insert(0, %r)
from sunscreen_amd import _lib
from sunscreen_amd.workloads import chi_sq_optimized, dot_product
L = _lib.load()
rng = np.random.default_rng(int(sys.argv[1]))
used = C.c_int64()

def hdr(total, compr=0):
    return bytes([0x5E, 0xA1, 16, 4, 0, compr, 0, 0]) + struct.pack("<Q", total)

def ct_blob(n, k, size, compr):
    data = rng.integers(0, 1 << 40, size * k * n, dtype=np.uint64)
    written = C.c_int64()
    pid = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    assert L.hipbfv_wire_encode_ciphertext(pid, True, size, n, k, data.ctypes.data_as(_lib.u64p), compr, None, 0, C.byref(written)) == 0
    buf = C.create_string_buffer(written.value)
    assert L.hipbfv_wire_encode_ciphertext(pid, True, size, n, k, data.ctypes.data_as(_lib.u64p), compr, buf, written.value, C.byref(written)) == 0
    return buf.raw[: written.value]

def pt_blob(count):
    body = bytes(32) + struct.pack("<Qd", count, 1.0) + hdr(16 + 8 + 8 * count) + struct.pack("<Q", count) + rng.integers(0, 1 << 17, count, dtype=np.uint64).tobytes()
    return hdr(16 + len(body)) + body

def mutate(b):
    b = bytearray(b)
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 6))
        if not b:
            break
        if kind == 0:    # bit flip, biased towards the headers
            i = int(rng.integers(0, min(len(b), 96))) if rng.random() < 0.7 else int(rng.integers(0, len(b)))
            b[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:  # overwrite 8 bytes with an interesting length
            i = int(rng.integers(0, max(1, len(b) - 8)))
            v = [0, 1, 0xFF, 1 << 31, (1 << 63) - 1, (1 << 64) - 1, len(b), len(b) * 8][int(rng.integers(0, 8))]
            b[i:i + 8] = struct.pack("<Q", v)
        elif kind == 2:  # truncate
            del b[int(rng.integers(0, len(b))):]
        elif kind == 3:  # duplicate a slice
            i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
            b[i:i] = b[i:j][:4096]
        elif kind == 4:  # random bytes
            i = int(rng.integers(0, len(b)))
            b[i:i + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
        else:            # drop a slice
            i, j = sorted(int(x) for x in rng.integers(0, len(b), 2))
            del b[i:j]
    return bytes(b)

seeds_ct = [ct_blob(1024, 2, 2, 0), ct_blob(1024, 3, 3, 0), ct_blob(2048, 1, 2, 2), ct_blob(1024, 2, 2, 2)]
seeds_pt = [pt_blob(1024), pt_blob(7)]
out = np.zeros(1 << 16, dtype=np.uint64)
pid = C.create_string_buffer(32)
ntt, size, n, k, cnt = C.c_bool(), C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
calls = accepted = 0
for it in range(int(sys.argv[2])):
    m = mutate(seeds_ct[it %% len(seeds_ct)])
    hr = L.hipbfv_wire_decode_ciphertext(m, len(m), pid, C.byref(ntt), C.byref(size), C.byref(n), C.byref(k), out.ctypes.data_as(_lib.u64p), out.size, C.byref(used))
    accepted += hr == 0
    hr = L.hipbfv_wire_decode_ciphertext(m, len(m), None, None, None, None, None, None, 0, C.byref(used))
    m = mutate(seeds_pt[it %% len(seeds_pt)])
    hr = L.hipbfv_wire_decode_plaintext(m, len(m), pid, C.byref(cnt), out.ctypes.data_as(_lib.u64p), out.size, C.byref(used))
    accepted += hr == 0
    calls += 3
# the JSON program loader
progs = [chi_sq_optimized().to_json().encode(), dot_product(8).to_json().encode()]
for it in range(int(sys.argv[2]) // 2):
    m = mutate(progs[it %% 2])
    h = C.c_void_p()
    assert L.hipbfv_Program_Create(C.byref(h)) == 0
    hr = L.hipbfv_Program_LoadJson(h, m, len(m))
    accepted += hr == 0
    calls += 1
    assert L.hipbfv_Program_Destroy(h) == 0
print("fuzz ok", calls, accepted)
""" % ROOT


def test_mutated_wire_objects_and_program_json_never_crash_the_decoders():
    for seed in (1, 2):
        out = subprocess.run([sys.executable, "-c", CHILD, str(seed), "4000"], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "fuzz ok" in out.stdout, (seed, out.returncode, out.stdout[-400:], out.stderr[-2000:])


def test_parsers_are_clean_under_address_and_ub_sanitizers(tmp_path):
    """The same mutation campaign against the parsers' internal entry points (wire_unpack_*, Program::load_json) in a build with
    AddressSanitizer + UBSan + leak detection: out-of-bounds reads that do not crash a normal build, signed overflows and leaks
    abort this one (tests/native/parser_fuzz_asan.cpp; 300 K inputs by hand were clean)."""
    import shutil

    if not shutil.which("g++") or not os.path.isdir("/opt/rocm/include"):
        import pytest

        pytest.skip("needs g++ and the HIP headers")
    sys.path.insert(0, ROOT)
    from sunscreen_amd.workloads import chi_sq_optimized, dot_product

    seeds = []
    for name, prog in (("chi.json", chi_sq_optimized()), ("dot.json", dot_product(8))):
        p = tmp_path / name
        p.write_text(prog.to_json())
        seeds.append(str(p))
    exe = str(tmp_path / "parser_fuzz_asan")
    csrc = os.path.join(ROOT, "sunscreen_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I", csrc, "-x", "c++", os.path.join(ROOT, "tests", "native", "parser_fuzz_asan.cpp"),
                           os.path.join(csrc, "wire.cpp"), os.path.join(csrc, "program.cpp"), os.path.join(csrc, "program_plan.cpp"), "-ldl", "-Wl,--unresolved-symbols=ignore-all", "-o", exe],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe, "7", "4000"] + seeds, capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0 and "asan fuzz ok" in out.stdout, (out.returncode, out.stdout[-300:], out.stderr[-3000:])
