"""include/hipbfv.hpp -- the compiled-language host mirror of the `seal_fhe` crate's surface (the reference's host side is
Rust; no Rust toolchain exists here) -- builds against the C ABI and behaves like the crate: examples/simple_multiply.cpp is
the reference's examples/simple_multiply (15 * 5 = 75, examples/simple_multiply/src/main.rs:57-80) plus the rotation / wire
format / transparent-ciphertext checks of seal_fhe/src/bfv_evaluator.rs:322-970, written against that header.

The device half (`./simple_multiply` without arguments) is part of the `-m gpu` suite, built at -O0 and at -O2 -- the
compiled-language consumer a Rust crate linking libhipbfv.so resembles most.  One of two round-1 runs of it failed on
the GPU box (commit 6df2a2d): root cause and regression test in tests/native/interpose_check.cpp (symbol interposition
of the library's then-exported C++ internals; the library now exports the C ABI only)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LIB = os.path.join(ROOT, "sunscreen_amd", "lib")


def _build(tmp_path, opt="-O0", src=os.path.join(ROOT, "examples", "simple_multiply.cpp"), extra=()):
    exe = str(tmp_path / (os.path.basename(src)[:-4] + opt))
    subprocess.check_call(["g++", "-std=c++17", opt, "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", LIB, "-lhipbfv",
                           "-Wl,-rpath," + LIB, "-o", exe] + list(extra))
    return exe


def test_cpp_mirror_builds_and_its_host_side_behaves_like_the_crate(tmp_path):
    """No GPU: parameter objects, prime-generation known answers (modulus.rs:279-313), builder errors, HRESULT -> Error
    mapping (error.rs:65-78), deep copies / moves, Plaintext from a polynomial string."""
    out = subprocess.run([_build(tmp_path), "--host-only"], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "host-only ok" in out.stdout


def test_library_exports_the_c_abi_only():
    """Nothing but include/hipbfv.h's functions is in the dynamic symbol table: no C++ symbol of the library can be bound
    to (or replaced by) a consumer's."""
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIB, "libhipbfv.so")], capture_output=True, text=True, check=True).stdout
    names = [l.split()[-1] for l in out.splitlines() if l.strip()]
    header = open(os.path.join(ROOT, "include", "hipbfv.h")).read()
    declared = set(re.findall(r"^long\s+([A-Za-z_0-9]+)\s*\(", header, flags=re.M))
    assert names and not [n for n in names if n.startswith("_Z")]
    assert set(names) == declared, sorted(set(names) ^ declared)[:10]


def test_a_consumer_defining_the_librarys_internal_symbols_cannot_interpose_them(tmp_path):
    """Regression test for the round-1 failure: the executable defines and exports `hipbfv::Context::~Context()` (what the
    first hipbfv.hpp did by accident at -O0); SEALContext_Create must run the library's own destructor."""
    exe = _build(tmp_path, "-O0", os.path.join(ROOT, "tests", "native", "interpose_check.cpp"), ["-rdynamic"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "no interposition" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["-O0", "-O2"])
def test_simple_multiply_example_runs_on_the_device(tmp_path, opt):
    """examples/simple_multiply (main.rs:57-80) through include/hipbfv.hpp: keygen, encode, encrypt, multiply, relinearize,
    rotations, wire-format round trip, transparent-result error -- all on the GPU, from a plain g++-built executable."""
    exe = _build(tmp_path, opt)
    for _ in range(2):  # the round-1 failure was intermittent: run it twice
        out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert "15 * 5 = 75" in out.stdout
    if opt == "-O0":
        chk = _build(tmp_path, "-O0", os.path.join(ROOT, "tests", "native", "interpose_check.cpp"), ["-rdynamic"])
        out = subprocess.run([chk], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "SEALContext_Create -> 0x0" in out.stdout, (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_concurrent_handle_calls_are_combined_without_changing_a_bit(tmp_path):
    """sunscreen_runtime runs ready graph nodes from a rayon pool (run.rs:415-469): 24 native threads on one evaluator, mixed
    multiply / relinearize / rotate / square calls, one thread also asking for a transparent product.  The library combines
    concurrent calls into batched launches (capi.cpp Combiner); every thread must get the bits it gets alone, and only the
    offending call may fail.  Run with combining on (default) and off."""
    exe = _build(tmp_path, "-O1", os.path.join(ROOT, "tests", "native", "combine_check.cpp"), ["-lpthread"])
    # HIPBFV_NO_SMALL_BATCH=0: the product default (the `pipeline_selection` fixture of conftest.py pins it to 1 for GPU tests)
    for env in ({"HIPBFV_NO_SMALL_BATCH": "0"}, {"HIPBFV_NO_SMALL_BATCH": "0", "HIPBFV_NO_COMBINE": "1"}, {"HIPBFV_NO_SMALL_BATCH": "1"}):
        for _ in range(3):  # which calls meet in a batch differs from run to run
            out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
            assert out.returncode == 0 and "combine ok" in out.stdout, (env, out.returncode, out.stdout, out.stderr)


def test_native_thread_tools_build(tmp_path):
    """No GPU: the native multi-thread check and the drop-in throughput tool compile and link against the library with
    warnings as errors (they run in the GPU suite / by hand)."""
    _build(tmp_path, "-O1", os.path.join(ROOT, "tests", "native", "combine_check.cpp"), ["-lpthread"])
    _build(tmp_path, "-O2", os.path.join(ROOT, "tools", "mt_dropin.cpp"), ["-lpthread"])


def test_combining_protocol_under_thread_sanitizer(tmp_path):
    """No GPU: sunscreen_amd/csrc/flat_combiner.hpp -- the queueing protocol that turns concurrent handle-level calls into
    batches -- with a mock executor under ThreadSanitizer: 48 threads x 400 requests of three kinds, one and two leaders; every
    request executed exactly once in a batch of its own kind, results visible to the owner, never more batches in flight than
    leaders, no data race."""
    exe = str(tmp_path / "combiner_tsan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsanitize=thread", "-I", os.path.join(ROOT, "sunscreen_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "combiner_tsan.cpp"), "-lpthread", "-o", exe])
    for leaders in ("1", "2"):
        out = subprocess.run([exe, leaders], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "combiner ok" in out.stdout and "WARNING: ThreadSanitizer" not in out.stderr, (out.stdout, out.stderr[-3000:])
