"""include/hipbfv.hpp -- the compiled-language host mirror of the `seal_fhe` crate's surface (the reference's host side is
Rust; no Rust toolchain exists here) -- builds against the C ABI and behaves like the crate: examples/simple_multiply.cpp is
the reference's examples/simple_multiply (15 * 5 = 75, examples/simple_multiply/src/main.rs:57-80) plus the rotation / wire
format / transparent-ciphertext checks of seal_fhe/src/bfv_evaluator.rs:322-970, written against that header.

The device half (`./simple_multiply` without arguments) ran on the MI355X box (profiles/r01_final_cpp_simple_multiply.log);
it is not part of the `-m gpu` suite yet: see DESIGN.md section 10 (symbol visibility of the library)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "simple_multiply")
    lib = os.path.join(ROOT, "sunscreen_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "simple_multiply.cpp"), "-L", lib, "-lhipbfv", "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_cpp_mirror_builds_and_its_host_side_behaves_like_the_crate(tmp_path):
    """No GPU: parameter objects, prime-generation known answers (modulus.rs:279-313), builder errors, HRESULT -> Error
    mapping (error.rs:65-78), deep copies / moves, Plaintext from a polynomial string."""
    out = subprocess.run([_build(tmp_path), "--host-only"], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "host-only ok" in out.stdout
