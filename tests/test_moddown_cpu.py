"""Host proof of the FP64 form of the key switch's mod-down step (sunscreen_amd/csrc/moddown_d.hpp; SEAL
Evaluator::switch_key_inplace behind Evaluator_Relinearize / Evaluator_RotateRows, seal_fhe/src/evaluator_base.rs:214-240, :300-407).

The device code is plain IEEE double arithmetic (fma, add, rint), so its exactness is checked on the CPU against 128-bit
integers: tests/native/moddown_check.cpp includes the header the kernels would include and verifies, for data and special primes
of 36 ... 50 bits (p above and below q; primes at the top, the bottom and inside their size class), on random and extreme
operands, that the canonical result equals the one the tail kernels' 64-bit integer path computes."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fp64_mod_down_is_exact(tmp_path):
    exe = str(tmp_path / "moddown_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "sunscreen_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "moddown_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("ok ") and int(last.split()[1]) >= 4_000_000, out.stdout
