"""Host proof of the exact sum-of-products used by the head / tail kernels' base conversions
(sunscreen_amd/csrc/griddot.hpp; SEAL RNSTool fastbconv behind Evaluator_Multiply, seal_fhe/src/evaluator_base.rs:198-212).

The device code is plain IEEE double arithmetic (fma, add, rint), so its exactness is checked here on the CPU against
128-bit integers: tests/native/griddot_check.cpp includes the same header the kernels include, asks plan_grid_dot() for a
grid and, for every accepted bound (the n=8192 and n=16384 default sizes among them), verifies on random, maximal and
signed operands that (a) the two accumulators are the exact sum, (b) reduce(H) + err is an exact integer below 2^51
congruent to it, (c) the canonical residue the kernels derive from it is the true one."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grid_dot_is_exact(tmp_path):
    exe = str(tmp_path / "griddot_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "sunscreen_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "griddot_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("ok "), out.stdout
    # the default parameter sets must be among the planned bounds (44/46-bit x 5 terms, 49/45-bit x 9 terms, and -- centred constants --
    # 49/47-bit x 8 terms: n = 16384 with the nine 48-bit auxiliary primes of the derived base bound)
    assert int(last.split("planned=")[1]) >= 7, out.stdout
