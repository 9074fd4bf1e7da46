"""The generator behind keys and encryption randomness: ChaCha20 as a counter-based PRF under 256-bit
keys derived from a 512-bit seed, OS entropy via getrandom(2) with no fallback (sunscreen_amd/csrc/rng.hpp)."""
import os
import subprocess

import numpy as np
import pytest

from tests import rng_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_function_is_rfc8439_chacha20_and_seed_derivation_properties(tmp_path):
    exe = str(tmp_path / "rng_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(ROOT, "tests", "native", "rng_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout)
    # the numpy restatement the GPU test uses agrees with the header (seed 0x1234: secret.k[0], pub.k[0], one block word)
    words = out.stdout.split()[2:]
    t = rng_ref.seed_from_u64_for_tests(0x1234)
    blk = rng_ref.chacha20_block(t["secret"], 7, 0, 0, 0x5E)
    assert [f"{t['secret'][0]:08x}", f"{t['pub'][0]:08x}", f"{int(blk[0]):08x}"] == words
    # RFC 8439 2.3.2 through the numpy restatement too
    key = np.frombuffer(bytes(range(32)), dtype="<u4")
    got = rng_ref.chacha20_block(key, 1, 0x09000000, 0x4A000000, 0)
    assert [int(got[0]), int(got[1]), int(got[15])] == [0xE4E7F110, 0x15593BD1, 0x4E3C50A2]


def test_no_weak_generator_or_fallback_seed_in_the_product():
    src = ""
    for f in ("capi.cpp", "kernels_client.hip", "evaluator_client.cpp", "rng.hpp"):
        src += open(os.path.join(ROOT, "sunscreen_amd", "csrc", f)).read()
    assert "philox" not in src.lower()
    assert "/dev/urandom" not in src and "getrandom(" in src
    assert "uintptr_t)e ^" not in src and "uintptr_t)salt" not in src  # the pointer-derived fallback seeds are gone


@pytest.mark.gpu
def test_device_generator_is_the_documented_chacha20():
    """The ternary secret of KeyGenerator(seed=s) is ((w0 * 3) >> 32) - 1 with w0 = word 0 of
    ChaCha20_block(secret key of seed s, (x, stream 0, 0, 0x5E)) -- recomputed here on the host; and a Galois-free public
    key's uniform polynomial comes from the OTHER key."""
    from oracle import bfv_oracle as O
    from sunscreen_amd import Context, KeyGenerator

    n = 4096
    primes, t = O.bfv_default(n), O.plain_batching(n, 16)
    o = O.Oracle(n, primes, t)
    ctx = Context.from_raw(n, primes, t)
    seed = 0x1234
    kg = KeyGenerator(ctx, seed=seed)
    sk = kg.secret_key().to_array(ctx)
    s = o.ntt(0, sk[0], inverse=True).astype(np.int64)
    s = np.where(s > primes[0] // 2, s - primes[0], s)
    keys = rng_ref.seed_from_u64_for_tests(seed)
    w0 = rng_ref.chacha20_block(keys["secret"], np.arange(n), 0, 0, 0x5E)[0]
    expect = ((w0 * np.uint64(3)) >> np.uint64(32)).astype(np.int64) - 1
    assert (s == expect).all()
    # public key = first key stream (stream 1 -> block input 1 << 8): pk1 = a, 128 uniform bits >> 1 reduced mod q_i
    pk = kg.create_public_key().to_array(ctx)
    blk = rng_ref.chacha20_block(keys["pub"], np.arange(n), 1 << 8, 0, 0xA0)
    for i in range(len(primes)):
        r = [blk[4 * i + j].astype(object) for j in range(4)]
        wide = (((r[0] << 32) | r[1]) << 64) | ((r[2] << 32) | r[3])
        assert (pk[1, i].astype(object) == (wide >> 1) % primes[i]).all(), i
    # two generators seeded by the OS differ (512 bits of getrandom each)
    a, b = KeyGenerator(ctx).secret_key().to_array(ctx), KeyGenerator(ctx).secret_key().to_array(ctx)
    assert not (a == b).all()
