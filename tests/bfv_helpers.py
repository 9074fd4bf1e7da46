"""Shared helpers for the parity tests: parameter sets and signed batch encode/decode.

Parameter sets mirror the reference's fixtures:
  * "seal_fhe_unit"  : seal_fhe/src/bfv_evaluator.rs:255-282  n=8192, create(8192,[50,30,30,50,50]), batching(8192,32)
  * "default_N"      : seal_fhe/tests/test_common.rs:3-29     bfv_default(n, TC128) + batching(n, lane_bits)
  * "simple_multiply": BASELINE.json configs[0]: n=4096 default primes, t=262144 (sunscreen/src/compiler.rs:149-157)
"""
from __future__ import annotations

import functools

import numpy as np

from oracle import bfv_oracle as O


def params(name: str):
    if name == "seal_fhe_unit":
        return 8192, O.coeff_modulus_create(8192, [50, 30, 30, 50, 50]), O.plain_batching(8192, 32)
    if name.startswith("default_"):
        parts = name.split("_")
        n = int(parts[1])
        bits = int(parts[2]) if len(parts) > 2 else 17
        return n, O.bfv_default(n), O.plain_batching(n, bits)
    if name == "simple_multiply":
        return 4096, O.bfv_default(4096), 262144
    if name == "toy_64":
        return 64, O.coeff_modulus_create(64, [36, 36, 37]), 257
    raise KeyError(name)


@functools.lru_cache(maxsize=None)
def oracle_for(name: str) -> O.Oracle:
    n, primes, t = params(name)
    return O.Oracle(n, primes, t)


def encode_signed(o: O.Oracle, vals) -> np.ndarray:
    v = np.asarray(vals, dtype=np.int64) % o.t
    return o.batch_encode(v.astype(np.uint64))


def decode_signed(o: O.Oracle, plain) -> np.ndarray:
    v = o.batch_decode(plain).astype(np.int64)
    return np.where(v > o.t // 2, v - o.t, v)


def make_vec(n: int) -> np.ndarray:
    # seal_fhe/src/bfv_evaluator.rs:284-292
    return np.array([n // 2 - i for i in range(n)], dtype=np.int64)


def make_small_vec(n: int) -> np.ndarray:
    # seal_fhe/src/bfv_evaluator.rs:294-302
    return np.array([16 - i % 32 for i in range(n)], dtype=np.int64)
