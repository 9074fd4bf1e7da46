"""ctypes wrapper around the CPU oracle (oracle/liboracle_bfv.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (sunscreen_amd/) never imports this module.

The oracle restates the arithmetic the reference runs inside Microsoft SEAL 4.0 behind
seal_fhe::Evaluator (seal_fhe/src/evaluator.rs:7-280); see oracle/bfv_oracle.h for the pinning
status of each piece.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_bfv.so")

u64p = C.POINTER(C.c_uint64)


def build(force: bool = False) -> str:
    """Compile the oracle with the committed Makefile (gcc only, no reference sources)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.ora_ctx_create.restype = C.c_void_p
        L.ora_ctx_create.argtypes = [C.c_uint32, u64p, C.c_size_t, C.c_uint64]
        L.ora_ctx_destroy.argtypes = [C.c_void_p]
        for name in ("ora_ctx_K", "ora_ctx_key_count", "ora_ctx_bsk_count"):
            getattr(L, name).restype = C.c_size_t
            getattr(L, name).argtypes = [C.c_void_p]
        L.ora_ctx_n.restype = C.c_uint32
        L.ora_ctx_n.argtypes = [C.c_void_p]
        L.ora_ctx_prime.restype = C.c_uint64
        L.ora_ctx_prime.argtypes = [C.c_void_p, C.c_size_t]
        L.ora_ctx_bsk_prime.restype = C.c_uint64
        L.ora_ctx_bsk_prime.argtypes = [C.c_void_p, C.c_size_t]
        L.ora_ctx_gamma.restype = C.c_uint64
        L.ora_ctx_gamma.argtypes = [C.c_void_p]
        L.ora_ctx_plain.restype = C.c_uint64
        L.ora_ctx_plain.argtypes = [C.c_void_p]
        L.ora_ctx_total_coeff_bits.restype = C.c_int
        L.ora_ctx_total_coeff_bits.argtypes = [C.c_void_p]
        L.ora_is_prime.restype = C.c_int
        L.ora_is_prime.argtypes = [C.c_uint64]
        L.ora_get_primes.restype = C.c_size_t
        L.ora_get_primes.argtypes = [C.c_uint64, C.c_int, C.c_size_t, u64p]
        L.ora_coeff_modulus_create.restype = C.c_int
        L.ora_coeff_modulus_create.argtypes = [C.c_uint32, C.POINTER(C.c_int), C.c_size_t, u64p]
        L.ora_plain_batching.restype = C.c_uint64
        L.ora_plain_batching.argtypes = [C.c_uint32, C.c_int]
        L.ora_bfv_default.restype = C.c_size_t
        L.ora_bfv_default.argtypes = [C.c_uint32, C.c_int, u64p]
        L.ora_minimal_primitive_root.restype = C.c_uint64
        L.ora_minimal_primitive_root.argtypes = [C.c_uint32, C.c_uint64]
        L.ora_galois_elt_from_step.restype = C.c_uint32
        L.ora_galois_elt_from_step.argtypes = [C.c_void_p, C.c_int]
        L.ora_galois_elts_all.restype = C.c_size_t
        L.ora_galois_elts_all.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.ora_seed.argtypes = [C.c_uint64]
        L.ora_bench_mul_relin.restype = C.c_double
        L.ora_bench_mul_relin.argtypes = [C.c_void_p, u64p, u64p, u64p, u64p, C.c_size_t, C.c_int]
        L.ora_bench_ntt.restype = C.c_double
        L.ora_bench_ntt.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_size_t, C.c_int]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(u64p)


def _arr(x, dtype=np.uint64) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=dtype))


# ----------------------------------------------------------------------------- number theory


def is_prime(v: int) -> bool:
    return bool(lib().ora_is_prime(v))


def get_primes(factor: int, bits: int, count: int) -> list[int]:
    out = np.zeros(count, dtype=np.uint64)
    k = lib().ora_get_primes(factor, bits, count, _p(out))
    return [int(x) for x in out[:k]]


def coeff_modulus_create(n: int, bit_sizes: list[int]) -> list[int]:
    """CoeffModulus::create (seal_fhe/src/modulus.rs:149-180)."""
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    bs = (C.c_int * len(bit_sizes))(*bit_sizes)
    rc = lib().ora_coeff_modulus_create(n, bs, len(bit_sizes), _p(out))
    if rc != 0:
        raise ValueError("cannot find enough primes")
    return [int(x) for x in out]


def plain_batching(n: int, bits: int) -> int:
    """PlainModulus::batching (seal_fhe/src/modulus.rs:100-116)."""
    return int(lib().ora_plain_batching(n, bits))


def bfv_default(n: int, sec: int = 128) -> list[int]:
    """CoeffModulus::bfv_default (seal_fhe/src/modulus.rs:182-205)."""
    out = np.zeros(32, dtype=np.uint64)
    k = lib().ora_bfv_default(n, sec, _p(out))
    return [int(x) for x in out[:k]]


def minimal_primitive_root(two_n: int, q: int) -> int:
    return int(lib().ora_minimal_primitive_root(two_n, q))


def seed(s: int) -> None:
    lib().ora_seed(s)


# ----------------------------------------------------------------------------- context


class Oracle:
    """A BFV context + evaluator + (test-only) client operations on numpy uint64 arrays.

    Ciphertexts are ``uint64[size][K][N]`` arrays, keys ``uint64[K][2][K+1][N]`` (NTT form).
    """

    def __init__(self, n: int, coeff_modulus: list[int], plain_modulus: int):
        L = lib()
        cm = _arr(coeff_modulus)
        self._h = L.ora_ctx_create(n, _p(cm), len(coeff_modulus), plain_modulus)
        if not self._h:
            raise ValueError("invalid BFV parameters")
        self.n = n
        self.t = plain_modulus
        self.key_primes = [int(x) for x in coeff_modulus]
        self.K = int(L.ora_ctx_K(self._h))
        self.KK = int(L.ora_ctx_key_count(self._h))
        self.primes = self.key_primes[: self.K]
        self.bsk = [int(L.ora_ctx_bsk_prime(self._h, j)) for j in range(L.ora_ctx_bsk_count(self._h))]
        self.gamma = int(L.ora_ctx_gamma(self._h))
        self.total_coeff_bits = int(L.ora_ctx_total_coeff_bits(self._h))

    def __del__(self):
        try:
            if self._h:
                lib().ora_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- NTT
    def ntt(self, idx: int, x: np.ndarray, inverse: bool = False) -> np.ndarray:
        y = _arr(x).copy()
        (lib().ora_ntt_inverse if inverse else lib().ora_ntt_forward)(C.c_void_p(self._h), C.c_size_t(idx), _p(y))
        return y

    def ntt_bsk(self, j: int, x: np.ndarray, inverse: bool = False) -> np.ndarray:
        y = _arr(x).copy()
        (lib().ora_ntt_inverse_bsk if inverse else lib().ora_ntt_forward_bsk)(
            C.c_void_p(self._h), C.c_size_t(j), _p(y)
        )
        return y

    # -- evaluator
    def _ct(self, a) -> np.ndarray:
        a = _arr(a)
        assert a.ndim == 3 and a.shape[1] == self.K and a.shape[2] == self.n, a.shape
        return a

    # SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT (seal_fhe/build.rs:46-66).  The C functions finish their result before they
    # report it transparent, so tests that push degenerate operands through the arithmetic (all-zero polynomials at the
    # edges of the BEHZ bounds) may switch the exception off and still read the bits.
    throw_on_transparent = True

    def _chk(self, rc: int):
        if rc == -2 and not self.throw_on_transparent:
            return
        if rc != 0:
            raise RuntimeError({-1: "invalid argument", -2: "transparent ciphertext", -3: "missing key"}.get(rc, str(rc)))

    def mod_switch_to_next(self, ct) -> np.ndarray:
        """Evaluator::mod_switch_to_next (BFV): uint64[size][K-1][N] at the next level (an Oracle built from
        key_primes[:K-1] + [special prime] continues from there)."""
        ct = self._ct(ct)
        out = np.zeros((ct.shape[0], self.K - 1, self.n), dtype=np.uint64)
        self._chk(lib().ora_mod_switch_to_next(C.c_void_p(self._h), _p(ct), C.c_size_t(ct.shape[0]), _p(out)))
        return out

    def next_level(self) -> "Oracle":
        return Oracle(self.n, self.key_primes[: self.K - 1] + self.key_primes[self.K :], self.t)

    def add(self, a, b, sub: bool = False) -> np.ndarray:
        a, b = self._ct(a), self._ct(b)
        out = np.zeros((max(a.shape[0], b.shape[0]), self.K, self.n), dtype=np.uint64)
        fn = lib().ora_sub if sub else lib().ora_add
        self._chk(fn(C.c_void_p(self._h), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]), _p(out)))
        return out

    def sub(self, a, b) -> np.ndarray:
        return self.add(a, b, sub=True)

    def negate(self, a) -> np.ndarray:
        a = self._ct(a)
        out = np.zeros_like(a)
        self._chk(lib().ora_negate(C.c_void_p(self._h), _p(a), C.c_size_t(a.shape[0]), _p(out)))
        return out

    def multiply(self, a, b) -> np.ndarray:
        a, b = self._ct(a), self._ct(b)
        out = np.zeros((a.shape[0] + b.shape[0] - 1, self.K, self.n), dtype=np.uint64)
        self._chk(
            lib().ora_multiply(C.c_void_p(self._h), _p(a), C.c_size_t(a.shape[0]), _p(b), C.c_size_t(b.shape[0]), _p(out))
        )
        return out

    def relinearize(self, ct3, rk) -> np.ndarray:
        ct3 = self._ct(ct3)
        assert ct3.shape[0] == 3
        rk = _arr(rk)
        out = np.zeros((2, self.K, self.n), dtype=np.uint64)
        self._chk(lib().ora_relinearize(C.c_void_p(self._h), _p(ct3), _p(rk), _p(out)))
        return out

    def multiply_many(self, cts, rk) -> np.ndarray:
        """SEAL 4.0 Evaluator::multiply_many (seal_fhe/src/evaluator.rs:38-50, bfv_evaluator.rs multiply_many; algorithm
        [RECALLED] native/src/seal/evaluator.cpp): NOT a left fold -- a work list.  First level: adjacent pairs
        (0,1), (2,3), ... are multiplied and relinearised, an odd last element is appended as it is; then, walking the list
        from the front, elements i and i+1 are multiplied, relinearised and APPENDED to the list until one element is
        left past the cursor; the last element is the result.  The order decides which products share a level, i.e. the
        noise and the bits.  (SEAL squares a pair whose operands alias; squaring gives the bits of multiply(x, x).)"""
        cts = [self._ct(c) for c in cts]
        if not cts:
            raise ValueError("encrypteds vector must not be empty")
        if len(cts) == 1:
            return cts[0].copy()
        work = [self.relinearize(self.multiply(cts[i], cts[i + 1]), rk) for i in range(0, len(cts) - 1, 2)]
        if len(cts) & 1:
            work.append(cts[-1].copy())
        i = 0
        while i + 1 < len(work):
            work.append(self.relinearize(self.multiply(work[i], work[i + 1]), rk))
            i += 2
        return work[-1]

    def exponentiate(self, ct, exponent: int, rk) -> np.ndarray:
        """SEAL 4.0 Evaluator::exponentiate_inplace (seal_fhe/src/evaluator.rs:84-157 exponentiate; [RECALLED]): exponent 0 is
        an error, 1 returns the input, otherwise multiply_many over `exponent` copies of the ciphertext."""
        if exponent == 0:
            raise ValueError("exponent cannot be 0")
        ct = self._ct(ct)
        return ct.copy() if exponent == 1 else self.multiply_many([ct] * exponent, rk)

    def _gk_ptrs(self, gk: dict[int, np.ndarray]):
        arr = (u64p * self.n)()
        keep = []
        for elt, key in gk.items():
            key = _arr(key)
            keep.append(key)
            arr[(elt - 1) >> 1] = _p(key)
        return arr, keep

    def apply_galois(self, ct, elt: int, gk: dict[int, np.ndarray]) -> np.ndarray:
        ct = self._ct(ct)
        arr, keep = self._gk_ptrs(gk)
        out = np.zeros_like(ct)
        self._chk(lib().ora_apply_galois(C.c_void_p(self._h), _p(ct), C.c_uint32(elt), arr, _p(out)))
        return out

    def rotate_rows(self, ct, steps: int, gk: dict[int, np.ndarray]) -> np.ndarray:
        ct = self._ct(ct)
        arr, keep = self._gk_ptrs(gk)
        out = np.zeros_like(ct)
        self._chk(lib().ora_rotate_rows(C.c_void_p(self._h), _p(ct), C.c_int(steps), arr, _p(out)))
        return out

    def rotate_columns(self, ct, gk: dict[int, np.ndarray]) -> np.ndarray:
        ct = self._ct(ct)
        arr, keep = self._gk_ptrs(gk)
        out = np.zeros_like(ct)
        self._chk(lib().ora_rotate_columns(C.c_void_p(self._h), _p(ct), arr, _p(out)))
        return out

    def galois_elt_from_step(self, step: int) -> int:
        return int(lib().ora_galois_elt_from_step(C.c_void_p(self._h), step))

    def apply_galois_poly(self, poly, elt: int) -> np.ndarray:
        poly = _arr(poly)
        out = np.zeros_like(poly)
        self._chk(lib().ora_apply_galois_poly(C.c_void_p(self._h), _p(poly), C.c_uint32(elt), _p(out)))
        return out

    def _plain_op(self, fn, ct, plain) -> np.ndarray:
        ct = self._ct(ct)
        plain = _arr(plain)
        out = np.zeros_like(ct)
        self._chk(fn(C.c_void_p(self._h), _p(ct), C.c_size_t(ct.shape[0]), _p(plain), C.c_size_t(plain.size), _p(out)))
        return out

    def add_plain(self, ct, plain) -> np.ndarray:
        return self._plain_op(lib().ora_add_plain, ct, plain)

    def sub_plain(self, ct, plain) -> np.ndarray:
        return self._plain_op(lib().ora_sub_plain, ct, plain)

    def multiply_plain(self, ct, plain) -> np.ndarray:
        return self._plain_op(lib().ora_multiply_plain, ct, plain)

    def behz_extend(self, poly_q) -> np.ndarray:
        poly_q = _arr(poly_q)
        out = np.zeros((len(self.bsk), self.n), dtype=np.uint64)
        lib().ora_behz_extend(C.c_void_p(self._h), _p(poly_q), _p(out))
        return out

    def behz_floor_sk(self, poly_q_bsk) -> np.ndarray:
        poly_q_bsk = _arr(poly_q_bsk)
        out = np.zeros((self.K, self.n), dtype=np.uint64)
        lib().ora_behz_floor_sk(C.c_void_p(self._h), _p(poly_q_bsk), _p(out))
        return out

    # -- client side (test only)
    def keygen(self, relin: bool = True, galois_elts: list[int] | str | None = None):
        h = C.c_void_p(self._h)
        sk = np.zeros((self.KK, self.n), dtype=np.uint64)
        lib().ora_keygen_secret(h, _p(sk))
        pk = np.zeros((2, self.KK, self.n), dtype=np.uint64)
        lib().ora_keygen_public(h, _p(sk), _p(pk))
        rk = None
        if relin and self.KK > 1:
            rk = np.zeros((self.K, 2, self.KK, self.n), dtype=np.uint64)
            lib().ora_keygen_relin(h, _p(sk), _p(rk))
        gk = {}
        if galois_elts == "all":
            buf = (C.c_uint32 * 64)()
            cnt = lib().ora_galois_elts_all(h, buf)
            galois_elts = [int(buf[i]) for i in range(cnt)]
        for elt in galois_elts or []:
            key = np.zeros((self.K, 2, self.KK, self.n), dtype=np.uint64)
            lib().ora_keygen_galois(h, _p(sk), C.c_uint32(elt), _p(key))
            gk[int(elt)] = key
        return sk, pk, rk, gk

    def encrypt(self, pk, plain) -> np.ndarray:
        plain = _arr(plain)
        ct = np.zeros((2, self.K, self.n), dtype=np.uint64)
        lib().ora_encrypt(C.c_void_p(self._h), _p(_arr(pk)), _p(plain), C.c_size_t(plain.size), _p(ct))
        return ct

    def encrypt_symmetric(self, sk, plain) -> np.ndarray:
        plain = _arr(plain)
        ct = np.zeros((2, self.K, self.n), dtype=np.uint64)
        lib().ora_encrypt_symmetric(C.c_void_p(self._h), _p(_arr(sk)), _p(plain), C.c_size_t(plain.size), _p(ct))
        return ct

    def decrypt(self, ct, sk) -> np.ndarray:
        ct = self._ct(ct)
        out = np.zeros(self.n, dtype=np.uint64)
        lib().ora_decrypt(C.c_void_p(self._h), _p(ct), C.c_size_t(ct.shape[0]), _p(_arr(sk)), _p(out))
        return out

    def dot_with_secret(self, ct, sk) -> np.ndarray:
        ct = self._ct(ct)
        out = np.zeros((self.K, self.n), dtype=np.uint64)
        lib().ora_dot_with_secret(C.c_void_p(self._h), _p(ct), C.c_size_t(ct.shape[0]), _p(_arr(sk)), _p(out))
        return out

    def noise_budget(self, ct, sk) -> int:
        """Decryptor::invariant_noise_budget (seal_fhe/src/encryptor_decryptor.rs:640-660):
        bits(q) - bits(|t * ct(s) mod q| centred) - 1, computed with Python integers."""
        d = self.dot_with_secret(ct, sk)
        q = 1
        for p in self.primes:
            q *= p
        # CRT-compose each coefficient
        coefs = [0] * self.n
        for i, p in enumerate(self.primes):
            qi = q // p
            inv = pow(qi % p, -1, p)
            col = d[i]
            for k in range(self.n):
                coefs[k] += int(col[k]) * inv % p * qi
        worst = 0
        for k in range(self.n):
            v = (coefs[k] % q) * self.t % q
            if v > q // 2:
                v = q - v
            worst = max(worst, v)
        bits = worst.bit_length()
        return max(0, q.bit_length() - bits - 1)

    def batch_encode(self, values) -> np.ndarray:
        values = _arr(values)
        assert values.size == self.n
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().ora_batch_encode(C.c_void_p(self._h), _p(values), _p(out)) != 0:
            raise ValueError("batching unsupported / value out of range")
        return out

    def batch_decode(self, plain) -> np.ndarray:
        plain = _arr(plain)
        if plain.size < self.n:
            plain = np.concatenate([plain, np.zeros(self.n - plain.size, dtype=np.uint64)])
        out = np.zeros(self.n, dtype=np.uint64)
        if lib().ora_batch_decode(C.c_void_p(self._h), _p(plain), _p(out)) != 0:
            raise ValueError("batching unsupported")
        return out

    # -- cpu_baseline timing legs (bench.py)
    def bench_mul_relin(self, a, b, rk, threads: int = 1):
        a, b, rk = _arr(a), _arr(b), _arr(rk)
        out = np.zeros_like(a)
        secs = lib().ora_bench_mul_relin(C.c_void_p(self._h), _p(a), _p(b), _p(rk), _p(out), a.shape[0], threads)
        return secs, out

    def bench_ntt(self, x, nprimes: int, threads: int = 1):
        x = _arr(x).copy()
        secs = lib().ora_bench_ntt(C.c_void_p(self._h), _p(x), x.shape[0], nprimes, threads)
        return secs, x
