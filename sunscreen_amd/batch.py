"""GPU batch executor: Evaluator operations over device-resident batches of ciphertexts.

Replaces the reference's per-node dispatch (sunscreen_runtime/src/run.rs:160-341: one
`evaluator.<op>()` FFI call and at least one allocation per graph node per ciphertext) with one
sequence of kernel launches per operation over `count` independent ciphertexts.  Tensors are
`torch.int64` CUDA tensors holding uint64 bit patterns, shape [count, size, K, N]; PyTorch is used
only for device memory and streams -- all arithmetic happens in libhipbfv.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np
import torch

from . import _lib
from .seal import BFVEvaluator, Context, GaloisKeys, RelinearizationKeys, _check


def to_device(a: np.ndarray, device: str = "cuda:0") -> torch.Tensor:
    a = np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    return torch.from_numpy(a.view(np.int64)).to(device)


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


def _ptr(t: torch.Tensor):
    assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class BatchEvaluator:
    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._ev = BFVEvaluator(ctx)
        self._h = self._ev.get_handle()
        self.n, self.K, self.KK = ctx.poly_modulus_degree, ctx.K, ctx.KK

    def check(self) -> None:
        """Raise HipBfvError (COR_E_INVALIDOPERATION, as the reference's runtime does: sunscreen/tests/features.rs:8-34) if
        any batched operation since the last call produced a transparent ciphertext; synchronises the current stream.
        The operations themselves stay asynchronous: call this where the reference would have seen the error -- at the
        latest before results leave the device."""
        first = C.c_uint64()
        _check(_lib.load().hipbfv_batch_status(self._h, C.byref(first), _stream()))

    def set_transparent_check(self, enabled: bool) -> None:
        _check(_lib.load().hipbfv_set_batch_transparent_check(self._h, enabled))

    def set_chunk_ops(self, chunk: int) -> None:
        _check(_lib.load().hipbfv_set_chunk_ops(self._h, chunk))

    # ---- per-kernel HIP-event timing (bench.py) ----
    def profile(self, enabled: bool = True) -> None:
        _check(_lib.load().hipbfv_profile_enable(self._h, enabled))

    def profile_reset(self) -> None:
        _check(_lib.load().hipbfv_profile_reset(self._h))

    def profile_read(self) -> dict[str, dict]:
        L = _lib.load()
        cnt = C.c_uint32()
        _check(L.hipbfv_profile_kernel_count(C.byref(cnt)))
        out = {}
        for i in range(cnt.value):
            name = C.create_string_buffer(64)
            ms, launches, units = C.c_double(), C.c_uint64(), C.c_uint64()
            _check(L.hipbfv_profile_read(self._h, i, name, 64, C.byref(ms), C.byref(launches), C.byref(units)))
            if launches.value:
                out[name.value.decode()] = {"ms": ms.value, "launches": launches.value, "units": units.value}
        return out

    def _shape_ok(self, t: torch.Tensor, size=None):
        assert t.dim() == 4 and t.shape[2] == self.K and t.shape[3] == self.n, tuple(t.shape)
        if size is not None:
            assert t.shape[1] == size, tuple(t.shape)

    def _new(self, count: int, size: int, like: torch.Tensor) -> torch.Tensor:
        return torch.empty((count, size, self.K, self.n), dtype=torch.int64, device=like.device)

    # ---- a1 / a2 ----
    def multiply(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a)
        self._shape_ok(b)
        count, sa, sb = a.shape[0], a.shape[1], b.shape[1]
        assert b.shape[0] == count
        out = out if out is not None else self._new(count, sa + sb - 1, a)
        _check(_lib.load().hipbfv_batch_multiply(self._h, _ptr(a), sa, _ptr(b), sb, _ptr(out), count, _stream()))
        return out

    def relinearize(self, ct3: torch.Tensor, rk: RelinearizationKeys, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct3, 3)
        out = out if out is not None else self._new(ct3.shape[0], 2, ct3)
        _check(_lib.load().hipbfv_batch_relinearize(self._h, _ptr(ct3), rk.get_handle(), _ptr(out), ct3.shape[0], _stream()))
        return out

    def multiply_relin(self, a: torch.Tensor, b: torch.Tensor, rk: RelinearizationKeys, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a, 2)
        self._shape_ok(b, 2)
        assert a.shape[0] == b.shape[0]
        out = out if out is not None else self._new(a.shape[0], 2, a)
        _check(_lib.load().hipbfv_batch_multiply_relin(self._h, _ptr(a), _ptr(b), rk.get_handle(), _ptr(out), a.shape[0], _stream()))
        return out

    # ---- a3 ----
    def apply_galois(self, ct: torch.Tensor, galois_elt: int, gk: GaloisKeys, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        _check(_lib.load().hipbfv_batch_apply_galois(self._h, _ptr(ct), galois_elt, gk.get_handle(), _ptr(out), ct.shape[0], _stream()))
        return out

    def rotate_rows(self, ct: torch.Tensor, steps: int, gk: GaloisKeys, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        _check(_lib.load().hipbfv_batch_rotate_rows(self._h, _ptr(ct), steps, gk.get_handle(), _ptr(out), ct.shape[0], _stream()))
        return out

    def rotate_columns(self, ct: torch.Tensor, gk: GaloisKeys, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        _check(_lib.load().hipbfv_batch_rotate_columns(self._h, _ptr(ct), gk.get_handle(), _ptr(out), ct.shape[0], _stream()))
        return out

    # ---- per-key batches (multi-tenant: the reference passes the keys per call, sunscreen_runtime/src/run.rs:100-105) ----
    @staticmethod
    def _key_sets(key_sets, key_index, count: int):
        handles = (C.c_void_p * len(key_sets))(*[k.get_handle() for k in key_sets])
        idx = np.ascontiguousarray(np.asarray(key_index, dtype=np.uint32))
        assert idx.shape == (count,), (idx.shape, count)
        return handles, len(key_sets), idx.ctypes.data_as(C.POINTER(C.c_uint32)), idx

    def relinearize_keys(self, ct3: torch.Tensor, key_sets: Sequence[RelinearizationKeys], key_index, out: torch.Tensor | None = None) -> torch.Tensor:
        """Item i is relinearised with key_sets[key_index[i]] (key_index: `count` host integers)."""
        self._shape_ok(ct3, 3)
        out = out if out is not None else self._new(ct3.shape[0], 2, ct3)
        hs, n, ip, _keep = self._key_sets(key_sets, key_index, ct3.shape[0])
        _check(_lib.load().hipbfv_batch_relinearize_keys(self._h, _ptr(ct3), hs, n, ip, _ptr(out), ct3.shape[0], _stream()))
        return out

    def multiply_relin_keys(self, a: torch.Tensor, b: torch.Tensor, key_sets: Sequence[RelinearizationKeys], key_index,
                            out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a, 2)
        self._shape_ok(b, 2)
        assert a.shape[0] == b.shape[0]
        out = out if out is not None else self._new(a.shape[0], 2, a)
        hs, n, ip, _keep = self._key_sets(key_sets, key_index, a.shape[0])
        _check(_lib.load().hipbfv_batch_multiply_relin_keys(self._h, _ptr(a), _ptr(b), hs, n, ip, _ptr(out), a.shape[0], _stream()))
        return out

    def apply_galois_keys(self, ct: torch.Tensor, galois_elt: int, key_sets: Sequence[GaloisKeys], key_index, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        hs, n, ip, _keep = self._key_sets(key_sets, key_index, ct.shape[0])
        _check(_lib.load().hipbfv_batch_apply_galois_keys(self._h, _ptr(ct), galois_elt, hs, n, ip, _ptr(out), ct.shape[0], _stream()))
        return out

    def rotate_rows_keys(self, ct: torch.Tensor, steps: int, key_sets: Sequence[GaloisKeys], key_index, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        hs, n, ip, _keep = self._key_sets(key_sets, key_index, ct.shape[0])
        _check(_lib.load().hipbfv_batch_rotate_rows_keys(self._h, _ptr(ct), steps, hs, n, ip, _ptr(out), ct.shape[0], _stream()))
        return out

    def rotate_columns_keys(self, ct: torch.Tensor, key_sets: Sequence[GaloisKeys], key_index, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(ct, 2)
        out = out if out is not None else self._new(ct.shape[0], 2, ct)
        hs, n, ip, _keep = self._key_sets(key_sets, key_index, ct.shape[0])
        _check(_lib.load().hipbfv_batch_rotate_columns_keys(self._h, _ptr(ct), hs, n, ip, _ptr(out), ct.shape[0], _stream()))
        return out

    # ---- a5 ----
    def add(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a)
        assert a.shape == b.shape
        out = out if out is not None else torch.empty_like(a)
        _check(_lib.load().hipbfv_batch_add(self._h, _ptr(a), _ptr(b), _ptr(out), a.shape[1], a.shape[0], _stream()))
        return out

    def sub(self, a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a)
        assert a.shape == b.shape
        out = out if out is not None else torch.empty_like(a)
        _check(_lib.load().hipbfv_batch_sub(self._h, _ptr(a), _ptr(b), _ptr(out), a.shape[1], a.shape[0], _stream()))
        return out

    def negate(self, a: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        self._shape_ok(a)
        out = out if out is not None else torch.empty_like(a)
        _check(_lib.load().hipbfv_batch_negate(self._h, _ptr(a), _ptr(out), a.shape[1], a.shape[0], _stream()))
        return out

    def _plain(self, fn, ct: torch.Tensor, plain: torch.Tensor, out):
        self._shape_ok(ct)
        assert plain.shape[-1] == self.n
        stride = 0 if plain.dim() == 1 or plain.shape[0] == 1 else self.n
        if stride:
            assert plain.shape[0] == ct.shape[0]
        out = out if out is not None else torch.empty_like(ct)
        _check(fn(self._h, _ptr(ct), ct.shape[1], _ptr(plain), stride, _ptr(out), ct.shape[0], _stream()))
        return out

    def add_plain(self, ct, plain, out=None):
        return self._plain(_lib.load().hipbfv_batch_add_plain, ct, plain, out)

    def sub_plain(self, ct, plain, out=None):
        return self._plain(_lib.load().hipbfv_batch_sub_plain, ct, plain, out)

    # ---- a4 ----
    def multiply_plain(self, ct, plain, out=None):
        return self._plain(_lib.load().hipbfv_batch_multiply_plain, ct, plain, out)

    # ---- 8f row 3: the steps either side of the path, on device-resident batches ----
    def encode(self, values: torch.Tensor, signed: bool = False) -> torch.Tensor:
        """BatchEncoder: int64[batch, N] slot values -> int64[batch, N] plaintext coefficients."""
        assert values.dim() == 2 and values.shape[1] == self.n
        out = torch.empty_like(values)
        _check(_lib.load().hipbfv_batch_encode(self._h, _ptr(values), _ptr(out), values.shape[0], int(signed), _stream()))
        return out

    def decode(self, plain: torch.Tensor, signed: bool = False) -> torch.Tensor:
        assert plain.dim() == 2 and plain.shape[1] == self.n
        out = torch.empty_like(plain)
        _check(_lib.load().hipbfv_batch_decode(self._h, _ptr(plain), _ptr(out), plain.shape[0], int(signed), _stream()))
        return out

    def decrypt(self, ct: torch.Tensor, secret_key) -> torch.Tensor:
        """int64[batch, size, K, N] -> int64[batch, N] plaintext coefficients (zero padded); secret_key: seal.SecretKey."""
        assert ct.dim() == 4 and ct.shape[2] == self.K and ct.shape[3] == self.n
        out = torch.empty((ct.shape[0], self.n), dtype=torch.int64, device=ct.device)
        _check(_lib.load().hipbfv_batch_decrypt(self._h, _ptr(ct), ct.shape[1], secret_key.get_handle(), _ptr(out), ct.shape[0], _stream()))
        return out

    def encrypt(self, plain: torch.Tensor, public_key, seed: int | bytes | None = None, first_op: int = 0) -> torch.Tensor:
        """int64[batch, N] (or one shared int64[N]) plaintexts -> fresh encryptions int64[batch', 2, K, N].
        seed: None = 512 fresh bits from the OS (production); 64 bytes = SEAL's prng_seed_type; an int = the TEST-ONLY
        64-bit seed (reproducible batches)."""
        shared = plain.dim() == 1
        count = 1 if shared else plain.shape[0]
        out = torch.empty((count, 2, self.K, self.n), dtype=torch.int64, device=plain.device)
        if isinstance(seed, int):
            _check(_lib.load().hipbfv_batch_encrypt(self._h, _ptr(plain), 0 if shared else self.n, public_key.get_handle(), seed, first_op,
                                                    _ptr(out), count, _stream()))
        else:
            assert seed is None or len(seed) == 64
            _check(_lib.load().hipbfv_batch_encrypt_seeded(self._h, _ptr(plain), 0 if shared else self.n, public_key.get_handle(), seed, first_op,
                                                           _ptr(out), count, _stream()))
        return out

    def mod_switch(self, ct: torch.Tensor) -> torch.Tensor:
        """mod_switch_to_next on a batch: int64[batch, size, K, N] -> int64[batch, size, K-1, N], the layout of
        `self.ctx.next_level()` (build a BatchEvaluator on that context to continue there)."""
        assert ct.dim() == 4 and ct.shape[2] == self.K and ct.shape[3] == self.n
        out = torch.empty((ct.shape[0], ct.shape[1], self.K - 1, self.n), dtype=torch.int64, device=ct.device)
        _check(_lib.load().hipbfv_batch_mod_switch(self._h, _ptr(ct), ct.shape[1], _ptr(out), ct.shape[0], _stream()))
        return out

    # ---- plaintext-matrix x ciphertext-vector products (examples/pir) ----
    def plain_to_ntt(self, plain: torch.Tensor) -> torch.Tensor:
        """int64[..., N] plaintexts -> int64[..., K, N]: the transform-domain operand multiply_plain builds internally."""
        assert plain.shape[-1] == self.n
        flat = plain.reshape(-1, self.n).contiguous()
        out = torch.empty((flat.shape[0], self.K, self.n), dtype=torch.int64, device=plain.device)
        _check(_lib.load().hipbfv_batch_plain_to_ntt(self._h, _ptr(flat), self.n, _ptr(out), flat.shape[0], _stream()))
        # an all-zero plaintext has no transformed form (SEAL refuses every product with it, and the consumers of `out` cannot
        # see it any more): the producer recorded it, and this is where the caller learns -- static data is transformed once.
        # check() synchronises the stream and reads-and-resets the evaluator's ONE status word: what it raises covers every
        # batched operation since the previous check, not this call alone ("Asynchronous status").
        self.check()
        return out.reshape(tuple(plain.shape[:-1]) + (self.K, self.n))

    def ct_to_ntt(self, ct: torch.Tensor) -> torch.Tensor:
        assert ct.dim() == 4 and ct.shape[2] == self.K and ct.shape[3] == self.n
        out = torch.empty_like(ct)
        _check(_lib.load().hipbfv_batch_ct_to_ntt(self._h, _ptr(ct), ct.shape[1], _ptr(out), ct.shape[0], _stream()))
        return out

    def dot_plain_ntt(self, ctn: torch.Tensor, pntt: torch.Tensor) -> torch.Tensor:
        """ctn: int64[cols, 2, K, N] (ct_to_ntt), pntt: int64[rows, cols, K, N] (plain_to_ntt) ->
        int64[rows, 2, K, N] = sum_j multiply_plain(ct_j, plain[row][j]), coefficient form."""
        cols, rows = ctn.shape[0], pntt.shape[0]
        assert ctn.shape[1] == 2 and pntt.shape[1] == cols and pntt.shape[2] == self.K
        out = torch.empty((rows, 2, self.K, self.n), dtype=torch.int64, device=ctn.device)
        _check(_lib.load().hipbfv_batch_dot_plain_ntt(self._h, _ptr(ctn), cols, _ptr(pntt), rows, _ptr(out), _stream()))
        return out

    # ---- a6: NTT entry points (BASELINE config 2) ----
    def ntt(self, data: torch.Tensor, nprimes: int, inverse: bool = False) -> torch.Tensor:
        """In-place negacyclic NTT of int64[polys, N]; polynomial p uses key-level prime p % nprimes."""
        assert data.dim() == 2 and data.shape[1] == self.n
        _check(_lib.load().hipbfv_batch_ntt(self._h, _ptr(data), data.shape[0], nprimes, inverse, _stream()))
        return data
