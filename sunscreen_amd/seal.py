"""Host-side mirror of the `seal_fhe` crate surface for the evaluator path, on top of the C ABI.

The reference's host layer for this path is the Rust crate `seal_fhe` (seal_fhe/src/*.rs); no Rust
toolchain exists in this environment, so the same interface is mirrored here in Python with the
same type and method names, argument meaning and error behaviour, each method calling exactly the
C entry point the Rust method calls (cited per method).  This lets the parity tests read like the
reference's own tests (seal_fhe/src/bfv_evaluator.rs:322-970).

Nothing here computes on the CPU: every operation is a call into libhipbfv.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

import numpy as np

from . import _lib


class HipBfvError(Exception):
    """Mirror of seal_fhe::Error (seal_fhe/src/error.rs:10-78)."""

    NAMES = {
        _lib.E_POINTER: "InvalidPointer",
        _lib.E_INVALIDARG: "InvalidArgument",
        _lib.E_OUTOFMEMORY: "OutOfMemory",
        _lib.E_UNEXPECTED: "Unexpected",
        _lib.COR_E_IO: "InternalError",
        _lib.COR_E_INVALIDOPERATION: "InternalError",
    }

    def __init__(self, hresult: int, detail: str = ""):
        self.hresult = hresult & 0xFFFFFFFF
        self.kind = self.NAMES.get(self.hresult, "Unknown")
        super().__init__(f"{self.kind} (0x{self.hresult:08X}){': ' + detail if detail else ''}")


def _check(hr: int) -> None:
    """convert_seal_error (seal_fhe/src/error.rs:82-91)."""
    if hr != 0:
        raise HipBfvError(hr, _lib.last_error())


class SecurityLevel:
    """seal_fhe/src/context.rs:14-40"""

    NONE = 0
    TC128 = 128
    TC192 = 192
    TC256 = 256


class Modulus:
    """seal_fhe/src/modulus.rs:95-131"""

    def __init__(self, value: int):
        self._h = C.c_void_p()
        _check(_lib.load().Modulus_Create1(value, C.byref(self._h)))

    @classmethod
    def _adopt(cls, handle: int) -> "Modulus":
        m = cls.__new__(cls)
        m._h = C.c_void_p(handle)
        return m

    def value(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().Modulus_Value(self._h, C.byref(v)))
        return v.value

    def get_handle(self):
        return self._h

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Modulus_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None

    def __eq__(self, other):
        return isinstance(other, Modulus) and self.value() == other.value()

    def __repr__(self):
        return f"Modulus({self.value()})"


class CoefficientModulus:
    """seal_fhe/src/modulus.rs:149-250"""

    @staticmethod
    def create(degree: int, bit_sizes: Sequence[int]) -> list[Modulus]:
        n = len(bit_sizes)
        bits = (C.c_int * n)(*bit_sizes)
        out = (C.c_void_p * n)()
        _check(_lib.load().CoeffModulus_Create1(degree, n, bits, out))
        return [Modulus._adopt(out[i]) for i in range(n)]

    @staticmethod
    def bfv_default(degree: int, security_level: int = SecurityLevel.TC128) -> list[Modulus]:
        length = C.c_uint64()
        L = _lib.load()
        _check(L.CoeffModulus_BFVDefault(degree, security_level, C.byref(length), None))
        out = (C.c_void_p * length.value)()
        _check(L.CoeffModulus_BFVDefault(degree, security_level, C.byref(length), out))
        return [Modulus._adopt(out[i]) for i in range(length.value)]

    @staticmethod
    def max_bit_count(degree: int, security_level: int = SecurityLevel.TC128) -> int:
        bits = C.c_int()
        _check(_lib.load().CoeffModulus_MaxBitCount(degree, security_level, C.byref(bits)))
        return bits.value


class PlainModulus:
    """seal_fhe/src/modulus.rs:252-275"""

    @staticmethod
    def batching(degree: int, bit_size: int) -> Modulus:
        return CoefficientModulus.create(degree, [bit_size])[0]

    @staticmethod
    def raw(val: int) -> Modulus:
        return Modulus(val)


class EncryptionParameters:
    """seal_fhe/src/encryption_parameters.rs:60-200 (BFV only)"""

    def __init__(self, handle):
        self._h = handle

    def get_handle(self):
        return self._h

    def get_poly_modulus_degree(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().EncParams_GetPolyModulusDegree(self._h, C.byref(v)))
        return v.value

    def get_plain_modulus(self) -> Modulus:
        h = C.c_void_p()
        _check(_lib.load().EncParams_GetPlainModulus(self._h, C.byref(h)))
        return Modulus._adopt(h.value)

    def get_coefficient_modulus(self) -> list[Modulus]:
        L = _lib.load()
        n = C.c_uint64()
        _check(L.EncParams_GetCoeffModulus(self._h, C.byref(n), None))
        out = (C.c_void_p * n.value)()
        _check(L.EncParams_GetCoeffModulus(self._h, C.byref(n), out))
        return [Modulus._adopt(out[i]) for i in range(n.value)]

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().EncParams_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class BfvEncryptionParametersBuilder:
    """seal_fhe/src/encryption_parameters.rs:205-330"""

    def __init__(self):
        self._degree = None
        self._coeff = None
        self._plain = None

    def set_poly_modulus_degree(self, degree: int):
        self._degree = degree
        return self

    def set_coefficient_modulus(self, modulus: Sequence[Modulus]):
        self._coeff = list(modulus)
        return self

    def set_plain_modulus(self, modulus: Modulus):
        self._plain = modulus
        return self

    def set_plain_modulus_u64(self, modulus: int):
        self._plain = int(modulus)
        return self

    def build(self) -> EncryptionParameters:
        if self._degree is None:
            raise ValueError("DegreeNotSet")
        if self._coeff is None:
            raise ValueError("CoefficientModulusNotSet")
        if self._plain is None:
            raise ValueError("PlainModulusNotSet")
        L = _lib.load()
        h = C.c_void_p()
        _check(L.EncParams_Create1(1, C.byref(h)))
        p = EncryptionParameters(h)
        _check(L.EncParams_SetPolyModulusDegree(h, self._degree))
        arr = (C.c_void_p * len(self._coeff))(*[m.get_handle() for m in self._coeff])
        _check(L.EncParams_SetCoeffModulus(h, len(self._coeff), arr))
        if isinstance(self._plain, Modulus):
            _check(L.EncParams_SetPlainModulus1(h, self._plain.get_handle()))
        else:
            _check(L.EncParams_SetPlainModulus2(h, self._plain))
        return p


class Context:
    """seal_fhe/src/context.rs:45-115"""

    def __init__(self, params: EncryptionParameters, expand_mod_chain: bool = True, security_level: int = SecurityLevel.TC128):
        self._h = C.c_void_p()
        _check(_lib.load().SEALContext_Create(params.get_handle(), expand_mod_chain, security_level, C.byref(self._h)))
        self._query()

    @classmethod
    def new_insecure(cls, params: EncryptionParameters, expand_mod_chain: bool = True) -> "Context":
        return cls(params, expand_mod_chain, SecurityLevel.NONE)

    @classmethod
    def from_raw(cls, poly_modulus_degree: int, coeff_modulus: Iterable[int], plain_modulus: int) -> "Context":
        """hipbfv extension: build a context from plain integers (no security-level check)."""
        c = cls.__new__(cls)
        cm = np.ascontiguousarray(np.asarray(list(coeff_modulus), dtype=np.uint64))
        c._h = C.c_void_p()
        _check(
            _lib.load().hipbfv_Context_Create(
                poly_modulus_degree, cm.ctypes.data_as(_lib.u64p), cm.size, plain_modulus, C.byref(c._h)
            )
        )
        c._query()
        return c

    def _query(self):
        n, K, KK, t = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(_lib.load().hipbfv_Context_Info(self._h, C.byref(n), C.byref(K), C.byref(KK), C.byref(t)))
        self.poly_modulus_degree, self.K, self.KK, self.plain_modulus = n.value, K.value, KK.value, t.value
        v = C.c_uint64()
        self.key_primes = []
        for i in range(self.KK):
            _check(_lib.load().hipbfv_Context_GetPrime(self._h, i, C.byref(v)))
            self.key_primes.append(v.value)
        cnt, own = C.c_uint64(), C.c_int()
        _check(_lib.load().hipbfv_Context_AuxBase(self._h, C.byref(cnt), None, 0, C.byref(own)))
        buf = (C.c_uint64 * cnt.value)()
        _check(_lib.load().hipbfv_Context_AuxBase(self._h, C.byref(cnt), buf, cnt.value, C.byref(own)))
        self.aux_primes = list(buf)  # B..., m_sk (internal to multiply; see include/hipbfv.h)
        self.aux_fp64 = bool(own.value & 1)
        self.aux_mixed = bool(own.value & 16)  # integer-policy data primes beside FP64-policy auxiliary primes
        self.packed_mul = bool(own.value & 2)  # 48-bit packed intermediates in the split multiply / key switch
        self.packed_ks = bool(own.value & 4)
        self.packed_mul_rows = bool(own.value & 32)  # ... per row: only the rows whose prime is below 2^48 (r04)
        self.packed_ks_rows = bool(own.value & 64)  # the key switch's rows per key prime (r06; packed_ks is set too)
        self.conv_grid = bool(own.value & 8)  # base-conversion sums formed exactly and reduced once (griddot.hpp)

    def get_handle(self):
        return self._h

    def next_level(self) -> "Context":
        """The next level of the modulus-switching chain (one data prime fewer, same special prime): where
        mod_switch_to_next puts its result.  hipbfv extension for the batched API."""
        c = Context.__new__(Context)
        c._h = C.c_void_p()
        _check(_lib.load().hipbfv_Context_NextLevel(self._h, C.byref(c._h)))
        c._query()
        return c

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().SEALContext_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class Plaintext:
    """seal_fhe/src/plaintext_ciphertext.rs:36-300"""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.load().Plaintext_Create1(None, C.byref(self._h)))

    @classmethod
    def from_coefficients(cls, coeffs: Sequence[int]) -> "Plaintext":
        p = cls()
        p.resize(len(coeffs))
        for i, c in enumerate(coeffs):
            if c:
                p.set_coefficient(i, int(c))
        return p

    @classmethod
    def from_hex_string(cls, hex_str: str) -> "Plaintext":
        """SEAL's polynomial string, e.g. "1234x^2 + 4321" (plaintext_ciphertext.rs:180-217)."""
        p = cls.__new__(cls)
        p._h = C.c_void_p()
        _check(_lib.load().Plaintext_Create4(hex_str.encode(), None, C.byref(p._h)))
        return p

    def get_handle(self):
        return self._h

    def resize(self, count: int):
        _check(_lib.load().Plaintext_Resize(self._h, count))

    def len(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().Plaintext_CoeffCount(self._h, C.byref(v)))
        return v.value

    __len__ = len

    def get_coefficient(self, index: int) -> int:
        v = C.c_uint64()
        _check(_lib.load().Plaintext_CoeffAt(self._h, index, C.byref(v)))
        return v.value

    def set_coefficient(self, index: int, value: int):
        _check(_lib.load().Plaintext_SetCoeffAt(self._h, index, value))

    def is_ntt_form(self) -> bool:
        v = C.c_bool()
        _check(_lib.load().Plaintext_IsNTTForm(self._h, C.byref(v)))
        return v.value

    def as_bytes(self, compression: int = 2) -> bytes:
        L = _lib.load()
        size = C.c_int64()
        _check(L.Plaintext_SaveSize(self._h, compression, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        written = C.c_int64()
        _check(L.Plaintext_Save(self._h, buf, size.value, compression, C.byref(written)))
        return buf.raw[: written.value]

    @classmethod
    def from_bytes(cls, ctx: "Context", data: bytes) -> "Plaintext":
        p = cls()
        read = C.c_int64()
        _check(_lib.load().Plaintext_Load(p._h, ctx.get_handle(), data, len(data), C.byref(read)))
        return p

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Plaintext_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class Ciphertext:
    """seal_fhe/src/plaintext_ciphertext.rs:326-504"""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.load().Ciphertext_Create1(None, C.byref(self._h)))

    @classmethod
    def from_array(cls, ctx: Context, data: np.ndarray) -> "Ciphertext":
        """hipbfv extension (stands in for Ciphertext::from_bytes): data = uint64[size][K][N]."""
        data = np.ascontiguousarray(np.asarray(data, dtype=np.uint64))
        assert data.ndim == 3 and data.shape[1] == ctx.K and data.shape[2] == ctx.poly_modulus_degree, data.shape
        c = cls()
        _check(_lib.load().hipbfv_Ciphertext_Assign(c._h, ctx.get_handle(), data.shape[0], data.ctypes.data_as(_lib.u64p)))
        return c

    def to_array(self) -> np.ndarray:
        size, K, n = self.num_polynomials(), self.coeff_modulus_size(), self.poly_modulus_degree()
        out = np.zeros((size, K, n), dtype=np.uint64)
        _check(_lib.load().hipbfv_Ciphertext_Export(self._h, out.ctypes.data_as(_lib.u64p), out.size))
        return out

    def clone(self) -> "Ciphertext":
        c = Ciphertext.__new__(Ciphertext)
        c._h = C.c_void_p()
        _check(_lib.load().Ciphertext_Create2(self._h, C.byref(c._h)))
        return c

    def get_handle(self):
        return self._h

    def num_polynomials(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().Ciphertext_Size(self._h, C.byref(v)))
        return v.value

    def coeff_modulus_size(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().Ciphertext_CoeffModulusSize(self._h, C.byref(v)))
        return v.value

    def poly_modulus_degree(self) -> int:
        v = C.c_uint64()
        _check(_lib.load().Ciphertext_PolyModulusDegree(self._h, C.byref(v)))
        return v.value

    def get_data(self, index: int) -> int:
        v = C.c_uint64()
        _check(_lib.load().Ciphertext_GetDataAt1(self._h, index, C.byref(v)))
        return v.value

    def get_coefficient(self, poly_index: int, coeff_index: int) -> list[int]:
        k = self.coeff_modulus_size()
        buf = (C.c_uint64 * k)()
        _check(_lib.load().Ciphertext_GetDataAt2(self._h, poly_index, coeff_index, buf))
        return [int(x) for x in buf]

    def is_ntt_form(self) -> bool:
        v = C.c_bool()
        _check(_lib.load().Ciphertext_IsNTTForm(self._h, C.byref(v)))
        return v.value

    def as_bytes(self, compression: int = 2) -> bytes:
        """SEAL 4.0 wire format, zstd by default (seal_fhe/src/plaintext_ciphertext.rs:451-477)."""
        L = _lib.load()
        size = C.c_int64()
        _check(L.Ciphertext_SaveSize(self._h, compression, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        written = C.c_int64()
        _check(L.Ciphertext_Save(self._h, buf, size.value, compression, C.byref(written)))
        return buf.raw[: written.value]

    @classmethod
    def from_bytes(cls, ctx: "Context", data: bytes) -> "Ciphertext":
        """seal_fhe/src/plaintext_ciphertext.rs:479-497"""
        c = cls()
        read = C.c_int64()
        _check(_lib.load().Ciphertext_Load(c._h, ctx.get_handle(), data, len(data), C.byref(read)))
        return c

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Ciphertext_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class _KSwitchKeys:
    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.load().KSwitchKeys_Create1(C.byref(self._h)))

    def get_handle(self):
        return self._h

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().KSwitchKeys_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


def _keys_as_bytes(self, compression: int = 2) -> bytes:
    L = _lib.load()
    size = C.c_int64()
    _check(L.KSwitchKeys_SaveSize(self._h, compression, C.byref(size)))
    buf = C.create_string_buffer(size.value)
    written = C.c_int64()
    _check(L.KSwitchKeys_Save(self._h, buf, size.value, compression, C.byref(written)))
    return buf.raw[: written.value]


def _keys_from_bytes(cls, ctx: "Context", data: bytes):
    k = cls()
    read = C.c_int64()
    _check(_lib.load().KSwitchKeys_Load(k._h, ctx.get_handle(), data, len(data), C.byref(read)))
    return k


def _keys_to_array(self, ctx: "Context", index: int = 0) -> np.ndarray:
    """hipbfv extension: one key-switching key, uint64[K][2][K+1][N]; index 0 = relin, (elt-1)/2 = Galois."""
    out = np.empty((ctx.K, 2, ctx.KK, ctx.poly_modulus_degree), dtype=np.uint64)
    _check(_lib.load().hipbfv_KSwitchKeys_Read(self._h, index, out.ctypes.data_as(_lib.u64p)))
    return out


def _keys_has(self, index: int) -> bool:
    present = C.c_bool()
    _check(_lib.load().hipbfv_KSwitchKeys_Has(self._h, index, C.byref(present)))
    return present.value


_KSwitchKeys.to_array = _keys_to_array
_KSwitchKeys.has_key = _keys_has
_KSwitchKeys.as_bytes = _keys_as_bytes
_KSwitchKeys.from_bytes = classmethod(_keys_from_bytes)


class RelinearizationKeys(_KSwitchKeys):
    """seal_fhe/src/key_generator.rs:467-575"""

    @classmethod
    def from_array(cls, ctx: Context, key: np.ndarray) -> "RelinearizationKeys":
        """hipbfv extension (stands in for RelinearizationKeys::from_bytes): key = uint64[K][2][K+1][N]."""
        key = np.ascontiguousarray(np.asarray(key, dtype=np.uint64))
        assert key.shape == (ctx.K, 2, ctx.KK, ctx.poly_modulus_degree), key.shape
        k = cls()
        _check(_lib.load().hipbfv_KSwitchKeys_AssignRelin(k._h, ctx.get_handle(), key.ctypes.data_as(_lib.u64p)))
        return k


class GaloisKeys(_KSwitchKeys):
    """seal_fhe/src/key_generator.rs:631-729"""

    @classmethod
    def from_arrays(cls, ctx: Context, keys: dict[int, np.ndarray]) -> "GaloisKeys":
        """hipbfv extension: keys maps Galois element -> uint64[K][2][K+1][N]."""
        g = cls()
        for elt, key in keys.items():
            key = np.ascontiguousarray(np.asarray(key, dtype=np.uint64))
            assert key.shape == (ctx.K, 2, ctx.KK, ctx.poly_modulus_degree), key.shape
            _check(_lib.load().hipbfv_KSwitchKeys_AssignGalois(g._h, ctx.get_handle(), int(elt), key.ctypes.data_as(_lib.u64p)))
        return g


class BFVEvaluator:
    """Mirror of `impl Evaluator for BFVEvaluator` (seal_fhe/src/bfv_evaluator.rs:12-248,
    evaluator_base.rs:55-407).  Out-of-place methods allocate an empty Ciphertext and let the callee
    size it; `_inplace` methods pass the operand handle as the destination, exactly like the crate."""

    def __init__(self, ctx: Context):
        self._ctx = ctx
        self._h = C.c_void_p()
        _check(_lib.load().Evaluator_Create(ctx.get_handle(), C.byref(self._h)))  # evaluator_base.rs:74-80

    def get_handle(self):
        return self._h

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Evaluator_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None

    # -- negate / add / sub (evaluator_base.rs:89-182)
    def negate_inplace(self, a: Ciphertext) -> None:
        _check(_lib.load().Evaluator_Negate(self._h, a._h, a._h))

    def negate(self, a: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Negate(self._h, a._h, out._h))
        return out

    def add_inplace(self, a: Ciphertext, b: Ciphertext) -> None:
        _check(_lib.load().Evaluator_Add(self._h, a._h, b._h, a._h))

    def add(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Add(self._h, a._h, b._h, out._h))
        return out

    def add_many(self, a: Sequence[Ciphertext]) -> Ciphertext:
        out = Ciphertext()
        arr = (C.c_void_p * len(a))(*[x._h for x in a])
        _check(_lib.load().Evaluator_AddMany(self._h, len(a), arr, out._h))
        return out

    def sub_inplace(self, a: Ciphertext, b: Ciphertext) -> None:
        _check(_lib.load().Evaluator_Sub(self._h, a._h, b._h, a._h))

    def sub(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Sub(self._h, a._h, b._h, out._h))
        return out

    # -- multiply / square (evaluator_base.rs:184-260)
    def multiply_inplace(self, a: Ciphertext, b: Ciphertext) -> None:
        _check(_lib.load().Evaluator_Multiply(self._h, a._h, b._h, a._h, None))

    def multiply(self, a: Ciphertext, b: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Multiply(self._h, a._h, b._h, out._h, None))
        return out

    def multiply_many(self, a: Sequence[Ciphertext], relin_keys: RelinearizationKeys) -> Ciphertext:
        out = Ciphertext()
        arr = (C.c_void_p * len(a))(*[x._h for x in a])
        _check(_lib.load().Evaluator_MultiplyMany(self._h, len(a), arr, relin_keys._h, out._h, None))
        return out

    def square_inplace(self, a: Ciphertext) -> None:
        _check(_lib.load().Evaluator_Square(self._h, a._h, a._h, None))

    def square(self, a: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Square(self._h, a._h, out._h, None))
        return out

    def exponentiate(self, a: Ciphertext, exponent: int, relin_keys: RelinearizationKeys) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Exponentiate(self._h, a._h, exponent, relin_keys._h, out._h, None))
        return out

    def exponentiate_inplace(self, a: Ciphertext, exponent: int, relin_keys: RelinearizationKeys) -> None:
        _check(_lib.load().Evaluator_Exponentiate(self._h, a._h, exponent, relin_keys._h, a._h, None))

    # -- plaintext operands (evaluator_base.rs:320-404)
    def add_plain(self, a: Ciphertext, b: Plaintext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_AddPlain(self._h, a._h, b._h, out._h))
        return out

    def add_plain_inplace(self, a: Ciphertext, b: Plaintext) -> None:
        _check(_lib.load().Evaluator_AddPlain(self._h, a._h, b._h, a._h))

    def sub_plain(self, a: Ciphertext, b: Plaintext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_SubPlain(self._h, a._h, b._h, out._h))
        return out

    def sub_plain_inplace(self, a: Ciphertext, b: Plaintext) -> None:
        _check(_lib.load().Evaluator_SubPlain(self._h, a._h, b._h, a._h))

    def multiply_plain(self, a: Ciphertext, b: Plaintext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_MultiplyPlain(self._h, a._h, b._h, out._h, None))
        return out

    def multiply_plain_inplace(self, a: Ciphertext, b: Plaintext) -> None:
        _check(_lib.load().Evaluator_MultiplyPlain(self._h, a._h, b._h, a._h, None))

    # -- modulus switching (evaluator.rs:84-157)
    def mod_switch_to_next(self, a: Ciphertext) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_ModSwitchToNext1(self._h, a._h, out._h, None))
        return out

    def mod_switch_to_next_inplace(self, a: Ciphertext) -> None:
        _check(_lib.load().Evaluator_ModSwitchToNext1(self._h, a._h, a._h, None))

    def mod_switch_to_next_plaintext(self, p: "Plaintext") -> "Plaintext":
        out = Plaintext()
        _check(_lib.load().Evaluator_ModSwitchToNext2(self._h, p.get_handle(), out.get_handle()))
        return out

    def mod_switch_to_next_inplace_plaintext(self, p: "Plaintext") -> None:
        _check(_lib.load().Evaluator_ModSwitchToNext2(self._h, p.get_handle(), p.get_handle()))

    # -- key switching (bfv_evaluator.rs:143-247)
    def relinearize_inplace(self, a: Ciphertext, relin_keys: RelinearizationKeys) -> None:
        _check(_lib.load().Evaluator_Relinearize(self._h, a._h, relin_keys._h, a._h, None))

    def relinearize(self, a: Ciphertext, relin_keys: RelinearizationKeys) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_Relinearize(self._h, a._h, relin_keys._h, out._h, None))
        return out

    def rotate_rows(self, a: Ciphertext, steps: int, galois_keys: GaloisKeys) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_RotateRows(self._h, a._h, steps, galois_keys._h, out._h, None))
        return out

    def rotate_rows_inplace(self, a: Ciphertext, steps: int, galois_keys: GaloisKeys) -> None:
        _check(_lib.load().Evaluator_RotateRows(self._h, a._h, steps, galois_keys._h, a._h, None))

    def rotate_columns(self, a: Ciphertext, galois_keys: GaloisKeys) -> Ciphertext:
        out = Ciphertext()
        _check(_lib.load().Evaluator_RotateColumns(self._h, a._h, galois_keys._h, out._h, None))
        return out

    def rotate_columns_inplace(self, a: Ciphertext, galois_keys: GaloisKeys) -> None:
        _check(_lib.load().Evaluator_RotateColumns(self._h, a._h, galois_keys._h, a._h, None))

# ---- the steps either side of the evaluator --------------------------------------------------
class _AsymKey:
    """SecretKey / PublicKey handles (seal_fhe/src/key_generator.rs:200-430): made by KeyGenerator, loaded from
    SEAL's wire format, or assigned from raw residues."""

    _prefix = ""
    _polys = 1

    def __init__(self):
        self._h = C.c_void_p()
        _check(getattr(_lib.load(), self._prefix + "_Create1")(C.byref(self._h)))

    def get_handle(self):
        return self._h

    @classmethod
    def from_array(cls, ctx: "Context", data: np.ndarray):
        """data: uint64[(2,) K+1, N] key-level NTT-form residues (SEAL's in-memory layout)."""
        k = cls()
        arr = np.ascontiguousarray(np.asarray(data, dtype=np.uint64))
        assert arr.size == cls._polys * ctx.KK * ctx.poly_modulus_degree, arr.shape
        _check(getattr(_lib.load(), "hipbfv_" + cls._prefix + "_Assign")(k._h, ctx.get_handle(), arr.ctypes.data_as(_lib.u64p)))
        return k

    def to_array(self, ctx: "Context") -> np.ndarray:
        """hipbfv extension: the key-level NTT-form residues, uint64[(2,) K+1, N]."""
        shape = (ctx.KK, ctx.poly_modulus_degree) if self._polys == 1 else (self._polys, ctx.KK, ctx.poly_modulus_degree)
        out = np.empty(shape, dtype=np.uint64)
        _check(getattr(_lib.load(), "hipbfv_" + self._prefix + "_Read")(self._h, out.ctypes.data_as(_lib.u64p)))
        return out

    def as_bytes(self, compression: int = 2) -> bytes:
        L = _lib.load()
        size = C.c_int64()
        _check(getattr(L, self._prefix + "_SaveSize")(self._h, compression, C.byref(size)))
        buf = C.create_string_buffer(size.value)
        written = C.c_int64()
        _check(getattr(L, self._prefix + "_Save")(self._h, buf, size.value, compression, C.byref(written)))
        return buf.raw[: written.value]

    @classmethod
    def from_bytes(cls, ctx: "Context", data: bytes):
        k = cls()
        read = C.c_int64()
        _check(getattr(_lib.load(), cls._prefix + "_Load")(k._h, ctx.get_handle(), data, len(data), C.byref(read)))
        return k

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                getattr(_load(), self._prefix + "_Destroy")(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class SecretKey(_AsymKey):
    _prefix, _polys = "SecretKey", 1


class PublicKey(_AsymKey):
    _prefix, _polys = "PublicKey", 2


class KeyGenerator:
    """seal_fhe/src/key_generator.rs:20-200.  Keys are sampled and assembled on the device."""

    def __init__(self, ctx: "Context", secret_key: SecretKey | None = None, seed: int | None = None):
        self._ctx = ctx
        self._h = C.c_void_p()
        L = _lib.load()
        if secret_key is None:
            if seed is not None:  # hipbfv extension: reproducible secret and keys (tests)
                _check(L.hipbfv_KeyGenerator_CreateSeeded(ctx.get_handle(), seed, C.byref(self._h)))
            else:
                _check(L.KeyGenerator_Create1(ctx.get_handle(), C.byref(self._h)))
        else:
            _check(L.KeyGenerator_Create2(ctx.get_handle(), secret_key.get_handle(), C.byref(self._h)))
            if seed is not None:
                _check(L.hipbfv_KeyGenerator_SetSeed(self._h, seed))

    @classmethod
    def new_from_secret_key(cls, ctx: "Context", secret_key: SecretKey) -> "KeyGenerator":
        return cls(ctx, secret_key)

    def set_seed(self, seed: int) -> None:
        """hipbfv extension: make the keys created afterwards reproducible (tests)."""
        _check(_lib.load().hipbfv_KeyGenerator_SetSeed(self._h, seed))

    def _adopt(self, cls):
        k = cls.__new__(cls)
        k._h = C.c_void_p()
        return k

    def secret_key(self) -> SecretKey:
        k = self._adopt(SecretKey)
        _check(_lib.load().KeyGenerator_SecretKey(self._h, C.byref(k._h)))
        return k

    def create_public_key(self) -> PublicKey:
        k = self._adopt(PublicKey)
        _check(_lib.load().KeyGenerator_CreatePublicKey(self._h, False, C.byref(k._h)))
        return k

    def create_relinearization_keys(self) -> RelinearizationKeys:
        k = self._adopt(RelinearizationKeys)
        _check(_lib.load().KeyGenerator_CreateRelinKeys(self._h, False, C.byref(k._h)))
        return k

    def create_galois_keys(self, galois_elts: Sequence[int] | None = None, steps: Sequence[int] | None = None) -> GaloisKeys:
        """All keys (SEAL create_galois_keys()), the keys of the given Galois elements, or -- `steps` -- of the given
        rotation steps (SEAL create_galois_keys(steps); step 0 = the column rotation)."""
        k = self._adopt(GaloisKeys)
        L = _lib.load()
        if steps is not None:
            arr = (C.c_int * len(steps))(*[int(e) for e in steps])
            _check(L.KeyGenerator_CreateGaloisKeysFromSteps(self._h, len(steps), arr, False, C.byref(k._h)))
        elif galois_elts is None:
            _check(L.KeyGenerator_CreateGaloisKeysAll(self._h, False, C.byref(k._h)))
        else:
            arr = (C.c_uint32 * len(galois_elts))(*[int(e) for e in galois_elts])
            _check(L.KeyGenerator_CreateGaloisKeysFromElts(self._h, len(galois_elts), arr, False, C.byref(k._h)))
        return k

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().KeyGenerator_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class BFVEncoder:
    """seal_fhe/src/encoder.rs:30-215 (BatchEncoder)."""

    def __init__(self, ctx: "Context"):
        self._ctx = ctx
        self._h = C.c_void_p()
        _check(_lib.load().BatchEncoder_Create(ctx.get_handle(), C.byref(self._h)))

    def get_slot_count(self) -> int:
        n = C.c_uint64()
        _check(_lib.load().BatchEncoder_GetSlotCount(self._h, C.byref(n)))
        return n.value

    def encode_unsigned(self, data: Sequence[int]) -> Plaintext:
        p = Plaintext()
        arr = (C.c_uint64 * len(data))(*[int(v) for v in data])
        _check(_lib.load().BatchEncoder_Encode1(self._h, len(data), arr, p.get_handle()))
        return p

    def encode_signed(self, data: Sequence[int]) -> Plaintext:
        p = Plaintext()
        arr = (C.c_int64 * len(data))(*[int(v) for v in data])
        _check(_lib.load().BatchEncoder_Encode2(self._h, len(data), arr, p.get_handle()))
        return p

    def decode_unsigned(self, plaintext: Plaintext) -> list[int]:
        n = self.get_slot_count()
        out = (C.c_uint64 * n)()
        size = C.c_uint64()
        _check(_lib.load().BatchEncoder_Decode1(self._h, plaintext.get_handle(), C.byref(size), out, None))
        return list(out[: size.value])

    def decode_signed(self, plaintext: Plaintext) -> list[int]:
        n = self.get_slot_count()
        out = (C.c_int64 * n)()
        size = C.c_uint64()
        _check(_lib.load().BatchEncoder_Decode2(self._h, plaintext.get_handle(), C.byref(size), out, None))
        return list(out[: size.value])

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().BatchEncoder_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class Decryptor:
    """seal_fhe/src/encryptor_decryptor.rs:596-690."""

    def __init__(self, ctx: "Context", secret_key: SecretKey):
        self._keep = (ctx, secret_key)
        self._h = C.c_void_p()
        _check(_lib.load().Decryptor_Create(ctx.get_handle(), secret_key.get_handle(), C.byref(self._h)))

    def decrypt(self, ciphertext: "Ciphertext") -> Plaintext:
        p = Plaintext()
        _check(_lib.load().Decryptor_Decrypt(self._h, ciphertext.get_handle(), p.get_handle()))
        return p

    def invariant_noise_budget(self, ciphertext: "Ciphertext") -> int:
        b = C.c_int()
        _check(_lib.load().Decryptor_InvariantNoiseBudget(self._h, ciphertext.get_handle(), C.byref(b)))
        return b.value

    def invariant_noise(self, ciphertext: "Ciphertext") -> float:
        """|[t * ct(s)]_q|_inf / q (encryptor_decryptor.rs:660-683); decryption is correct below 1/2."""
        v = C.c_double()
        _check(_lib.load().Decryptor_InvariantNoise(self._h, ciphertext.get_handle(), C.byref(v)))
        return v.value

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Decryptor_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class PolynomialArray:
    """seal_fhe/src/data_structures.rs:17-304 (the Sunscreen fork's export type, consumed by logproof)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.load().PolynomialArray_Create(None, C.byref(self._h)))

    @classmethod
    def _adopt(cls, handle) -> "PolynomialArray":
        a = cls.__new__(cls)
        a._h = handle
        return a

    @classmethod
    def new_from_ciphertext(cls, ctx: "Context", ciphertext: "Ciphertext") -> "PolynomialArray":
        h = C.c_void_p()
        _check(_lib.load().PolynomialArray_CreateFromCiphertext(None, ctx.get_handle(), ciphertext.get_handle(), C.byref(h)))
        return cls._adopt(h)

    @classmethod
    def new_from_public_key(cls, ctx: "Context", public_key: PublicKey) -> "PolynomialArray":
        h = C.c_void_p()
        _check(_lib.load().PolynomialArray_CreateFromPublicKey(None, ctx.get_handle(), public_key.get_handle(), C.byref(h)))
        return cls._adopt(h)

    @classmethod
    def new_from_secret_key(cls, ctx: "Context", secret_key: SecretKey) -> "PolynomialArray":
        h = C.c_void_p()
        _check(_lib.load().PolynomialArray_CreateFromSecretKey(None, ctx.get_handle(), secret_key.get_handle(), C.byref(h)))
        return cls._adopt(h)

    def get_handle(self):
        return self._h

    def clone(self) -> "PolynomialArray":
        h = C.c_void_p()
        _check(_lib.load().PolynomialArray_Copy(self._h, C.byref(h)))
        return self._adopt(h)

    def _flag(self, name: str) -> bool:
        v = C.c_bool()
        _check(getattr(_lib.load(), "PolynomialArray_" + name)(self._h, C.byref(v)))
        return v.value

    def _size(self, name: str) -> int:
        v = C.c_uint64()
        _check(getattr(_lib.load(), "PolynomialArray_" + name)(self._h, C.byref(v)))
        return v.value

    def is_reserved(self) -> bool:
        return self._flag("IsReserved")

    def is_rns(self) -> bool:
        return self._flag("IsRns")

    def is_multiprecision(self) -> bool:
        return not self.is_rns()

    def to_rns(self) -> None:
        _check(_lib.load().PolynomialArray_ToRns(self._h))

    def to_multiprecision(self) -> None:
        _check(_lib.load().PolynomialArray_ToMultiprecision(self._h))

    def num_polynomials(self) -> int:
        return self._size("PolySize")

    def poly_modulus_degree(self) -> int:
        return self._size("PolyModulusDegree")

    def coeff_modulus_size(self) -> int:
        return self._size("CoeffModulusSize")

    def as_u64s(self) -> np.ndarray:
        out = np.empty(self._size("ExportSize"), dtype=np.uint64)
        _check(_lib.load().PolynomialArray_PerformExport(self._h, out.ctypes.data_as(_lib.u64p)))
        return out

    def as_rns_u64s(self) -> np.ndarray:
        """[poly][rns][coeff]; the array keeps its current format (data_structures.rs:226-243)."""
        was_mp = self.is_reserved() and self.is_multiprecision()
        if was_mp:
            self.to_rns()
        out = self.as_u64s()
        if was_mp:
            self.to_multiprecision()
        return out

    def as_multiprecision_u64s(self) -> np.ndarray:
        """[poly][coeff][limb], least-significant limb first (data_structures.rs:198-224)."""
        was_rns = self.is_reserved() and self.is_rns()
        if was_rns:
            self.to_multiprecision()
        out = self.as_u64s()
        if was_rns:
            self.to_rns()
        return out

    def drop_modulus(self) -> "PolynomialArray":
        if self.coeff_modulus_size() == 1:
            raise ValueError("ModulusChainTooSmall")
        h = C.c_void_p()
        _check(_lib.load().PolynomialArray_Drop(self._h, C.byref(h)))
        return self._adopt(h)

    def __eq__(self, other):
        return isinstance(other, PolynomialArray) and np.array_equal(self.as_rns_u64s(), other.as_rns_u64s())

    __hash__ = None

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().PolynomialArray_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None


class Encryptor:
    """seal_fhe/src/encryptor_decryptor.rs:140-600: public-key, secret-key or both (`with_public_key`,
    `with_secret_key`, `with_public_and_secret_key`)."""

    def __init__(self, ctx: "Context", public_key: PublicKey | None = None, seed: int | None = None, secret_key: SecretKey | None = None):
        self._keep = (ctx, public_key, secret_key)
        self._h = C.c_void_p()
        _check(_lib.load().Encryptor_Create(ctx.get_handle(), public_key.get_handle() if public_key else None,
                                            secret_key.get_handle() if secret_key else None, C.byref(self._h)))
        if seed is not None:
            _check(_lib.load().hipbfv_Encryptor_SetSeed(self._h, seed))

    @classmethod
    def with_public_and_secret_key(cls, ctx: "Context", public_key: PublicKey, secret_key: SecretKey) -> "Encryptor":
        return cls(ctx, public_key, secret_key=secret_key)

    @classmethod
    def with_secret_key(cls, ctx: "Context", secret_key: SecretKey) -> "Encryptor":
        return cls(ctx, None, secret_key=secret_key)

    def encrypt(self, plaintext: Plaintext) -> "Ciphertext":
        c = Ciphertext()
        _check(_lib.load().Encryptor_Encrypt(self._h, plaintext.get_handle(), c.get_handle(), None))
        return c

    @staticmethod
    def _seed_words(seed: Sequence[int]):
        assert len(seed) == 8
        return (C.c_uint64 * 8)(*[int(w) for w in seed])

    def encrypt_return_components(self, plaintext: Plaintext, seed: Sequence[int] | None = None, disable_special_modulus: bool = True):
        """-> (ciphertext, u, e, r); with `seed` ([u64; 8]) it is the fork's encrypt_return_components_deterministic."""
        c, u, e, r = Ciphertext(), PolynomialArray(), PolynomialArray(), Plaintext()
        L = _lib.load()
        if seed is None:
            _check(L.Encryptor_EncryptReturnComponents(self._h, plaintext.get_handle(), disable_special_modulus, c.get_handle(),
                                                       u.get_handle(), e.get_handle(), r.get_handle(), None))
        else:
            _check(L.Encryptor_EncryptReturnComponentsSetSeed(self._h, plaintext.get_handle(), disable_special_modulus, c.get_handle(),
                                                              u.get_handle(), e.get_handle(), r.get_handle(), self._seed_words(seed), None))
        return c, u, e, r

    def encrypt_deterministic(self, plaintext: Plaintext, seed: Sequence[int]) -> "Ciphertext":
        return self.encrypt_return_components(plaintext, seed, disable_special_modulus=False)[0]

    def encrypt_symmetric(self, plaintext: Plaintext) -> "Ciphertext":
        c = Ciphertext()
        _check(_lib.load().Encryptor_EncryptSymmetric(self._h, plaintext.get_handle(), False, c.get_handle(), None))
        return c

    def encrypt_symmetric_return_components(self, plaintext: Plaintext, seed: Sequence[int] | None = None):
        """-> (ciphertext, e, r)."""
        c, e, r = Ciphertext(), PolynomialArray(), Plaintext()
        L = _lib.load()
        if seed is None:
            _check(L.Encryptor_EncryptSymmetricReturnComponents(self._h, plaintext.get_handle(), c.get_handle(), e.get_handle(),
                                                                r.get_handle(), None))
        else:
            _check(L.Encryptor_EncryptSymmetricReturnComponentsSetSeed(self._h, plaintext.get_handle(), c.get_handle(), e.get_handle(),
                                                                       r.get_handle(), self._seed_words(seed), None))
        return c, e, r

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().Encryptor_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None
