"""Batch executor for compiled FHE program graphs (host mirror of csrc/program.cpp).

The reference executes a compiled `FheProgram` with `run_program_unchecked`
(sunscreen_runtime/src/run.rs:100-357): one evaluator call per node for ONE set of inputs.
`FheProgram.run` executes the same graph over a batch of independent input sets on the GPU.
Graphs are built node by node (the reference's tests do the same, run.rs:595-881) or loaded from the
serde JSON form of `FheProgram`.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Sequence

import torch

from . import _lib
from .batch import BatchEvaluator, _ptr, _stream
from .seal import GaloisKeys, RelinearizationKeys, _check

OPS = [
    "ShiftLeft", "ShiftRight", "SwapRows", "Relinearize", "Multiply", "MultiplyPlaintext", "Add", "AddPlaintext",
    "Negate", "Sub", "SubPlaintext", "InputCiphertext", "InputPlaintext", "Literal", "OutputCiphertext",
]
EDGES = {"Left": 0, "Right": 1, "Unary": 2}


def encode_plaintext_literal(n: int, key_primes: Sequence[int], t: int, seal_plaintext_bytes: bytes, security_level: int = 0) -> bytes:
    """bincode 1.x (little-endian, fixed-width, u64 lengths, u32 variant indices) of
    InnerPlaintext::Seal(vec![WithContext { params: Params { lattice_dimension, coeff_modulus, plain_modulus,
    scheme_type: Bfv, security_level }, data: <SEAL-serialised Plaintext> }]) -- sunscreen_runtime/src/lib.rs:37-42,
    serialization.rs:16-60, metadata.rs:72-97.  security_level is the variant index (0 = TC128)."""
    import struct

    out = struct.pack("<IQ", 0, 1)
    out += struct.pack("<QQ", n, len(key_primes)) + struct.pack("<%dQ" % len(key_primes), *key_primes)
    out += struct.pack("<QII", t, 0, security_level)
    out += struct.pack("<Q", len(seal_plaintext_bytes)) + bytes(seal_plaintext_bytes)
    return out


class TransformedPlaintext:
    """A plaintext argument already lifted to the data primes and transformed (BatchEvaluator.plain_to_ntt): int64[K,N]
    shared by the batch or int64[batch,K,N].  Only MultiplyPlaintext nodes may consume it (ProgramInput kind 2)."""

    def __init__(self, tensor: torch.Tensor):
        assert tensor.is_cuda and tensor.dtype == torch.int64 and tensor.is_contiguous()
        self.tensor = tensor


class FheProgram:
    def __init__(self):
        self._h = C.c_void_p()
        _check(_lib.load().hipbfv_Program_Create(C.byref(self._h)))
        self.nodes: list[tuple[str, int]] = []
        self.edges: list[tuple[int, int, str]] = []

    def __del__(self, _load=_lib.load):  # the default argument outlives the module globals at interpreter shutdown
        if getattr(self, "_h", None):
            try:
                _load().hipbfv_Program_Destroy(self._h)
            except Exception:  # finalisers never raise
                pass
            self._h = None

    # -- construction (names follow sunscreen_fhe_program::FheProgramTrait, lib.rs:170-250)
    def _node(self, op: str, arg: int = 0) -> int:
        nid = C.c_uint32()
        _check(_lib.load().hipbfv_Program_AddNode(self._h, OPS.index(op), arg, C.byref(nid)))
        self.nodes.append((op, arg))
        return nid.value

    def _edge(self, src: int, dst: int, kind: str) -> None:
        _check(_lib.load().hipbfv_Program_AddEdge(self._h, src, dst, EDGES[kind]))
        self.edges.append((src, dst, kind))

    def _binary(self, op: str, left: int, right: int) -> int:
        n = self._node(op)
        self._edge(left, n, "Left")
        self._edge(right, n, "Right")
        return n

    def _unary(self, op: str, x: int) -> int:
        n = self._node(op)
        self._edge(x, n, "Unary")
        return n

    def append_input_ciphertext(self, index: int) -> int:
        return self._node("InputCiphertext", index)

    def append_input_plaintext(self, index: int) -> int:
        return self._node("InputPlaintext", index)

    def append_input_literal(self, value: int) -> int:
        return self._node("Literal", value)

    def append_plaintext_literal(self, inner_plaintext_bytes: bytes) -> int:
        """Literal::Plaintext(bytes): bincode of InnerPlaintext::Seal([WithContext{Params, SEAL plaintext}]) as
        the compiler stores it (sunscreen/src/fhe/mod.rs:370-376); see `encode_plaintext_literal`."""
        nid = C.c_uint32()
        raw = bytes(inner_plaintext_bytes)
        _check(_lib.load().hipbfv_Program_AddPlaintextLiteral(self._h, raw, len(raw), C.byref(nid)))
        self.nodes.append(("Literal", {"Plaintext": list(raw)}))
        return nid.value

    def append_add(self, a, b):
        return self._binary("Add", a, b)

    def append_sub(self, a, b):
        return self._binary("Sub", a, b)

    def append_multiply(self, a, b):
        return self._binary("Multiply", a, b)

    def append_add_plaintext(self, a, b):
        return self._binary("AddPlaintext", a, b)

    def append_sub_plaintext(self, a, b):
        return self._binary("SubPlaintext", a, b)

    def append_multiply_plaintext(self, a, b):
        return self._binary("MultiplyPlaintext", a, b)

    def append_rotate_left(self, a, literal):
        return self._binary("ShiftLeft", a, literal)

    def append_rotate_right(self, a, literal):
        return self._binary("ShiftRight", a, literal)

    def append_negate(self, a):
        return self._unary("Negate", a)

    def append_swap_rows(self, a):
        return self._unary("SwapRows", a)

    def append_relinearize(self, a):
        return self._unary("Relinearize", a)

    def append_output_ciphertext(self, a):
        return self._unary("OutputCiphertext", a)

    # -- serde JSON form of FheProgram
    @classmethod
    def from_json(cls, text: str) -> "FheProgram":
        p = cls()
        raw = text.encode()
        _check(_lib.load().hipbfv_Program_LoadJson(p._h, raw, len(raw)))
        g = json.loads(text)
        g = g.get("graph", g)
        for nd in g["nodes"]:
            op = nd["operation"]
            if isinstance(op, str):
                p.nodes.append((op, 0))
            else:
                (name, payload), = op.items()
                p.nodes.append((name, payload.get("U64", payload) if isinstance(payload, dict) else payload))
        p.edges = [tuple(e) for e in g["edges"]]
        return p

    def to_json(self) -> str:
        def enc(op, arg):
            if op in ("InputCiphertext", "InputPlaintext"):
                return {op: arg}
            if op == "Literal":
                return {"Literal": arg if isinstance(arg, dict) else {"U64": arg}}
            return op

        return json.dumps(
            {
                "graph": {
                    "nodes": [{"operation": enc(op, arg)} for op, arg in self.nodes],
                    "node_holes": [],
                    "edge_property": "directed",
                    "edges": [list(e) for e in self.edges],
                },
                "data": "Bfv",
            }
        )

    def describe(self) -> list[str]:
        """The schedule `run` follows: one line per step (kind, members, ...)."""
        need = C.c_uint64()
        _check(_lib.load().hipbfv_Program_Describe(self._h, None, 0, C.byref(need)))
        buf = C.create_string_buffer(need.value)
        _check(_lib.load().hipbfv_Program_Describe(self._h, buf, need.value, C.byref(need)))
        return buf.value.decode().splitlines()

    def num_outputs(self) -> int:
        n = C.c_uint64()
        _check(_lib.load().hipbfv_Program_NumOutputs(self._h, C.byref(n)))
        return n.value

    def run(self, ev, inputs, relin_keys=None, galois_keys=None, key_index=None) -> list[torch.Tensor]:
        return self.prepare(ev, inputs, relin_keys, galois_keys, key_index)()

    def prepare(
        self,
        ev: BatchEvaluator,
        inputs: Sequence[torch.Tensor],
        relin_keys: RelinearizationKeys | None = None,
        galois_keys: GaloisKeys | None = None,
        key_index=None,
    ):
        """Bind the arguments once and return a callable that runs the program on them (the argument tables are built here:
        a graph with tens of thousands of arguments -- examples/pir's database -- is run many times on the same buffers)."""
        """key_index (optional, `batch` host integers): one key set per client -- relin_keys / galois_keys are then SEQUENCES of key
        objects (an entry may be None) and input set i runs with relin_keys[key_index[i]], galois_keys[key_index[i]]
        (hipbfv_Program_RunKeys; the reference passes the keys per call, sunscreen_runtime/src/run.rs:100-105)."""
        """inputs[i]: int64[batch,2,K,N] ciphertext batch, int64[batch,N] / int64[N] plaintext(s) in coefficient form, or a
        `TransformedPlaintext` (int64[batch,K,N] / int64[K,N] from BatchEvaluator.plain_to_ntt: static data transformed once)."""
        batch = None
        for t in inputs:
            if not isinstance(t, TransformedPlaintext) and t.dim() == 4:
                batch = t.shape[0]
        assert batch is not None, "need at least one ciphertext argument"
        n_in = len(inputs)
        kinds = (C.c_uint32 * n_in)()
        ptrs = (C.c_void_p * n_in)()
        strides = (C.c_uint64 * n_in)()
        for i, t in enumerate(inputs):
            if isinstance(t, TransformedPlaintext):
                t = t.tensor
                assert t.shape[-2:] == (ev.K, ev.n) and (t.dim() == 2 or (t.dim() == 3 and t.shape[0] in (1, batch)))
                kinds[i] = 2
                strides[i] = 0 if t.dim() == 2 or t.shape[0] == 1 else ev.K * ev.n
            elif t.dim() == 4:
                assert t.shape[0] == batch and t.shape[1] == 2
                kinds[i], strides[i] = 0, 0
            else:
                kinds[i] = 1
                strides[i] = 0 if t.dim() == 1 or t.shape[0] == 1 else ev.n
            ptrs[i] = _ptr(t)
        n_out = self.num_outputs()
        dev = next(t for t in inputs if not isinstance(t, TransformedPlaintext)).device
        keep = list(inputs)  # the tables hold raw addresses: the tensors must outlive the callable
        if key_index is not None:
            import numpy as np

            idx = np.ascontiguousarray(np.asarray(key_index, dtype=np.uint32))
            assert idx.shape == (batch,), (idx.shape, batch)
            nsets = max(len(relin_keys) if relin_keys is not None else 0, len(galois_keys) if galois_keys is not None else 0)
            assert nsets > 0, "key_index needs sequences of key sets"

            def handles(seq):
                seq = list(seq) if seq is not None else []
                seq += [None] * (nsets - len(seq))
                return (C.c_void_p * nsets)(*[k.get_handle() if k is not None else None for k in seq])

            rks, gks = handles(relin_keys), handles(galois_keys)
            fnk = _lib.load().hipbfv_Program_RunKeys
            keep.append((idx, relin_keys, galois_keys))

            def call_keys() -> list[torch.Tensor]:
                outs = [torch.empty((batch, 2, ev.K, ev.n), dtype=torch.int64, device=dev) for _ in range(n_out)]
                optrs = (C.c_void_p * n_out)(*[_ptr(o) for o in outs])
                _check(fnk(self._h, ev._h, batch, n_in, kinds, ptrs, strides, nsets, rks, gks, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                           n_out, optrs, _stream()))
                return outs

            return call_keys
        fn = _lib.load().hipbfv_Program_Run
        rk = relin_keys.get_handle() if relin_keys is not None else None
        gk = galois_keys.get_handle() if galois_keys is not None else None

        def call() -> list[torch.Tensor]:
            assert keep is not None
            outs = [torch.empty((batch, 2, ev.K, ev.n), dtype=torch.int64, device=dev) for _ in range(n_out)]
            optrs = (C.c_void_p * n_out)(*[_ptr(o) for o in outs])
            _check(fn(self._h, ev._h, batch, n_in, kinds, ptrs, strides, rk, gk, n_out, optrs, _stream()))
            return outs

        return call
