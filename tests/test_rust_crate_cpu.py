"""rust/hip_bfv is written but cannot be compiled here (no rustc): keep it honest mechanically.  Every `bindgen::X(...)`
call in the crate must name a function include/hipbfv.h declares, with the declared number of arguments, and the crate
must implement every method of the reference's `trait Evaluator` (seal_fhe/src/evaluator.rs:7-280; the list below is the
contract, generated from the reference by name)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "hip_bfv")

TRAIT_METHODS = """negate_inplace negate add_inplace add add_many multiply_many sub_inplace sub multiply_inplace multiply
square_inplace square mod_switch_to_next mod_switch_to_next_inplace mod_switch_to_next_plaintext
mod_switch_to_next_inplace_plaintext exponentiate exponentiate_inplace add_plain add_plain_inplace sub_plain
sub_plain_inplace multiply_plain multiply_plain_inplace relinearize_inplace relinearize rotate_rows rotate_rows_inplace
rotate_columns rotate_columns_inplace""".split()


def _split_args(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _declared():
    header = open(os.path.join(ROOT, "include", "hipbfv.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    decl = {}
    for m in re.finditer(r"^long\s+([A-Za-z_0-9]+)\s*\((.*?)\);", header, flags=re.M | re.S):
        args = m.group(2).strip()
        decl[m.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return decl


def test_every_ffi_call_of_the_rust_crate_matches_the_header():
    decl = _declared()
    calls = 0
    for f in os.listdir(os.path.join(CRATE, "src")):
        src = open(os.path.join(CRATE, "src", f)).read()
        for m in re.finditer(r"bindgen::(\$?[A-Za-z_0-9]+)\s*\(", src):
            name = m.group(1)
            if name.startswith("$"):  # macro parameter ($destroy): checked through its instantiations below
                continue
            assert name in decl, f"{f}: bindgen::{name} is not declared in include/hipbfv.h"
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0)
                i += 1
            nargs = len(_split_args(src[m.end() : i - 1]))
            assert nargs == decl[name], f"{f}: bindgen::{name} called with {nargs} arguments, declared with {decl[name]}"
            calls += 1
        for m in re.finditer(r"(?:owned_handle|plain_handle)!\(\w+,\s*(\w+)\)", src):
            assert decl.get(m.group(1)) == 1, m.group(1)
        # serialisable!(Type, ctor, X_SaveSize, X_Save, X_Load): SEAL's arities 3 / 5 / 5
        for m in re.finditer(r"^serialisable!\(\w+,\s*[\w:]+,\s*(\w+),\s*(\w+),\s*(\w+)\)", src, flags=re.M):
            assert [decl.get(m.group(i)) for i in (1, 2, 3)] == [3, 5, 5], m.group(0)
    assert calls >= 100


# what sunscreen_runtime imports from seal_fhe (runtime.rs:20-23, keys.rs, serialization.rs) plus the types its public API hands out
CLIENT_SURFACE = {
    "Modulus": "new value",
    "CoefficientModulus": "create bfv_default max_bit_count",
    "PlainModulus": "batching raw",
    "BfvEncryptionParametersBuilder": "new set_poly_modulus_degree set_coefficient_modulus set_plain_modulus set_plain_modulus_u64 build",
    "EncryptionParameters": "get_poly_modulus_degree get_scheme get_plain_modulus get_coefficient_modulus",
    "Context": "new new_insecure",
    "KeyGenerator": "new new_from_secret_key secret_key create_public_key create_relinearization_keys create_galois_keys",
    "BFVEncoder": "new get_slot_count encode_unsigned encode_signed decode_unsigned decode_signed",
    "Encryptor": "with_public_and_secret_key with_public_key with_secret_key encrypt encrypt_symmetric",
    "Decryptor": "new decrypt invariant_noise_budget invariant_noise",
}


def test_the_crate_carries_the_client_side_surface_the_runtime_imports():
    src = open(os.path.join(CRATE, "src", "client.rs")).read()
    lib = open(os.path.join(CRATE, "src", "lib.rs")).read()
    exported = lib[lib.index("pub use client::") : lib.index("pub use evaluator::")]
    for ty, methods in CLIENT_SURFACE.items():
        if ty != "Context":
            assert re.search(rf"\b{ty}\b", exported), f"{ty} is not re-exported from lib.rs"
        body = "".join(m.group(0) for m in re.finditer(rf"^impl(?:<[^>]*>)? {ty}(?:<T>)? \{{.*?^\}}", src, flags=re.M | re.S))
        for name in methods.split():
            assert re.search(rf"pub fn {name}\(", body), f"{ty}::{name}"
    for ty in "PublicKey SecretKey Ciphertext Plaintext RelinearizationKeys GaloisKeys".split():
        assert re.search(rf"^serialisable!\({ty},", src, flags=re.M), f"{ty}: ToBytes / FromBytes"
    assert "ToBytes" in exported and "FromBytes" in exported and "SecurityLevel" in exported


def test_the_crate_implements_the_whole_evaluator_trait():
    src = open(os.path.join(CRATE, "src", "evaluator.rs")).read()
    trait = src[src.index("pub trait Evaluator") : src.index("pub struct BFVEvaluator")]
    impl = src[src.index("impl Evaluator for BFVEvaluator") :]
    for name in TRAIT_METHODS:
        assert re.search(rf"fn {name}\(", trait), name
        assert re.search(rf"fn {name}\(", impl), name
    # build.rs binds the header of THIS repository and links the library where the Makefile puts it
    build = open(os.path.join(CRATE, "build.rs")).read()
    assert "include/hipbfv.h" in build and "sunscreen_amd/lib" in build and "rustc-link-lib=dylib=hipbfv" in build
