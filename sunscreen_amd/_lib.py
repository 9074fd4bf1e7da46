"""ctypes binding of include/hipbfv.h.

There is deliberately no fallback: if libhipbfv.so is missing or fails to load, importing this
module raises.  PyTorch (when present) is imported first so that both share one HIP runtime
(torch bundles its own libamdhip64 with the same SONAME).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPBFV_LIB selects another build of the same library (kernel tuning A/B runs, tools/build_variant.sh)
LIB_PATH = os.environ.get("HIPBFV_LIB") or os.path.join(_HERE, "lib", "libhipbfv.so")

S_OK = 0
E_POINTER = 0x80004003
E_INVALIDARG = 0x80070057
E_OUTOFMEMORY = 0x8007000E
E_UNEXPECTED = 0x8000FFFF
COR_E_IO = 0x80131620
COR_E_INVALIDOPERATION = 0x80131509

vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)
i64p = C.POINTER(C.c_int64)
u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)

# name -> argtypes; every function returns `long` (HRESULT)
_SIGNATURES = {
    "Modulus_Create1": [u64, vpp],
    "Modulus_Create2": [vp, vpp],
    "Modulus_Destroy": [vp],
    "Modulus_Value": [vp, u64p],
    "CoeffModulus_MaxBitCount": [u64, C.c_int, C.POINTER(C.c_int)],
    "CoeffModulus_BFVDefault": [u64, C.c_int, u64p, vpp],
    "CoeffModulus_Create1": [u64, u64, C.POINTER(C.c_int), vpp],
    "EncParams_Create1": [C.c_uint8, vpp],
    "EncParams_Destroy": [vp],
    "EncParams_SetPolyModulusDegree": [vp, u64],
    "EncParams_GetPolyModulusDegree": [vp, u64p],
    "EncParams_SetCoeffModulus": [vp, u64, vpp],
    "EncParams_GetCoeffModulus": [vp, u64p, vpp],
    "EncParams_SetPlainModulus1": [vp, vp],
    "EncParams_SetPlainModulus2": [vp, u64],
    "EncParams_GetPlainModulus": [vp, vpp],
    "EncParams_GetScheme": [vp, C.POINTER(C.c_uint8)],
    "SEALContext_Create": [vp, C.c_bool, C.c_int, vpp],
    "SEALContext_Destroy": [vp],
    "Plaintext_Create1": [vp, vpp],
    "Plaintext_Create5": [vp, vpp], "Plaintext_Create4": [C.c_char_p, vp, vpp],
    "Plaintext_Destroy": [vp],
    "Plaintext_CoeffCount": [vp, u64p],
    "Plaintext_CoeffAt": [vp, u64, u64p],
    "Plaintext_SetCoeffAt": [vp, u64, u64],
    "Plaintext_Resize": [vp, u64],
    "Plaintext_IsNTTForm": [vp, C.POINTER(C.c_bool)],
    "Ciphertext_Create1": [vp, vpp],
    "Ciphertext_Create2": [vp, vpp],
    "Ciphertext_Destroy": [vp],
    "Ciphertext_Size": [vp, u64p],
    "Ciphertext_CoeffModulusSize": [vp, u64p],
    "Ciphertext_PolyModulusDegree": [vp, u64p],
    "Ciphertext_GetDataAt1": [vp, u64, u64p],
    "Ciphertext_GetDataAt2": [vp, u64, u64, u64p],
    "Ciphertext_IsNTTForm": [vp, C.POINTER(C.c_bool)],
    "Plaintext_SaveSize": [vp, C.c_uint8, C.POINTER(C.c_int64)],
    "Plaintext_Save": [vp, C.c_char_p, u64, C.c_uint8, C.POINTER(C.c_int64)],
    "Plaintext_Load": [vp, vp, C.c_char_p, u64, C.POINTER(C.c_int64)],
    "Ciphertext_SaveSize": [vp, C.c_uint8, C.POINTER(C.c_int64)],
    "Ciphertext_Save": [vp, C.c_char_p, u64, C.c_uint8, C.POINTER(C.c_int64)],
    "Ciphertext_Load": [vp, vp, C.c_char_p, u64, C.POINTER(C.c_int64)],
    "KSwitchKeys_SaveSize": [vp, C.c_uint8, C.POINTER(C.c_int64)],
    "KSwitchKeys_Save": [vp, C.c_char_p, u64, C.c_uint8, C.POINTER(C.c_int64)],
    "KSwitchKeys_Load": [vp, vp, C.c_char_p, u64, C.POINTER(C.c_int64)],
    "hipbfv_wire_parms_id": [u64, u64p, u64, u64, C.c_char_p],
    "hipbfv_wire_decode_ciphertext": [C.c_char_p, u64, C.c_char_p, C.POINTER(C.c_bool), u64p, u64p, u64p, u64p, u64, C.POINTER(C.c_int64)],
    "hipbfv_wire_encode_ciphertext": [C.c_char_p, C.c_bool, u64, u64, u64, u64p, C.c_uint8, C.c_char_p, u64, C.POINTER(C.c_int64)],
    "hipbfv_wire_decode_plaintext": [C.c_char_p, u64, C.c_char_p, u64p, u64p, u64, C.POINTER(C.c_int64)],
    "KSwitchKeys_Create1": [vpp],
    "KSwitchKeys_Create2": [vp, vpp],
    "KSwitchKeys_Destroy": [vp],
    "Evaluator_Create": [vp, vpp],
    "Evaluator_Destroy": [vp],
    "Evaluator_Negate": [vp, vp, vp],
    "Evaluator_Add": [vp, vp, vp, vp],
    "Evaluator_AddMany": [vp, u64, vpp, vp],
    "Evaluator_Sub": [vp, vp, vp, vp],
    "Evaluator_Multiply": [vp, vp, vp, vp, vp],
    "Evaluator_MultiplyMany": [vp, u64, vpp, vp, vp, vp],
    "Evaluator_Square": [vp, vp, vp, vp],
    "Evaluator_Relinearize": [vp, vp, vp, vp, vp],
    "Evaluator_Exponentiate": [vp, vp, u64, vp, vp, vp],
    "Evaluator_AddPlain": [vp, vp, vp, vp],
    "Evaluator_SubPlain": [vp, vp, vp, vp],
    "Evaluator_MultiplyPlain": [vp, vp, vp, vp, vp],
    "Evaluator_RotateRows": [vp, vp, C.c_int, vp, vp, vp],
    "Evaluator_RotateColumns": [vp, vp, vp, vp, vp],
    "hipbfv_version": [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)],
    "hipbfv_last_error": [C.c_char_p, u64],
    "hipbfv_build_flags": [C.c_char_p, u64],
    "hipbfv_set_device": [C.c_int],
    "hipbfv_set_throw_on_transparent": [C.c_bool],
    "hipbfv_Context_Create": [u64, u64p, u64, u64, vpp],
    "hipbfv_Context_Info": [vp, u64p, u64p, u64p, u64p],
    "hipbfv_Context_GetPrime": [vp, u64, u64p],
    "hipbfv_Context_AuxBase": [vp, u64p, u64p, u64, C.POINTER(C.c_int)],
    "hipbfv_Ciphertext_Assign": [vp, vp, u64, u64p],
    "hipbfv_Ciphertext_Export": [vp, u64p, u64],
    "hipbfv_Ciphertext_DevicePtr": [vp, C.POINTER(u64p)],
    "hipbfv_KSwitchKeys_AssignRelin": [vp, vp, u64p],
    "hipbfv_KSwitchKeys_AssignGalois": [vp, vp, C.c_uint32, u64p],
    "hipbfv_KSwitchKeys_DevicePtr": [vp, u64, C.POINTER(u64p)],
    "hipbfv_batch_multiply": [vp, vp, u64, vp, u64, vp, u64, vp],
    "hipbfv_batch_relinearize": [vp, vp, vp, vp, u64, vp],
    "hipbfv_batch_multiply_relin": [vp, vp, vp, vp, vp, u64, vp],
    "hipbfv_batch_apply_galois": [vp, vp, C.c_uint32, vp, vp, u64, vp],
    "hipbfv_batch_rotate_rows": [vp, vp, C.c_int, vp, vp, u64, vp],
    "hipbfv_batch_rotate_columns": [vp, vp, vp, vp, u64, vp],
    "hipbfv_batch_relinearize_keys": [vp, vp, vpp, u64, C.POINTER(C.c_uint32), vp, u64, vp],
    "hipbfv_batch_multiply_relin_keys": [vp, vp, vp, vpp, u64, C.POINTER(C.c_uint32), vp, u64, vp],
    "hipbfv_batch_apply_galois_keys": [vp, vp, C.c_uint32, vpp, u64, C.POINTER(C.c_uint32), vp, u64, vp],
    "hipbfv_batch_rotate_rows_keys": [vp, vp, C.c_int, vpp, u64, C.POINTER(C.c_uint32), vp, u64, vp],
    "hipbfv_batch_rotate_columns_keys": [vp, vp, vpp, u64, C.POINTER(C.c_uint32), vp, u64, vp],
    "hipbfv_batch_add": [vp, vp, vp, vp, u64, u64, vp],
    "hipbfv_batch_sub": [vp, vp, vp, vp, u64, u64, vp],
    "hipbfv_batch_negate": [vp, vp, vp, u64, u64, vp],
    "hipbfv_batch_add_plain": [vp, vp, u64, vp, u64, vp, u64, vp],
    "hipbfv_batch_sub_plain": [vp, vp, u64, vp, u64, vp, u64, vp],
    "hipbfv_batch_multiply_plain": [vp, vp, u64, vp, u64, vp, u64, vp],
    "hipbfv_batch_ntt": [vp, vp, u64, u64, C.c_bool, vp],
    "hipbfv_batch_mod_switch": [vp, vp, u64, vp, u64, vp],
    "hipbfv_Context_NextLevel": [vp, vpp],
    "Evaluator_ModSwitchToNext1": [vp, vp, vp, vp],
    "Evaluator_ModSwitchToNext2": [vp, vp, vp],
    "hipbfv_batch_encode": [vp, vp, vp, u64, C.c_int, vp],
    "hipbfv_batch_plain_to_ntt": [vp, vp, u64, vp, u64, vp],
    "hipbfv_batch_ct_to_ntt": [vp, vp, u64, vp, u64, vp],
    "hipbfv_batch_dot_plain_ntt": [vp, vp, u64, vp, u64, vp, vp],
    "hipbfv_batch_decode": [vp, vp, vp, u64, C.c_int, vp],
    "hipbfv_batch_decrypt": [vp, vp, C.c_uint32, vp, vp, u64, vp],
    "hipbfv_batch_encrypt": [vp, vp, u64, vp, u64, u64, vp, u64, vp],
    "hipbfv_batch_encrypt_seeded": [vp, vp, u64, vp, C.c_char_p, u64, vp, u64, vp],
    "hipbfv_SecretKey_Assign": [vp, vp, u64p],
    "hipbfv_PublicKey_Assign": [vp, vp, u64p],
    "hipbfv_Encryptor_SetSeed": [vp, u64],
    "SecretKey_Create1": [vpp], "SecretKey_Create2": [vp, vpp], "SecretKey_Destroy": [vp],
    "SecretKey_SaveSize": [vp, C.c_uint8, i64p], "SecretKey_Save": [vp, C.c_char_p, u64, C.c_uint8, i64p],
    "SecretKey_Load": [vp, vp, C.c_char_p, u64, i64p],
    "PublicKey_Create1": [vpp], "PublicKey_Create2": [vp, vpp], "PublicKey_Destroy": [vp],
    "PublicKey_SaveSize": [vp, C.c_uint8, i64p], "PublicKey_Save": [vp, C.c_char_p, u64, C.c_uint8, i64p],
    "PublicKey_Load": [vp, vp, C.c_char_p, u64, i64p],
    "hipbfv_KSwitchKeys_Read": [vp, u64, u64p], "hipbfv_KSwitchKeys_Has": [vp, u64, C.POINTER(C.c_bool)],
    "hipbfv_SecretKey_Read": [vp, u64p], "hipbfv_PublicKey_Read": [vp, u64p],
    "KeyGenerator_Create1": [vp, vpp], "KeyGenerator_Create2": [vp, vp, vpp], "KeyGenerator_Destroy": [vp],
    "KeyGenerator_SecretKey": [vp, vpp], "KeyGenerator_CreatePublicKey": [vp, C.c_bool, vpp],
    "KeyGenerator_CreateRelinKeys": [vp, C.c_bool, vpp],
    "KeyGenerator_CreateGaloisKeysFromElts": [vp, u64, C.POINTER(C.c_uint32), C.c_bool, vpp],
    "KeyGenerator_CreateGaloisKeysAll": [vp, C.c_bool, vpp],
    "KeyGenerator_CreateGaloisKeysFromSteps": [vp, u64, C.POINTER(C.c_int), C.c_bool, vpp],
    "hipbfv_KeyGenerator_SetSeed": [vp, u64], "hipbfv_KeyGenerator_CreateSeeded": [vp, u64, vpp],
    "BatchEncoder_Create": [vp, vpp], "BatchEncoder_Destroy": [vp],
    "BatchEncoder_Encode1": [vp, u64, u64p, vp], "BatchEncoder_Encode2": [vp, u64, C.POINTER(C.c_int64), vp],
    "BatchEncoder_Decode1": [vp, vp, u64p, u64p, vp], "BatchEncoder_Decode2": [vp, vp, u64p, C.POINTER(C.c_int64), vp],
    "BatchEncoder_GetSlotCount": [vp, u64p],
    "Decryptor_Create": [vp, vp, vpp], "Decryptor_Destroy": [vp], "Decryptor_Decrypt": [vp, vp, vp],
    "Decryptor_InvariantNoiseBudget": [vp, vp, C.POINTER(C.c_int)], "Decryptor_InvariantNoise": [vp, vp, C.POINTER(C.c_double)],
    "Encryptor_Create": [vp, vp, vp, vpp], "Encryptor_Destroy": [vp], "Encryptor_Encrypt": [vp, vp, vp, vp],
    "Encryptor_EncryptReturnComponents": [vp, vp, C.c_bool, vp, vp, vp, vp, vp],
    "Encryptor_EncryptReturnComponentsSetSeed": [vp, vp, C.c_bool, vp, vp, vp, vp, vp, vp],
    "Encryptor_EncryptSymmetric": [vp, vp, C.c_bool, vp, vp],
    "Encryptor_EncryptSymmetricReturnComponents": [vp, vp, vp, vp, vp, vp],
    "Encryptor_EncryptSymmetricReturnComponentsSetSeed": [vp, vp, vp, vp, vp, vp, vp],
    "PolynomialArray_Create": [vp, vpp], "PolynomialArray_CreateFromCiphertext": [vp, vp, vp, vpp],
    "PolynomialArray_CreateFromPublicKey": [vp, vp, vp, vpp], "PolynomialArray_CreateFromSecretKey": [vp, vp, vp, vpp],
    "PolynomialArray_Copy": [vp, vpp], "PolynomialArray_Destroy": [vp],
    "PolynomialArray_IsReserved": [vp, C.POINTER(C.c_bool)], "PolynomialArray_IsRns": [vp, C.POINTER(C.c_bool)],
    "PolynomialArray_ToRns": [vp], "PolynomialArray_ToMultiprecision": [vp],
    "PolynomialArray_PolySize": [vp, u64p], "PolynomialArray_PolyModulusDegree": [vp, u64p],
    "PolynomialArray_CoeffModulusSize": [vp, u64p], "PolynomialArray_ExportSize": [vp, u64p],
    "PolynomialArray_PerformExport": [vp, u64p], "PolynomialArray_Drop": [vp, vpp],
    "hipbfv_set_chunk_ops": [vp, u64],
    "hipbfv_Program_Create": [vpp],
    "hipbfv_Program_Destroy": [vp],
    "hipbfv_Program_AddNode": [vp, C.c_uint32, u64, C.POINTER(C.c_uint32)],
    "hipbfv_Program_AddPlaintextLiteral": [vp, C.c_char_p, u64, C.POINTER(C.c_uint32)],
    "hipbfv_Program_AddEdge": [vp, C.c_uint32, C.c_uint32, C.c_uint32],
    "hipbfv_Program_LoadJson": [vp, C.c_char_p, u64],
    "hipbfv_batch_status": [vp, C.POINTER(C.c_uint64), vp],
    "hipbfv_set_batch_transparent_check": [vp, C.c_bool],
    "hipbfv_Program_NumOutputs": [vp, u64p],
    "hipbfv_Program_Describe": [vp, C.c_char_p, C.c_uint64, u64p],
    "hipbfv_debug_f64_plan": [u64, C.c_uint32, C.POINTER(C.c_uint32)],
    "hipbfv_debug_aux_base": [u64, u64p, u64, u64, u64p, u64p, u64, C.POINTER(C.c_int)],
    "hipbfv_debug_graph_probe": [vp, vp, vp, vp, vp, u64, C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "hipbfv_Program_Run": [vp, vp, u64, u64, C.POINTER(C.c_uint32), vpp, u64p, vp, vp, u64, vpp, vp],
    "hipbfv_Program_RunKeys": [vp, vp, u64, u64, C.POINTER(C.c_uint32), vpp, u64p, u64, vpp, vpp, C.POINTER(C.c_uint32), u64, vpp, vp],
    "hipbfv_profile_enable": [vp, C.c_bool],
    "hipbfv_profile_reset": [vp],
    "hipbfv_profile_kernel_count": [C.POINTER(C.c_uint32)],
    "hipbfv_profile_read": [vp, C.c_uint32, C.c_char_p, u64, C.POINTER(C.c_double), u64p, u64p],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load libhipbfv.so (once).  Raises OSError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sunscreen_amd/csrc` -- there is no CPU fallback"
        )
    try:  # share torch's HIP runtime when torch is installed
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    # A/B runs against a library built from an OLDER tree (tools/ab_libs.sh with HIPBFV_LIB): entry points added since are absent
    # there.  Only honoured together with HIPBFV_LIB -- the default library must export everything (tests/test_cabi_cpu.py).
    tolerate = bool(os.environ.get("HIPBFV_LIB")) and os.environ.get("HIPBFV_LIB_ALLOW_MISSING") == "1"
    for name, args in _SIGNATURES.items():
        if tolerate and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_long
    _lib = lib
    return lib


def build_flags() -> str:
    """The compiler flags (macro definitions included) the loaded libhipbfv.so was built with."""
    buf = C.create_string_buffer(2048)
    load().hipbfv_build_flags(buf, 2048)
    return buf.value.decode(errors="replace")


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().hipbfv_last_error(buf, 512)
    return buf.value.decode(errors="replace")
